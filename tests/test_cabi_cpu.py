"""C-ABI checks that need no GPU: the gfx950 library loads, exports every symbol include/russell_hipmf.h
declares, and the product refuses to run without a HIP device (no CPU fallback)."""
import os
import re

import pytest

from russell_amd import _capi

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _build():
    import __graft_entry__ as g
    g.build()


def test_library_exports_every_declared_symbol():
    _build()
    lib = _capi.load()
    header = open(os.path.join(ROOT, "include", "russell_hipmf.h")).read()
    declared = set(re.findall(r"\b((?:complex_solver_hipmf|solver_hipmf|hipmf)_[a-z0-9_]+)\s*\(", header))
    assert declared, "no declarations parsed"
    assert declared == set(_capi.SYMBOLS), declared ^ set(_capi.SYMBOLS)
    for name in declared:
        assert hasattr(lib, name)


def test_no_device_means_loud_failure():
    _build()
    lib = _capi.load()
    if lib.hipmf_device_count() > 0:
        pytest.skip("a GPU is visible here")
    from russell_amd.backend import Hipmf
    with pytest.raises(RuntimeError):
        Hipmf()
    assert lib.solver_hipmf_new() is None
    lib.solver_hipmf_drop(None)  # NULL-safe like solver_cudss_drop (interface_cudss.cu:126-129)
