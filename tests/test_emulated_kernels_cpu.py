"""Kernel-logic regression on the CPU: the HIP kernels compiled against tools/hipemu (a development-only emulator:
fibers for __syncthreads, wave shuffles, the MFMA lane maps) and driven through the same C-ABI.

This is NOT parity evidence (parity is measured on a real MI355X by the -m gpu tests) and the emulated library is never
part of the product; it catches indexing / barrier / assembly-map mistakes before a GPU run."""
import os
import subprocess

import numpy as np
import pytest

from russell_amd import problems as P
from russell_amd.backend import Hipmf

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
EMU = os.path.join(ROOT, "tests", "emu", "libhipmf_emu.so")
CSRC = os.path.join(ROOT, "russell_amd", "csrc")


@pytest.fixture(scope="module")
def emu_lib():
    srcs = [os.path.join(CSRC, f) for f in ("symbolic.cpp", "matching.cpp", "numeric.cpp", "interface_hipmf.cpp")]
    deps = srcs + [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".hpp")]
    deps += [os.path.join(ROOT, "tools", "hipemu", "hip", "hip_runtime.h"), os.path.join(ROOT, "tools", "hipemu", "hipmf_device_rt.h")]
    if not os.path.exists(EMU) or any(os.path.getmtime(d) > os.path.getmtime(EMU) for d in deps):
        os.makedirs(os.path.dirname(EMU), exist_ok=True)
        subprocess.check_call(["g++", "-O1", "-std=c++17", "-fPIC", "-shared", "-w", "-I", os.path.join(ROOT, "tools", "hipemu"), "-I", CSRC,
                               "-x", "c++"] + srcs + ["-o", EMU])
    return EMU


def _solve(lib, n, rp, ci, v, b, **kw):
    s = Hipmf(lib)
    assert s.initialize(n, rp, ci, **kw) == 0
    code = s.factorize(v, compute_determinant=True)
    x = s.solve(b)
    st = s.stats()
    return s, code, x, st


def test_emulated_small_front_path(emu_lib):
    n, rp, ci, v = P.poisson2d(12, 9)
    xs = P.manufactured_solution(n)
    s, code, x, st = _solve(emu_lib, n, rp, ci, v, P.csr_matvec(n, rp, ci, v, xs))
    assert code == 0 and st["max_front"] <= 64
    assert np.max(np.abs(x - xs)) < 1e-12
    s.close()


def test_emulated_tiled_augmented_path(emu_lib):
    n, rp, ci, v = P.poisson2d(44, 40)  # top separators give fronts > 64: k_panel / k_update (MFMA) / k_fwd_big / k_bwd_big
    xs = P.manufactured_solution(n)
    s, code, x, st = _solve(emu_lib, n, rp, ci, v, P.csr_matvec(n, rp, ci, v, xs))
    assert code == 0 and st["max_front"] > 64
    assert np.max(np.abs(x - xs)) < 1e-11
    s.close()


def test_emulated_pivoting_and_determinant(emu_lib):
    dense = np.array([[2.0, 3.0, 0, 0, 0], [3.0, 0, 4.0, 0, 6.0], [0, -1.0, -3.0, 2.0, 0], [0, 0, 1.0, 0, 0], [0, 4.0, 2.0, 0, 1.0]])
    r, c = np.nonzero(dense)
    rp = np.concatenate([[0], np.cumsum(np.bincount(r, minlength=5))]).astype(np.int32)
    s, code, x, st = _solve(emu_lib, 5, rp, c.astype(np.int32), dense[r, c], np.array([8.0, 45.0, -3.0, 3.0, 19.0]))
    assert code == 0 and np.max(np.abs(x - np.arange(1, 6))) < 1e-13
    assert abs(s.det_coefficient * 10.0 ** s.det_exponent - 114.0) < 1e-10
    s.close()
