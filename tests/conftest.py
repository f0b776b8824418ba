import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def _gpu_count():
    """Number of HIP devices the in-tree library sees (0 without a GPU or without the library)."""
    try:
        import ctypes
        lib = ctypes.CDLL(os.path.join(ROOT, "russell_amd", "lib", "librussell_hipmf.so"))
        h = lib.solver_hipmf_new  # returns NULL when hipGetDeviceCount() < 1
        h.restype = ctypes.c_void_p
        p = h()
        if not p:
            return 0
        lib.solver_hipmf_drop.argtypes = [ctypes.c_void_p]
        lib.solver_hipmf_drop(p)
        return 1
    except Exception:
        return 0


def pytest_collection_modifyitems(config, items):
    """`-m gpu` tests SKIP (not fail) on a box without a device; everything else is untouched."""
    gpu_items = [it for it in items if it.get_closest_marker("gpu")]
    if not gpu_items or _gpu_count() > 0:
        return
    skip = pytest.mark.skip(reason="no HIP device visible")
    for it in gpu_items:
        it.add_marker(skip)


@pytest.fixture(scope="session", autouse=True)
def _native_pieces_are_built():
    """The tests load in-tree native libraries; build them first when a fresh checkout has none (no-op otherwise)."""
    lib = os.path.join(ROOT, "russell_amd", "lib")
    needed = [os.path.join(lib, f) for f in ("librussell_hipmf.so", "librussell_host.so", "solve_matrix_market", "brusselator_pde")]
    if not all(os.path.exists(f) for f in needed):
        import __graft_entry__

        __graft_entry__.build()


EMU = os.path.join(ROOT, "tests", "emu", "libhipmf_emu.so")
CSRC = os.path.join(ROOT, "russell_amd", "csrc")


@pytest.fixture(scope="session")
def emu_lib():
    """The HIP kernels compiled against tools/hipemu (development-only CPU emulator) behind the same C-ABI.

    Used by the CPU tests of kernel logic and of the host layers above the C-ABI; never part of the product."""
    import subprocess

    srcs = [os.path.join(CSRC, f) for f in ("symbolic.cpp", "matching.cpp", "numeric.cpp", "interface_hipmf.cpp", "interface_complex_hipmf.cpp", "fdm_device.cpp")]
    deps = srcs + [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".hpp")]
    deps += [os.path.join(ROOT, "tools", "hipemu", "hip", "hip_runtime.h"), os.path.join(ROOT, "tools", "hipemu", "hipmf_device_rt.h")]
    if not os.path.exists(EMU) or any(os.path.getmtime(d) > os.path.getmtime(EMU) for d in deps):
        os.makedirs(os.path.dirname(EMU), exist_ok=True)
        subprocess.check_call(["g++", "-O1", "-std=c++17", "-fPIC", "-shared", "-w", "-I", os.path.join(ROOT, "tools", "hipemu"), "-I", CSRC,
                               "-x", "c++"] + srcs + ["-o", EMU])
    return EMU
