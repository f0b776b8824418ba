"""The HIP kernels compiled against tools/hipemu with AddressSanitizer: out-of-bounds reads that the plain emulator (host memory: no
fault) and the GPU tests (a fault only at a page boundary) can both miss.  Round 5: tools/fuzz.py found on the device that the backward
leaf kernel of the blocked solves read the row structure past its end for a front that is root and leaf at once; this test is what
would have caught it on the CPU.  The emulator is a development tool, not parity evidence."""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "russell_amd", "csrc")
LIB = os.path.join(ROOT, "tests", "emu", "libhipmf_emu_asan.so")

SCRIPT = r'''
import os, sys
import numpy as np
import scipy.sparse as sp
sys.path.insert(0, %(root)r)
from russell_amd import problems as P
from russell_amd.backend import Hipmf
lib = %(lib)r
rng = np.random.default_rng(77)
cases = []
for n, blocks in ((1, 1), (5, 1), (16, 1), (12, 3), (40, 8), (96, 6)):  # fronts that are root and leaf at once, disconnected blocks
    A = sp.lil_matrix((n, n))
    size = n // blocks
    for b in range(blocks):
        lo, hi = b * size, (n if b == blocks - 1 else (b + 1) * size)
        A[lo:hi, lo:hi] = rng.uniform(-1, 1, (hi - lo, hi - lo)) + 4.0 * np.eye(hi - lo)
    A = sp.csr_matrix(A)
    A.sort_indices()
    cases.append((n, A.indptr.astype(np.int32), A.indices.astype(np.int32), A.data.astype(np.float64), A, {}))
for mat in (P.poisson2d(60, 50), P.poisson3d(9), P.convection_diffusion2d(30, peclet=30.0, scale_decades=0.0)):
    n, rp, ci, v = mat
    cases.append((n, rp, ci, v, sp.csr_matrix((v, ci, rp), shape=(n, n)), {}))
n, rp, ci, v = P.poisson2d(48, 44)
lrp, lci, lv = P.lower_triangle(n, rp, ci, v)
cases.append((n, lrp, lci, lv, sp.csr_matrix((v, ci, rp), shape=(n, n)), {"general_symmetric": True}))
for env in ({}, {"HIPMF_TAG_SOLVE": "0"}, {"HIPMF_UP_TOP_FRONTS": "1"}):
    os.environ.update(env)
    for n, rp, ci, v, A, kw in cases:
        s = Hipmf(lib)
        assert s.initialize(n, rp, ci, **kw) == 0
        assert s.factorize(v) == 0
        for nrhs in (1, 2, 18, 40):  # (18, 40: two and three block groups per launch, the last group partly filled)
            XS = rng.standard_normal((nrhs, n))
            B = np.array([A @ XS[j] for j in range(nrhs)])
            X = s.solve_many(B) if nrhs > 1 else s.solve(B[0])[None, :]
            assert np.max(np.abs(X - XS)) <= 1e-9 * max(1.0, np.max(np.abs(XS))), (n, nrhs, env)
        s.close()
    for k in env:
        del os.environ[k]
print("asan run done")
'''


def test_solve_kernels_under_address_sanitizer():
    asan = subprocess.run(["gcc", "-print-file-name=libasan.so"], capture_output=True, text=True).stdout.strip()
    if not asan or not os.path.exists(asan):
        pytest.skip("no libasan in this toolchain")
    srcs = [os.path.join(CSRC, f) for f in ("symbolic.cpp", "matching.cpp", "numeric.cpp", "interface_hipmf.cpp", "interface_complex_hipmf.cpp", "fdm_device.cpp")]
    deps = srcs + [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".hpp")]
    deps += [os.path.join(ROOT, "tools", "hipemu", "hip", "hip_runtime.h"), os.path.join(ROOT, "tools", "hipemu", "hipmf_device_rt.h")]
    if not os.path.exists(LIB) or any(os.path.getmtime(d) > os.path.getmtime(LIB) for d in deps):
        os.makedirs(os.path.dirname(LIB), exist_ok=True)
        subprocess.check_call(["g++", "-O1", "-g", "-fsanitize=address", "-fno-omit-frame-pointer", "-std=c++17", "-fPIC", "-shared", "-w", "-I",
                               os.path.join(ROOT, "tools", "hipemu"), "-I", CSRC, "-x", "c++"] + srcs + ["-o", LIB])
    env = dict(os.environ, LD_PRELOAD=asan, ASAN_OPTIONS="detect_leaks=0:detect_stack_use_after_return=0")
    r = subprocess.run([sys.executable, "-c", SCRIPT % {"root": ROOT, "lib": LIB}], env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0 and "asan run done" in r.stdout, (r.stdout[-500:], r.stderr[-3000:])
