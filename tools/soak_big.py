#!/usr/bin/env python3
"""Soak at the sizes of the concurrency test (tests/test_round5_gpu.py), longer and with a third handle: three host threads on one device --
a real 1M-DOF system (single solves + a re-factorisation now and then), a complex 250k-unknown system (the shape of Radau5's K_comp) and a
60^3 system solved in blocks of 16 right-hand sides -- N solves each, every dependency-driven launch behind the per-device gate.
Asserts: no fallback of any kind, every residual small, no solve longer than 50 x the median of its thread.  usage: soak_big.py [N]"""
import ctypes as C
import os
import sys
import threading
import time

import numpy as np
import scipy.sparse as sp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from russell_amd import _capi
from russell_amd import problems as P
from russell_amd.backend import Hipmf

N = int(sys.argv[1]) if len(sys.argv) > 1 else 2000
lib = _capi.load()
out = {}


def metric(A, vmax, x, b):
    return float(np.max(np.abs(A @ x - b)) / (vmax + 1.0))


def real_side():
    n, rp, ci, v = P.poisson2d(1000)
    A = sp.csr_matrix((v, ci, rp), shape=(n, n))
    b = np.random.default_rng(1).standard_normal(n)
    s = Hipmf()
    assert s.initialize(n, rp, ci) == 0 and s.factorize(v) == 0
    ts = []
    for it in range(N):
        if it % 100 == 50:
            assert s.factorize(v) == 0
        t0 = time.perf_counter()
        x = s.solve(b)
        ts.append(time.perf_counter() - t0)
    out["real 1M"] = (ts, s.counter("fused_fallbacks") + s.counter("chain_fallbacks"), s.counter("gate_waits"), metric(A, float(np.max(np.abs(v))), x, b))
    s.close()


def complex_side():
    n, rp, ci, v = P.poisson2d(500)
    rows = np.repeat(np.arange(n), np.diff(rp))
    z = -v.astype(complex)
    z[rows == ci] += (3.0 + 2.0j)
    A = sp.csr_matrix((z, ci, rp), shape=(n, n))
    zb = np.random.default_rng(2).standard_normal(2 * n)
    h = lib.complex_solver_hipmf_new()
    vals = np.ascontiguousarray(np.stack([z.real, z.imag], axis=1).ravel())
    assert lib.complex_solver_hipmf_initialize(h, 0, 1, -1.0, -1, 0, 0, n, np.ascontiguousarray(rp, dtype=np.int32), np.ascontiguousarray(ci, dtype=np.int32),
                                               vals.ctypes.data_as(C.c_void_p)) == 0
    i32 = C.c_int32
    eo, es, npert = i32(0), i32(0), i32(0)
    rc, dr, di, de = C.c_double(0), C.c_double(0), C.c_double(0), C.c_double(0)
    assert lib.complex_solver_hipmf_factorize(h, C.byref(eo), C.byref(es), C.byref(npert), C.byref(rc), C.byref(dr), C.byref(di), C.byref(de), 0, 0, vals) == 0
    x = np.zeros(2 * n)
    ts = []
    for _ in range(N):
        t0 = time.perf_counter()
        assert lib.complex_solver_hipmf_solve(h, x, zb, 0) == 0
        ts.append(time.perf_counter() - t0)
    xc, bc = x[0::2] + 1j * x[1::2], zb[0::2] + 1j * zb[1::2]
    out["complex 250k"] = (ts, lib.complex_solver_hipmf_get_counter(h, 2) + lib.complex_solver_hipmf_get_counter(h, 7), lib.complex_solver_hipmf_get_counter(h, 11),
                           float(np.max(np.abs(A @ xc - bc)) / (np.max(np.abs(z)) + 1.0)))
    lib.complex_solver_hipmf_drop(h)


def blocked_side():
    n, rp, ci, v = P.poisson3d(60)
    A = sp.csr_matrix((v, ci, rp), shape=(n, n))
    B = np.random.default_rng(3).standard_normal((16, n))
    s = Hipmf()
    assert s.initialize(n, rp, ci) == 0 and s.factorize(v) == 0
    ts = []
    for _ in range(max(1, N // 10)):
        t0 = time.perf_counter()
        X = s.solve_many(B)
        ts.append(time.perf_counter() - t0)
    worst = max(metric(A, float(np.max(np.abs(v))), X[j], B[j]) for j in range(16))
    out["60^3 x 16 columns"] = (ts, s.counter("fused_fallbacks") + s.counter("chain_fallbacks"), s.counter("gate_waits"), worst)
    s.close()


t0 = time.perf_counter()
th = [threading.Thread(target=f) for f in (real_side, complex_side, blocked_side)]
for t in th:
    t.start()
for t in th:
    t.join()
ok = len(out) == 3
for tag, (ts, fb, waits, res) in sorted(out.items()):
    med, mx = float(np.median(ts)), max(ts)
    print("%-18s %5d solves: median %.2f ms, max %.2f ms (%.1f x), fallbacks %d, waits at the device gate %d, residual %.1e" % (tag, len(ts), 1e3 * med, 1e3 * mx, mx / med, fb, waits, res))
    ok = ok and fb == 0 and res <= 1e-10 and mx < 50.0 * med
print("elapsed %.1f s; %s" % (time.perf_counter() - t0, "ok" if ok else "FAILED"))
sys.exit(0 if ok else 1)
