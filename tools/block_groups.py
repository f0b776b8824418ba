#!/usr/bin/env python3
"""Blocks of right-hand sides per dependency-driven launch (HIPMF_BLOCK_GROUPS, kernels_solve_fused.hpp SfGroups): the same nrhs independent
random right-hand sides solved with 1, 2 and 4 groups per launch in ONE process (one handle per setting).
usage: block_groups.py 2d|3d|3dl N nrhs [groups ...]     (3dl: the lower triangle, L D L^T)"""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from russell_amd import problems as P
from russell_amd.backend import Hipmf
kind, N, nrhs = sys.argv[1], int(sys.argv[2]), int(sys.argv[3])
groups = [int(a) for a in sys.argv[4:]] or [1, 2, 4]
n, rp, ci, v = (P.poisson2d(N) if kind == "2d" else P.poisson3d(N))
B = np.empty((nrhs, n))
for j in range(nrhs):
    B[j] = np.random.default_rng([20260927, j]).standard_normal(n)
import scipy.sparse as sp
A = sp.csr_matrix((v, ci, rp), shape=(n, n))
if kind == "3dl":
    lrp, lci, lv = P.lower_triangle(n, rp, ci, v)
ref = None
for g in groups:
    os.environ["HIPMF_BLOCK_GROUPS"] = str(g)
    s = Hipmf()
    if kind == "3dl":
        assert s.initialize(n, lrp, lci, general_symmetric=True) == 0
        vv = lv
    else:
        assert s.initialize(n, rp, ci) == 0
        vv = v
    d_v, d_b, d_x = s.dev_alloc(vv.nbytes), s.dev_alloc(B.nbytes), s.dev_alloc(B.nbytes)
    s.h2d(d_v, vv), s.h2d(d_b, B)
    assert s.factorize_device(d_v) == 0
    s.lib.hipmf_device_synchronize()
    t0 = time.perf_counter()
    s.solve_device(d_x, d_b, nrhs, n)
    s.lib.hipmf_device_synchronize()
    t_first = (time.perf_counter() - t0) * 1e3
    ts = []
    for _ in range(3):
        t0 = time.perf_counter()
        s.solve_device(d_x, d_b, nrhs, n)
        s.lib.hipmf_device_synchronize()
        ts.append((time.perf_counter() - t0) * 1e3)
    X = np.zeros_like(B)
    s.d2h(X, d_x)
    worst = 0.0
    for j0 in range(0, nrhs, 32):
        R = A @ X[j0:j0 + 32].T - B[j0:j0 + 32].T
        worst = max(worst, float(np.max(np.abs(R))) / (float(np.max(np.abs(v))) + 1.0))
    if ref is None:
        ref = X
    st = s.stats()
    print("%s N=%d n=%d nrhs=%d groups=%d (in use %d): first call %.1f ms, then %s ms -> %.4f ms/rhs; refinement steps %d; relative_error %.2e; "
          "bitwise equal to the first setting %s (max diff %.1e); fallbacks %d" %
          (kind, N, n, nrhs, g, s.counter("block_groups"), t_first, ["%.1f" % t for t in ts], min(ts) / nrhs, st.get("refinement_steps", -1), worst,
           np.array_equal(X, ref), float(np.max(np.abs(X - ref))), s.counter("fused_fallbacks")), flush=True)
    for p in (d_v, d_b, d_x):
        s.dev_free(p)
    s.close()
