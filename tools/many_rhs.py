#!/usr/bin/env python3
"""Many right-hand sides on one GPU: solve_device with nrhs columns resident in HBM (blocks of SF_KMAX = 8 columns share
one read of the factor) against nrhs single solves.  usage: many_rhs.py 2d|3d N nrhs [refinement_nstep]"""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from russell_amd import problems as P
from russell_amd.backend import Hipmf
kind, N, nrhs = sys.argv[1], int(sys.argv[2]), int(sys.argv[3])
nref = int(sys.argv[4]) if len(sys.argv) > 4 else -1
n, rp, ci, v = (P.poisson2d(N) if kind == "2d" else P.poisson3d(N))
rng = np.random.default_rng(20260927)
XS = rng.standard_normal((nrhs, n))
B = np.array([P.csr_matvec(n, rp, ci, v, XS[j]) for j in range(nrhs)])
s = Hipmf()
assert s.initialize(n, rp, ci, refinement_nstep=nref) == 0
d_v, d_b, d_x = s.dev_alloc(v.nbytes), s.dev_alloc(B.nbytes), s.dev_alloc(B.nbytes)
s.h2d(d_v, v), s.h2d(d_b, B)
assert s.factorize_device(d_v) == 0
def timed(fn, reps=3):
    fn(); s.lib.hipmf_device_synchronize()
    t0 = time.perf_counter()
    for _ in range(reps): fn()
    s.lib.hipmf_device_synchronize()
    return (time.perf_counter() - t0) / reps * 1e3
t_blk = timed(lambda: s.solve_device(d_x, d_b, nrhs, n))
X = np.zeros_like(B); s.d2h(X, d_x)
def singles():
    for j in range(nrhs): s.solve_device(d_x + j * n * 8, d_b + j * n * 8, 1, n)
t_one = timed(singles)
X1 = np.zeros_like(B); s.d2h(X1, d_x)
st = s.stats()
fac_bytes = (st["nnz_l"] + st["nnz_u"]) * 12
print("%s N=%d n=%d nrhs=%d: blocked %.2f ms (%.3f ms/rhs), single %.2f ms (%.3f ms/rhs), speed-up %.2fx; bitwise equal %s; max err %.2e" %
      (kind, N, n, nrhs, t_blk, t_blk / nrhs, t_one, t_one / nrhs, t_one / t_blk, np.array_equal(X, X1), np.max(np.abs(X - XS))))
s.close()
