#!/usr/bin/env python3
"""Unsymmetric-valued 3D stress of the large-front schedules (4 / 8 panels per pass, split chains): the 7-point pattern of
poisson3d(N) with random unsymmetric off-diagonals, a diagonal that is NOT dominant in ~10 % of the rows, and random row
scaling 10^U(-4, 4); solved through the reference-style path (values known at initialize).  usage: stress3d_unsym.py N [seed]"""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from russell_amd import problems as P
from russell_amd.backend import Hipmf

N = int(sys.argv[1]) if len(sys.argv) > 1 else 48
seed = int(sys.argv[2]) if len(sys.argv) > 2 else 20260927
n, rp, ci, v = P.poisson3d(N)
rng = np.random.default_rng(seed)
rows = np.repeat(np.arange(n), np.diff(rp))
diag = rows == ci
v = v.copy()
v[~diag] *= 1.0 + 0.8 * rng.uniform(-1.0, 1.0, int(np.sum(~diag)))      # unsymmetric values
weak = rng.random(n) < 0.1
v[diag] = np.where(weak, 0.05 * rng.uniform(-1.0, 1.0, n), 6.0 + rng.uniform(-0.5, 0.5, n))  # weak / signed diagonal entries
v *= (10.0 ** rng.uniform(-4.0, 4.0, n))[rows]                          # bad row scaling
xs = P.manufactured_solution(n)
b = P.csr_matvec(n, rp, ci, v, xs)
s = Hipmf()
t0 = time.perf_counter()
assert s.initialize(n, rp, ci, values=v) == 0
t1 = time.perf_counter()
st = s.stats()
code = s.factorize(v)
x = s.solve(b)
t2 = time.perf_counter()
st2 = s.stats()
r = P.csr_matvec(n, rp, ci, v, x) - b
rel = float(np.max(np.abs(r)) / (np.max(np.abs(v)) + 1.0))
roww = np.zeros(n)
np.maximum.at(roww, rows, np.abs(v))
print("N=%d n=%d max_front=%d max_pivots=%d nsuper=%d levels=%d: initialize %.2f s, factorize code %d, perturbed pivots %d, refinement steps %d"
      % (N, n, st["max_front"], st["max_pivots"], st["nsuper"], st["nlevels"], t1 - t0, code, st2["n_perturbed"], st2["refinement_steps"]))
print("   relative_error (VerifyLinSys metric) %.2e; scaled residual max_i |r_i| / max_j |a_ij| %.2e; max |x - x*| %.2e; factor %.1f ms"
      % (rel, float(np.max(np.abs(r) / roww)), float(np.max(np.abs(x - xs))), st2["factor_ms"]))
s.close()
