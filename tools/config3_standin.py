#!/usr/bin/env python3
"""BASELINE config 3 stand-in (bbmat / af_shell10 are not in the tree and there is no network): unsymmetric, badly scaled
convection-diffusion matrices (5-point convection-diffusion with random row scaling 10^U(-6, 6)) through the
reference-style call path (values known at initialize -> matching if the diagonal is weak), against scipy's SuperLU
(tier-2 CPU baseline of SURVEY.md 8d: single-threaded, COLAMD; "SuperLU stand-in, not UMFPACK") on the same box.
usage: config3_standin.py nx [peclet]"""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from russell_amd import problems as P
from russell_amd.backend import Hipmf

nx = int(sys.argv[1]) if len(sys.argv) > 1 else 500
pe = float(sys.argv[2]) if len(sys.argv) > 2 else 0.7
n, rp, ci, v = P.convection_diffusion2d(nx, peclet=pe)
xs = P.manufactured_solution(n)
b = P.csr_matvec(n, rp, ci, v, xs)
rows = np.repeat(np.arange(n), np.diff(rp))
amax = np.max(np.abs(v))
print("convection-diffusion %dx%d: n=%d nnz=%d, |a| in [%.1e, %.1e]" % (nx, nx, n, rp[-1], np.min(np.abs(v)), amax), flush=True)

s = Hipmf()
t0 = time.perf_counter(); assert s.initialize(n, rp, ci, values=v) == 0; t1 = time.perf_counter()
d_v, d_b, d_x = s.dev_alloc(v.nbytes), s.dev_alloc(b.nbytes), s.dev_alloc(b.nbytes)
s.h2d(d_v, v), s.h2d(d_b, b)
for rep in range(3):
    ta = time.perf_counter(); code = s.factorize_device(d_v); s.lib.hipmf_device_synchronize(); tb = time.perf_counter()
    s.solve_device(d_x, d_b); s.lib.hipmf_device_synchronize(); tc = time.perf_counter()
x = np.zeros(n); s.d2h(x, d_x)
r = np.zeros(n); np.add.at(r, rows, v * x[ci])
st = s.stats()
print("HIPMF : initialize %.2f s, factorize %.2f ms (code %d, %d perturbed), solve %.2f ms (%d refinement steps); relative_error %.2e, max |x - x*|/|x*| %.2e" %
      (t1 - t0, (tb - ta) * 1e3, code, st["n_perturbed"], (tc - tb) * 1e3, st["refinement_steps"], np.max(np.abs(r - b)) / (amax + 1.0),
       np.max(np.abs(x - xs)) / np.max(np.abs(xs))), flush=True)
s.close()

try:
    import scipy.sparse as sp
    from scipy.sparse.linalg import splu
    A = sp.csr_matrix((v, ci, rp), shape=(n, n)).tocsc()
    t0 = time.perf_counter(); lu = splu(A, permc_spec="COLAMD"); t1 = time.perf_counter()
    xo = lu.solve(b); t2 = time.perf_counter()
    ro = A @ xo - b
    print("SuperLU (scipy splu, COLAMD, 1 thread; tier-2 stand-in, not UMFPACK): factorize %.1f ms, solve %.1f ms; relative_error %.2e, max |x - x*|/|x*| %.2e" %
          ((t1 - t0) * 1e3, (t2 - t1) * 1e3, np.max(np.abs(ro)) / (amax + 1.0), np.max(np.abs(xo - xs)) / np.max(np.abs(xs))))
except Exception as e:  # scipy missing on the box
    print("scipy splu not available:", e)
