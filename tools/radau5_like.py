#!/usr/bin/env python3
"""BASELINE config 5 stand-in: the linear-algebra part of Radau5 steps on the Brusselator-PDE Jacobian pattern
(russell_ode/src/samples.rs:497-612, radau5.rs:195-303): per step one REAL system (gamma I - J) and one COMPLEX system
((alpha + beta i) I - J) are re-factorised with new values on a fixed structure and solved, through the host mirror of
LinSolTrait / ComplexLinSolTrait (factorize(coo, None) on repeat calls: device-side value refresh).
usage: radau5_like.py npoint [nsteps]"""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from russell_amd import problems as P
from russell_amd.sparse import ComplexCooMatrix, ComplexLinSolver, CooMatrix, Genie, LinSolver, Sym

npoint = int(sys.argv[1]) if len(sys.argv) > 1 else 129
nsteps = int(sys.argv[2]) if len(sys.argv) > 2 else 6
n, rp, ci, v0 = P.brusselator_pattern(npoint, gamma=0.0)   # v0 = -J (gamma = 0)
rows = np.repeat(np.arange(n), np.diff(rp))
diag = rows == ci
nnz = len(v0)
print("npoint %d: ndim %d, jac nnz %d" % (npoint, n, nnz), flush=True)
GAMMA, ALPHA, BETA = 3.6378342527444957, 2.6810828736277521, 3.0504301992474105  # Radau5 constants (radau5.rs)
rng = np.random.default_rng(1)
real, cplx = LinSolver(Genie.Hipmf), ComplexLinSolver(Genie.Hipmf)
coo, ccoo = CooMatrix(n, n, nnz, Sym.No), ComplexCooMatrix(n, n, nnz, Sym.No)
t_fill = t_rf = t_rs = t_cf = t_cs = 0.0
for step in range(nsteps):
    h = 1e-4 * (1.0 + 0.3 * step)
    mJ = v0 * (1.0 + 0.01 * step)                      # the Jacobian changes a little every step
    t0 = time.perf_counter()
    coo.reset(); ccoo.reset()
    kr = mJ + (GAMMA / h) * diag
    kc = mJ.astype(complex) + ((ALPHA + BETA * 1j) / h) * diag
    coo.put_many(rows, ci, kr) if hasattr(coo, "put_many") else [coo.put(int(i), int(j), float(a)) for i, j, a in zip(rows, ci, kr)]
    ccoo.put_many(rows, ci, kc) if hasattr(ccoo, "put_many") else [ccoo.put(int(i), int(j), complex(a)) for i, j, a in zip(rows, ci, kc)]
    t1 = time.perf_counter()
    real.actual.factorize(coo, None)
    t2 = time.perf_counter()
    xs = rng.standard_normal(n)
    x = real.actual.solve(P.csr_matvec(n, rp, ci, kr, xs))
    t3 = time.perf_counter()
    cplx.actual.factorize(ccoo, None)
    t4 = time.perf_counter()
    zs = rng.standard_normal(n) + 1j * rng.standard_normal(n)
    rhs = ccoo.mat_vec_mul(zs)
    t5 = time.perf_counter()
    z = cplx.actual.solve(rhs)
    t6 = time.perf_counter()
    er, ec = np.max(np.abs(x - xs)) / np.max(np.abs(xs)), np.max(np.abs(z - zs)) / np.max(np.abs(zs))
    print("step %d: real factorize %.1f ms solve %.1f ms (err %.1e) | complex factorize %.1f ms solve %.1f ms (err %.1e) | host fill %.0f ms" %
          (step, (t2 - t1) * 1e3, (t3 - t2) * 1e3, er, (t4 - t3) * 1e3, (t6 - t5) * 1e3, ec, (t1 - t0) * 1e3), flush=True)
    if step > 0:
        t_rf += t2 - t1; t_rs += t3 - t2; t_cf += t4 - t3; t_cs += t6 - t5
k = max(nsteps - 1, 1)
print("repeat-call averages: real factorize %.1f ms, solve %.1f ms; complex factorize %.1f ms, solve %.1f ms (wall, host API incl. H2D/D2H)" %
      (t_rf / k * 1e3, t_rs / k * 1e3, t_cf / k * 1e3, t_cs / k * 1e3))
