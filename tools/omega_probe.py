#!/usr/bin/env python3
"""Componentwise backward error of the FIRST solve (no refinement) of random right-hand sides at C2: single-column kernels against the
blocked instances, with and without the leaf kernels.  omega = max_i |r_i| / (|A||x| + |b|)_i (the measure of Solver::solve's stopping rule)."""
import os, sys
import numpy as np
import scipy.sparse as sp
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from russell_amd import problems as P
from russell_amd.backend import Hipmf
n, rp, ci, v = P.poisson2d(int(sys.argv[1]) if len(sys.argv) > 1 else 1000)
A = sp.csr_matrix((v, ci, rp), shape=(n, n)); Aabs = abs(A)
B = np.array([np.random.default_rng([20260927, j]).standard_normal(n) for j in range(16)])
def omega(x, b):
    r = b - A @ x
    return float(np.max(np.abs(r) / (Aabs @ np.abs(x) + np.abs(b)))), float(np.max(np.abs(r)))
for tag, env in (("defaults", {}), ("HIPMF_LEAF_KERNELS=0", {"HIPMF_LEAF_KERNELS": "0"}), ("HIPMF_FUSED_SOLVE=0 (level-set kernels, single only)", {"HIPMF_FUSED_SOLVE": "0"})):
    os.environ.update(env)
    s = Hipmf()
    assert s.initialize(n, rp, ci, refinement_nstep=0) == 0 and s.factorize(v) == 0
    om1 = [omega(s.solve(B[j]), B[j]) for j in range(4)]
    print(tag, "| single:", " ".join("%.2e" % o[0] for o in om1), "| |r|:", "%.2e" % om1[0][1], flush=True)
    if "FUSED" not in tag:
        for nb in (16, 8):
            X = s.solve_many(B[:nb])
            om = [omega(X[j], B[j]) for j in range(4)]
            print(tag, "| block of %2d:" % nb, " ".join("%.2e" % o[0] for o in om), "| |r|:", "%.2e" % om[0][1], flush=True)
    s.close()
    for k in env: del os.environ[k]
