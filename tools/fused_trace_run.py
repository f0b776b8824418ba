#!/usr/bin/env python3
"""usage: fused_trace_run.py TRACE_FILE [GRID [NRHS]]  -- writes the HIPMF_SF_TRACE stamps of the last solve (NRHS columns in one block)"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ["HIPMF_SF_TRACE"] = sys.argv[1]
from russell_amd import problems as P
from russell_amd.backend import Hipmf
N = int(sys.argv[2]) if len(sys.argv) > 2 else 1000
n, rp, ci, v = P.poisson3d(N) if os.environ.get("TRACE_3D") else P.poisson2d(N)  # (TRACE_3D=1: the N^3 7-point matrix)
b = P.csr_matvec(n, rp, ci, v, P.manufactured_solution(n))
s = Hipmf()
if os.environ.get("TRACE_SYM"):  # (TRACE_SYM=1: handed over as its lower triangle -> L D L^T fronts)
    lrp, lci, lv = P.lower_triangle(n, rp, ci, v)
    assert s.initialize(n, lrp, lci, refinement_nstep=0, general_symmetric=True) == 0
    assert s.factorize(lv) == 0
else:
    assert s.initialize(n, rp, ci, refinement_nstep=0) == 0
    assert s.factorize(v) == 0
nrhs = int(sys.argv[3]) if len(sys.argv) > 3 else 1
import numpy as np
for _ in range(3):
    x = s.solve(b) if nrhs == 1 else s.solve_many(np.tile(b[None, :], (nrhs, 1)))
s.close()
