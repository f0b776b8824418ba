#!/usr/bin/env python3
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ["HIPMF_SF_TRACE"] = sys.argv[1]
from russell_amd import problems as P
from russell_amd.backend import Hipmf
n, rp, ci, v = P.poisson2d(int(sys.argv[2]) if len(sys.argv) > 2 else 1000)
b = P.csr_matvec(n, rp, ci, v, P.manufactured_solution(n))
s = Hipmf()
assert s.initialize(n, rp, ci, refinement_nstep=0) == 0
assert s.factorize(v) == 0
for _ in range(3):
    x = s.solve(b)
s.close()
