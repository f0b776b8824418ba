import sys, os, numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", ".."))
from russell_amd import problems as P
from russell_amd.backend import Hipmf
lib = sys.argv[1] if sys.argv[1] != "new" else None
n, rp, ci, v = P.poisson2d(1000)
b = P.csr_matvec(n, rp, ci, v, P.manufactured_solution(n))
s = Hipmf(lib)
assert s.initialize(n, rp, ci) == 0
for _ in range(30):
    assert s.factorize(v) == 0
x = s.solve(b)
print(lib, s.stats()["factor_ms"])
