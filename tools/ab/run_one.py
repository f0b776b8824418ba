import sys, os, numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", ".."))
from russell_amd import problems as P
from russell_amd.backend import Hipmf
lib = sys.argv[1] if sys.argv[1] != "new" else None
if lib and "r01" in lib:
    from russell_amd import _capi
    for k in ("solver_hipmf_get_counter","solver_hipmf_factor_parts","hipmf_comm_unique_id","hipmf_comm_init_rank","hipmf_comm_destroy","solver_hipmf_broadcast_factor","solver_hipmf_solve_many_sharded"): _capi.SYMBOLS.pop(k, None)
n, rp, ci, v = P.poisson2d(1000)
b = P.csr_matvec(n, rp, ci, v, P.manufactured_solution(n))
s = Hipmf(lib)
assert s.initialize(n, rp, ci) == 0
for _ in range(30):
    assert s.factorize(v) == 0
x = s.solve(b)
print(lib, s.stats()["factor_ms"])
