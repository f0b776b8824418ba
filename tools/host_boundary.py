"""Repeat call through the mirror of LinSolTrait with HOST pointers (the `host_api` section of bench.py alone): python tools/host_boundary.py [N]."""
import sys
import time

import numpy as np

sys.path.insert(0, __file__.rsplit("/", 2)[0])
from russell_amd import problems as P  # noqa: E402
from russell_amd import sparse as RS  # noqa: E402

N = int(sys.argv[1]) if len(sys.argv) > 1 else 1000
n, rp, ci, v = P.poisson2d(N)
xs = P.manufactured_solution(n)
b = P.csr_matvec(n, rp, ci, v, xs)
rows = np.repeat(np.arange(n, dtype=np.int32), np.diff(rp))
coo = RS.CooMatrix(n, n, len(v))
coo.put_many(rows, ci.astype(np.int32), v)
hs = RS.LinSolver(RS.Genie.Hipmf)
hs.actual.factorize(coo)
x = np.zeros(n)  # (the caller's x, reused from call to call as russell's solvers do)
for it in range(9):
    t0 = time.perf_counter()
    hs.actual.factorize(coo)
    t1 = time.perf_counter()
    hs.actual.solve(b, x=x)
    t2 = time.perf_counter()
    print("call %d: factorize %.3f ms, solve %.3f ms, error %.1e" % (it, 1e3 * (t1 - t0), 1e3 * (t2 - t1), np.max(np.abs(x - xs))), flush=True)
