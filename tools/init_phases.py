"""Phase times of solver_hipmf_initialize (verbose printout of the handle) for a Poisson matrix: python tools/init_phases.py [N] [2d|3d] [sym]."""
import os
import sys
import time

sys.path.insert(0, __file__.rsplit("/", 2)[0])
from russell_amd import problems as P  # noqa: E402
from russell_amd.backend import Hipmf  # noqa: E402

N = int(sys.argv[1]) if len(sys.argv) > 1 else 1000
kind = sys.argv[2] if len(sys.argv) > 2 else "2d"
n, rp, ci, v = P.poisson2d(N) if kind == "2d" else P.poisson3d(N)
sym = len(sys.argv) > 3
if sym:
    rp, ci, v = P.lower_triangle(n, rp, ci, v)
for rep in range(int(os.environ.get("INIT_REPS", "3"))):
    s = Hipmf()
    t = time.time()
    assert s.initialize(n, rp, ci, verbose=True, general_symmetric=sym) == 0
    print("initialize wall %.3f s" % (time.time() - t), flush=True)
    s.close()
