#!/usr/bin/env python3
"""initialize of the N^3 7-point matrix handed over as its lower triangle (BASELINE config 4's shape), wall clock + the library's phase line.
usage: python tools/init_3d_lower.py [N]   (HIPMF_ND_PAR_BFS: team breadth-first searches from that region size on)"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from russell_amd import problems as P  # noqa: E402
from russell_amd.backend import Hipmf  # noqa: E402

N = int(sys.argv[1]) if len(sys.argv) > 1 else 200
n, rp, ci, v = P.poisson3d(N)
lrp, lci, lv = P.lower_triangle(n, rp, ci, v)
os.environ["HIPMF_VERBOSE_INIT"] = "1"
s = Hipmf()
t0 = time.perf_counter()
assert s.initialize(n, lrp, lci, general_symmetric=True, verbose=True) == 0
print("initialize wall %.2f s, plan digest %x" % (time.perf_counter() - t0, s.counter("plan_digest") & 0xffffffffffffffff))
s.close()
