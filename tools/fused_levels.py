#!/usr/bin/env python3
"""Profiling aid: time of the dependency-driven forward pass when it is cut after L levels (HIPMF_SF_FWD_LEVELS)."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from russell_amd import problems as P  # noqa: E402
from russell_amd.backend import Hipmf  # noqa: E402

grid = int(sys.argv[1]) if len(sys.argv) > 1 else 1000
n, rp, ci, v = P.poisson2d(grid)
b = P.csr_matvec(n, rp, ci, v, P.manufactured_solution(n))
for lev in [1, 2, 3, 4, 5, 6, 8, 10, 12, 14, 16, 18, 20, 22]:
    os.environ["HIPMF_SF_FWD_LEVELS"] = str(lev)
    s = Hipmf()
    assert s.initialize(n, rp, ci, refinement_nstep=0) == 0
    d_v, d_b, d_x = s.dev_alloc(v.nbytes), s.dev_alloc(b.nbytes), s.dev_alloc(b.nbytes)
    s.h2d(d_v, v), s.h2d(d_b, b)
    assert s.factorize_device(d_v) == 0
    for _ in range(3):
        s.solve_device(d_x, d_b)
    s.reset_timers()
    for _ in range(5):
        s.solve_device(d_x, d_b)
    st = s.stats()
    print("levels %2d of %d: fwd %.1f us" % (lev, st["nlevels"], 1e3 * st["acc_fwd_ms"] / st["acc_tri_count"]))
    s.close()
    if lev >= st["nlevels"]:
        break
