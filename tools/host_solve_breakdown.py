#!/usr/bin/env python3
"""Where a host-pointer solve of the 1M-DOF system spends its time: explicit H2D + device solve + D2H against solver_hipmf_solve(x, rhs)."""
import sys
import time

import numpy as np

sys.path.insert(0, __file__.rsplit("/", 2)[0])
from russell_amd import problems as P  # noqa: E402
from russell_amd.backend import Hipmf  # noqa: E402

n, rp, ci, v = P.poisson2d(1000)
b = np.random.default_rng(0).standard_normal(n)
x = np.zeros(n)
s = Hipmf()
assert s.initialize(n, rp, ci) == 0 and s.factorize(v) == 0
d_b, d_x = s.dev_alloc(8 * n), s.dev_alloc(8 * n)
for it in range(8):
    t0 = time.perf_counter()
    s.h2d(d_b, b)
    t1 = time.perf_counter()
    s.solve_device(d_x, d_b)
    s.lib.hipmf_device_synchronize()
    t2 = time.perf_counter()
    s.d2h(x, d_x)
    t3 = time.perf_counter()
    assert s.lib.solver_hipmf_solve(s.h, x, b, 0) == 0
    t4 = time.perf_counter()
    if it >= 2:
        print("h2d %.3f ms, device solve %.3f ms, d2h %.3f ms (sum %.3f) | host solve %.3f ms" % (1e3 * (t1 - t0), 1e3 * (t2 - t1), 1e3 * (t3 - t2), 1e3 * (t3 - t0), 1e3 * (t4 - t3)))
s.close()
