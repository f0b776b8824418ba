#!/usr/bin/env python3
"""2D 5-point Poisson handed over as its lower triangle (L D L^T on the tiled fronts): factorize / solve times.  usage: sym2d.py [N]"""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from russell_amd import problems as P
from russell_amd.backend import Hipmf
N = int(sys.argv[1]) if len(sys.argv) > 1 else 1000
n, rp, ci, v = P.poisson2d(N)
xs = P.manufactured_solution(n)
b = P.csr_matvec(n, rp, ci, v, xs)
rp, ci, v = P.lower_triangle(n, rp, ci, v)
s = Hipmf()
assert s.initialize(n, rp, ci, general_symmetric=True) == 0
d_v, d_b, d_x = s.dev_alloc(v.nbytes), s.dev_alloc(b.nbytes), s.dev_alloc(b.nbytes)
s.h2d(d_v, v), s.h2d(d_b, b)
for rep in range(12):
    s.factorize_device(d_v); s.solve_device(d_x, d_b)
s.lib.hipmf_device_synchronize()
s.reset_timers()
for rep in range(20):
    s.factorize_device(d_v); s.solve_device(d_x, d_b)
s.lib.hipmf_device_synchronize()
st = s.stats()
x = np.zeros(n); s.d2h(x, d_x)
print("N=%d L D L^T: factor %.3f ms (last), max err %.2e" % (N, st["factor_ms"], np.max(np.abs(x - xs))))
s.close()
