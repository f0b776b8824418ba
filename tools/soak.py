#!/usr/bin/env python3
"""Soak: two host threads, one handle each (the Radau5 pattern: distinct handles used concurrently), ITERS x (factorize +
solve) on different matrices; every solution is checked, and the dependency-driven solves must never have fallen back to the
level-set schedule (solve_launches stays <= 6: wave-subtrees + mid + top per direction; no fallback counted).  usage: soak.py [ITERS]"""
import os, sys, threading, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from russell_amd import problems as P
from russell_amd.backend import Hipmf

iters = int(sys.argv[1]) if len(sys.argv) > 1 else 300
out = {}


def worker(name, prob):
    n, rp, ci, v = prob
    xs = P.manufactured_solution(n)
    s = Hipmf()
    assert s.initialize(n, rp, ci) == 0
    d_v, d_b, d_x = s.dev_alloc(v.nbytes), s.dev_alloc(8 * n), s.dev_alloc(8 * n)
    worst, launches = 0.0, 0
    rng = np.random.default_rng(len(name))
    x = np.zeros(n)
    for it in range(iters):
        vi = v * (1.0 + 0.01 * (it % 7))  # values change, structure fixed
        s.h2d(d_v, vi)
        s.h2d(d_b, P.csr_matvec(n, rp, ci, vi, xs))
        assert s.factorize_device(d_v) == 0
        s.solve_device(d_x, d_b)
        s.d2h(x, d_x)
        worst = max(worst, float(np.max(np.abs(x - xs))))
        launches = max(launches, s.stats()["solve_launches"])
    out[name] = (worst, launches, s.counter("fused_fallbacks") + s.counter("chain_fallbacks"))
    s.close()


t0 = time.perf_counter()
threads = [threading.Thread(target=worker, args=("poisson2d-400", P.poisson2d(400))),
           threading.Thread(target=worker, args=("poisson3d-36", P.poisson3d(36)))]
for t in threads:
    t.start()
for t in threads:
    t.join()
for k, (worst, launches, fb) in out.items():
    print("%s: %d iterations, worst max|x - x*| %.2e, solve launches per pass <= %d, fallbacks %d" % (k, iters, worst, launches, fb))
print("elapsed %.1f s" % (time.perf_counter() - t0))
assert all(l <= 6 and w < 1e-10 and fb == 0 for w, l, fb in out.values())
