#!/usr/bin/env python3
"""Config 4's matrix (200^3, L D L^T) with one rank's shard of 32 columns under a list of environment settings, one handle per setting in
ONE process: initialize / factorize / prepare / two timed blocked solves.  usage: config4_knobs.py [N] "A=1 B=2" "A=3" ..."""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from russell_amd import problems as P
from russell_amd.backend import Hipmf
args = sys.argv[1:]
N = int(args.pop(0)) if args and args[0].isdigit() else 200
settings = args or [""]
n, rp, ci, v = P.poisson3d(N)
lrp, lci, lv = P.lower_triangle(n, rp, ci, v)
nrhs = 32
B = np.stack([np.random.default_rng([20260927, j]).standard_normal(n) for j in range(nrhs)])
import scipy.sparse as sp
A = sp.csr_matrix((v, ci, rp), shape=(n, n))
for setting in settings:
    env = dict(kv.split("=") for kv in setting.split()) if setting else {}
    os.environ.update(env)
    s = Hipmf()
    assert s.initialize(n, lrp, lci, general_symmetric=True) == 0
    d_v, d_b, d_x = s.dev_alloc(lv.nbytes), s.dev_alloc(B.nbytes), s.dev_alloc(B.nbytes)
    s.h2d(d_v, lv), s.h2d(d_b, B)
    assert s.factorize_device(d_v) == 0
    s.prepare_solve_many(nrhs)
    ts = []
    for _ in range(3):
        s.lib.hipmf_device_synchronize()
        t0 = time.perf_counter()
        s.solve_device(d_x, d_b, nrhs, n)
        s.lib.hipmf_device_synchronize()
        ts.append(time.perf_counter() - t0)
    X = np.zeros((2, n)); s.d2h(X, d_x)
    err = max(float(np.max(np.abs(A @ X[j] - B[j]))) for j in range(2)) / (float(np.max(np.abs(v))) + 1.0)
    print("%-60s solves %s s  groups %d split_slabs %d fallbacks %d relative_error %.1e" % (setting or "(defaults)", ["%.4f" % t for t in ts], s.counter("block_groups"), s.counter("split_slabs"), s.counter("fused_fallbacks"), err), flush=True)
    for p in (d_v, d_b, d_x):
        s.dev_free(p)
    s.close()
    for k in env:
        del os.environ[k]
