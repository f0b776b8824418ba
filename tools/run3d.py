#!/usr/bin/env python3
"""3D 7-point Poisson on one GPU: phase times, accuracy, factor statistics (scaling study towards BASELINE config 4)."""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from russell_amd import problems as P
from russell_amd.backend import Hipmf
N = int(sys.argv[1]) if len(sys.argv) > 1 else 48
mode = sys.argv[2] if len(sys.argv) > 2 else "lu"  # "lu": general storage; "sym": lower triangle -> L D L^T on the tiled fronts
n, rp, ci, v = P.poisson3d(N)
xs = P.manufactured_solution(n)
b = P.csr_matvec(n, rp, ci, v, xs)
if mode == "sym":
    rows = np.repeat(np.arange(n, dtype=np.int32), np.diff(rp))
    keep = ci <= rows
    rp = np.concatenate([[0], np.cumsum(np.bincount(rows[keep], minlength=n))]).astype(np.int32)
    ci, v = ci[keep], v[keep]
s = Hipmf()
t0 = time.perf_counter(); code = s.initialize(n, rp, ci, general_symmetric=(mode == "sym")); t1 = time.perf_counter()
print("N=%d n=%d nnz=%d initialize code %d in %.2f s" % (N, n, rp[-1], code, t1 - t0), flush=True)
if code != 0:
    print(s._err(code, "initialize"))  # (out of memory: the message states the bytes the fronts need and the bytes free)
    sys.exit(1)
st = s.stats()
print({k: st[k] for k in ("nsuper", "nlevels", "max_front", "nnz_l", "flops", "pool_bytes")}, flush=True)
d_v, d_b, d_x = s.dev_alloc(v.nbytes), s.dev_alloc(b.nbytes), s.dev_alloc(b.nbytes)
s.h2d(d_v, v), s.h2d(d_b, b)
for rep in range(2):
    t0 = time.perf_counter(); code = s.factorize_device(d_v); s.lib.hipmf_device_synchronize(); t1 = time.perf_counter()
    s.solve_device(d_x, d_b); s.lib.hipmf_device_synchronize(); t2 = time.perf_counter()
    print("rep %d: factorize code %d %.1f ms, solve %.2f ms" % (rep, code, (t1 - t0) * 1e3, (t2 - t1) * 1e3), flush=True)
x = np.zeros(n); s.d2h(x, d_x)
st = s.stats()
print("max err %.2e; flops/s %.2f TF; sptrsv pair %.3f ms; refinement steps %d" % (np.max(np.abs(x - xs)), st["flops"] / (st["factor_ms"] * 1e-3) / 1e12, st["fwd_ms"] + st["bwd_ms"], st["refinement_steps"]))
s.close()
