#!/usr/bin/env python3
"""BASELINE config 4 on ONE GPU: 3D 7-point Poisson N^3 (default 200^3 = 8 M unknowns), handed over as its lower triangle
(general_symmetric -> L D L^T on the tiled fronts), NRHS right-hand sides (default 256) resident in HBM, blocked solves.

usage: python tools/config4_one_gpu.py [N [NRHS]]       (the 8-GPU split of the same job is bench.py --gpus 8 / many_rhs_multi_gpu.py)
"""
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from russell_amd import problems as P
from russell_amd.backend import Hipmf

N = int(sys.argv[1]) if len(sys.argv) > 1 else 200
NRHS = int(sys.argv[2]) if len(sys.argv) > 2 else 256
n, rp, ci, v = P.poisson3d(N)
rows = np.repeat(np.arange(n, dtype=np.int32), np.diff(rp))
keep = ci <= rows
rpl = np.concatenate([[0], np.cumsum(np.bincount(rows[keep], minlength=n))]).astype(np.int32)
cil, vl = ci[keep], v[keep]
del rows, keep
s = Hipmf(os.environ.get("HIPMF_LIB") or None)
t0 = time.perf_counter()
code = s.initialize(n, rpl, cil, general_symmetric=True)
t_init = time.perf_counter() - t0
if code != 0:
    print(s._err(code, "initialize"))
    sys.exit(1)
st = s.stats()
d_v = s.dev_alloc(vl.nbytes)
s.h2d(d_v, vl)
t0 = time.perf_counter()
code = s.factorize_device(d_v)
if code != 0:
    print(s._err(code, "factorize_device"))
    sys.exit(1)
s.lib.hipmf_device_synchronize()
t_fac = time.perf_counter() - t0
# column j = default_rng([20260927, j]).standard_normal(n): independent random right-hand sides
d_b, d_x = s.dev_alloc(8 * n * NRHS), s.dev_alloc(8 * n * NRHS)
if not d_b or not d_x:
    print("not enough device memory for %d right-hand sides beside the %.0f GB pool" % (NRHS, st["pool_bytes"] / 1e9))
    sys.exit(1)
col = np.empty(n)
t0 = time.perf_counter()
for j in range(NRHS):
    s.h2d(d_b + 8 * n * j, np.random.default_rng([20260927, j]).standard_normal(n))
t_h2d = time.perf_counter() - t0
t0 = time.perf_counter()
s.solve_device(d_x, d_b, NRHS, n)
s.lib.hipmf_device_synchronize()
t_solve = time.perf_counter() - t0
# VerifyLinSys' metric (verify_lin_sys.rs) of three columns against the full matrix on the host
err = 0.0
amax = float(np.max(np.abs(v)))
for j in (0, NRHS // 3, NRHS - 1):
    s.d2h(col, d_x + 8 * n * j)
    bj = np.random.default_rng([20260927, j]).standard_normal(n)
    err = max(err, float(np.max(np.abs(P.csr_matvec(n, rp, ci, v, col) - bj)) / (amax + 1.0)))
st = s.stats()
print(json.dumps({"workload": "3D 7-point Poisson %d^3 (n = %d), lower triangle (L D L^T), %d right-hand sides resident in HBM, one MI355X" % (N, n, NRHS),
                  "initialize_s": round(t_init, 2), "factorize_ms": round(t_fac * 1e3, 1), "solve_all_ms": round(t_solve * 1e3, 1),
                  "ms_per_rhs": round(t_solve * 1e3 / NRHS, 2), "rhs_per_s": round(NRHS / t_solve, 1), "h2d_rhs_s": round(t_h2d, 2),
                  "pool_gb": round(st["pool_bytes"] / 1e9, 1), "nnz_l": st["nnz_l"], "flops": st["flops"],
                  "lu_equivalent_tflops": round(st["flops"] / t_fac / 1e12, 1), "max_relative_error_3_columns": err,
                  "fused_fallbacks": s.stats().get("fused_fallbacks", -1), "refinement_steps_first_column": st["refinement_steps"]}))
s.close()
