#!/usr/bin/env python3
"""A circuit-like matrix through the benchmark harness: 2D grid (resistor mesh) + a few supply nets tied to 5 % of the nodes each
(dense rows / columns) + rows scaled over 8 decades, written as MatrixMarket and solved by russell_amd/lib/solve_matrix_market.
usage: circuit_like.py [NX [HUBS]]"""
import json, os, subprocess, sys, tempfile, time
import numpy as np
import scipy.sparse as sp
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from russell_amd import problems as P

nx = int(sys.argv[1]) if len(sys.argv) > 1 else 500
nh = int(sys.argv[2]) if len(sys.argv) > 2 else 5
n0, rp, ci, v = P.poisson2d(nx)
rng = np.random.default_rng(42)
rows = np.repeat(np.arange(n0), np.diff(rp))
ri, cj, va = [rows], [ci], [v]
for h in range(nh):
    idx = rng.choice(n0, n0 // 20, replace=False)
    ri += [np.full(idx.size, n0 + h), idx, np.array([n0 + h])]
    cj += [idx, np.full(idx.size, n0 + h), np.array([n0 + h])]
    va += [np.full(idx.size, -0.01), np.full(idx.size, -0.01), np.array([0.01 * idx.size + 1.0])]
n = n0 + nh
A = sp.coo_matrix((np.concatenate(va), (np.concatenate(ri), np.concatenate(cj))), shape=(n, n)).tocsr()
A = (sp.diags(10.0 ** rng.uniform(-4, 4, n)) @ A).tocoo()
with tempfile.TemporaryDirectory() as tmp:
    path = os.path.join(tmp, "circuit_like.mtx")
    with open(path, "w") as f:
        f.write("%%MatrixMarket matrix coordinate real general\n")
        f.write("%d %d %d\n" % (n, n, A.nnz))
        np.savetxt(f, np.column_stack([A.row + 1, A.col + 1, A.data]), fmt="%d %d %.17g")
    for extra, label in (([], "hubs ordered last (default)"), (None, None)):
        if extra is None:
            break
        t0 = time.perf_counter()
        p = subprocess.run([os.path.join(ROOT, "russell_amd", "lib", "solve_matrix_market"), "-r", "2", path], capture_output=True, text=True)
        d = json.loads(p.stdout)
        print("%s: n=%d nnz=%d | %s | read %s, initialize %s, factorize %s, solve %s | relative_error %.2e, effective matching %s, perturbed pivots %d"
              % (label, n, A.nnz, d["matrix"]["symmetric"], d["time_human"]["read_matrix"], d["time_human"]["initialize"], d["time_human"]["factorize"],
                 d["time_human"]["solve"], d["verify"]["relative_error"], d["output"]["effective_matching"], d["output"]["perturbed_pivots"]))
