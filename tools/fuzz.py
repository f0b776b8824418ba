#!/usr/bin/env python3
"""Differential fuzzing against dense LAPACK: random sparse matrices of several kinds (dominant, weak diagonal with a hidden
permutation, hubs, disconnected blocks, symmetric-lower storage, many right-hand sides) at sizes up to ~700, solved through the
C-ABI; stops at the first mismatch and prints the seed.  usage: fuzz.py [CASES [SEED0]]"""
import os, sys
import numpy as np
import scipy.sparse as sp
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from russell_amd.backend import Hipmf

cases = int(sys.argv[1]) if len(sys.argv) > 1 else 200
seed0 = int(sys.argv[2]) if len(sys.argv) > 2 else 1000
ORDERING = int(os.environ.get("FUZZ_ORDERING", "0"))  # 0 nested dissection, 3 approximate minimum degree, 4 best of both (include/russell_hipmf.h)


def make(rng):
    n = int(rng.choice([1, 2, 5, 17, 32, 33, 64, 65, 100, 129, 200, 333, 500, 700]))
    kind = rng.choice(["dominant", "weak", "hubs", "blocks", "symlower", "band"])
    k = int(rng.uniform(1.0, 5.0) * n) + 1
    A = sp.coo_matrix((rng.uniform(-1, 1, k), (rng.integers(0, n, k), rng.integers(0, n, k))), shape=(n, n)).tolil()
    if kind == "band":
        A = sp.diags([rng.uniform(-1, 1, n - o) for o in (1, 2, 7) if o < n], [o for o in (1, 2, 7) if o < n], shape=(n, n)).tolil() if n > 1 else A
    if kind == "hubs" and n > 40:
        for h in range(2):
            idx = rng.choice(n, n // 2, replace=False)
            A[h, idx] = rng.uniform(-0.1, 0.1, idx.size)
            A[idx, h] = rng.uniform(-0.1, 0.1, (idx.size, 1))
    if kind == "blocks" and n > 3:
        h = n // 3
        A[:h, h:] = 0.0
        A[h:, :h] = 0.0
    if kind == "symlower":
        A = sp.tril(A + A.T).tolil()
    rowsum = np.asarray(abs(A.tocsr()).sum(axis=1)).ravel() + np.asarray(abs(A.tocsr()).sum(axis=0)).ravel()
    if kind == "weak":
        A.setdiag(0.0)
        perm = rng.permutation(n)
        for i in range(n):
            A[i, perm[i]] = (rowsum[i] + 1.0 + rng.random()) * (1 if rng.random() < 0.5 else -1)
    else:
        A.setdiag((rowsum + rng.uniform(0.2, 2.0, n)) * np.where(rng.random(n) < 0.3, -1.0, 1.0))
    A = A.tocsr()
    A.sort_indices()
    return kind, n, A


worst = 0.0
for c in range(cases):
    rng = np.random.default_rng(seed0 + c)
    kind, n, A = make(rng)
    sym = kind == "symlower"
    full = (A + sp.tril(A, -1).T).toarray() if sym else A.toarray()
    cond = np.linalg.cond(full)
    if not np.isfinite(cond) or cond > 1e10:
        continue
    rp, ci, v = A.indptr.astype(np.int32), A.indices.astype(np.int32), A.data.astype(np.float64)
    nr = int(rng.choice([1, 1, 3, 9, 17]))
    XS = rng.standard_normal((nr, n))
    B = XS @ full.T
    s = Hipmf()
    code = s.initialize(n, rp, ci, ordering=ORDERING, general_symmetric=sym, values=None if sym else v)
    assert code == 0, (seed0 + c, kind, n, "initialize", code)
    code = s.factorize(v, compute_determinant=True)
    assert code == 0, (seed0 + c, kind, n, "factorize", code)
    X = s.solve_many(B) if nr > 1 else s.solve(B[0])[None, :]
    want = np.linalg.solve(full, B.T).T
    err = float(np.max(np.abs(X - want)) / max(1.0, np.max(np.abs(want))))
    sign, logdet = np.linalg.slogdet(full)
    got = np.log10(abs(s.det_coefficient)) + s.det_exponent
    ok = err <= 1e-11 * max(1.0, cond) and abs(got - logdet / np.log(10.0)) < 1e-7 * max(1.0, abs(logdet)) and np.sign(s.det_coefficient) == sign
    worst = max(worst, err / max(1.0, cond))
    s.close()
    if not ok:
        print("MISMATCH seed %d kind %s n %d nrhs %d: err %.3e cond %.2e det got %.6f want %.6f sign %d/%d" % (seed0 + c, kind, n, nr, err, cond, got, logdet / np.log(10.0), int(np.sign(s.det_coefficient)), int(sign)))
        sys.exit(1)
print("%d cases ok (seeds %d..%d), worst error / cond = %.2e" % (cases, seed0, seed0 + cases - 1, worst))
