#!/usr/bin/env python3
"""Differential fuzzing of the tiled path against scipy's SuperLU: 2D / 3D stencil patterns of random size with random
unsymmetric values, random row scaling and random schedule knobs (panels per pass, chain links, small-front split, solve
lanes), one or several right-hand sides.  usage: fuzz_big.py [CASES [SEED0]]"""
import os, sys
import numpy as np
import scipy.sparse as sp
import scipy.sparse.linalg as spla
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from russell_amd import problems as P
from russell_amd.backend import Hipmf

cases = int(sys.argv[1]) if len(sys.argv) > 1 else 40
seed0 = int(sys.argv[2]) if len(sys.argv) > 2 else 100
worst = 0.0
for c in range(cases):
    rng = np.random.default_rng(seed0 + c)
    if rng.random() < 0.5:
        n, rp, ci, v = P.poisson2d(int(rng.integers(20, 160)), int(rng.integers(20, 160)))
    else:
        n, rp, ci, v = P.poisson3d(int(rng.integers(5, 26)), int(rng.integers(5, 26)), int(rng.integers(5, 26)))
    rows = np.repeat(np.arange(n), np.diff(rp))
    v = v * (1.0 + 0.6 * rng.uniform(-1, 1, v.size))
    v[rows == ci] *= 1.0 + rng.random(n)
    v *= (10.0 ** rng.uniform(-3, 3, n))[rows]
    knobs = {"HIPMF_UPD_G4": str(rng.choice([65, 128, 2048])), "HIPMF_UPD_G8": str(rng.choice([200, 400, 4096])),
             "HIPMF_SPLIT_PIVOTS": str(rng.choice([64, 96, 4096])), "HIPMF_SMALL_SPLIT": str(rng.choice([0, 20, 28])),
             "HIPMF_SOLVE_LANES": str(rng.choice([1, 2, 3]))}
    os.environ.update(knobs)
    A = sp.csr_matrix((v, ci, rp), shape=(n, n))
    sym = rng.random() < 0.4
    if sym:
        # symmetric (positive definite or with a few negative pivots) and handed over as the lower triangle: L D L^T on the tiled fronts
        A = sp.csr_matrix((np.abs(v), ci, rp), shape=(n, n))
        A = ((A + A.T) * 0.5).tocsr()
        d = np.asarray(abs(A).sum(axis=1)).ravel() * (1.0 + rng.random(n))
        if rng.random() < 0.5:
            d[rng.choice(n, max(1, n // 50), replace=False)] *= -1.0   # indefinite: a few negative diagonal entries
        A = (A - sp.diags(A.diagonal()) + sp.diags(d)).tocsr()
        A.sort_indices()
        L = sp.tril(A).tocsr()
        L.sort_indices()
        rp, ci, v = L.indptr.astype(np.int32), L.indices.astype(np.int32), L.data.astype(np.float64)
    nr = int(rng.choice([1, 1, 5, 19]))
    XS = rng.standard_normal((nr, n))
    B = (A @ XS.T).T
    s = Hipmf()
    assert s.initialize(n, rp, ci, ordering=int(os.environ.get("FUZZ_ORDERING", "0")), general_symmetric=sym, values=v if (rng.random() < 0.5 and not sym) else None) == 0
    code = s.factorize(v)
    assert code == 0, (seed0 + c, code)
    X = s.solve_many(B) if nr > 1 else s.solve(B[0])[None, :]
    want = spla.splu(A.tocsc()).solve(B.T).T
    r = (A @ X.T).T - B
    scaled = float(np.max(np.abs(r) / (abs(A) @ np.abs(X.T)).T.clip(1e-300)))
    err = float(np.max(np.abs(X - want)) / np.max(np.abs(want)))
    st = s.stats()
    s.close()
    worst = max(worst, scaled)
    if not (scaled < 1e-13 and err < 1e-6):
        print("MISMATCH seed %d n %d nrhs %d knobs %s: componentwise backward error %.3e, difference to SuperLU %.3e, max front %d" % (seed0 + c, n, nr, knobs, scaled, err, st["max_front"]))
        sys.exit(1)
print("%d cases ok (seeds %d..%d), worst componentwise backward error %.2e" % (cases, seed0, seed0 + cases - 1, worst))
