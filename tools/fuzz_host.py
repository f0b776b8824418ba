#!/usr/bin/env python3
"""Differential fuzzing of the layer ABOVE the C-ABI (host mirror of the reference's Rust API): real and complex COO matrices with
duplicate triplets, full and lower-triangular storage, first and repeated factorize (value refresh on the device), against
numpy.linalg.solve.  usage: fuzz_host.py [CASES [SEED0]]"""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from russell_amd.sparse import ComplexCooMatrix, ComplexLinSolver, CooMatrix, Genie, LinSolParams, LinSolver, Sym

cases = int(sys.argv[1]) if len(sys.argv) > 1 else 150
seed0 = int(sys.argv[2]) if len(sys.argv) > 2 else 300
worst = 0.0
for c in range(cases):
    rng = np.random.default_rng(seed0 + c)
    n = int(rng.choice([1, 3, 8, 33, 70, 150, 300]))
    cplx = rng.random() < 0.4
    lower = rng.random() < 0.3
    k = int(rng.uniform(1.5, 5.0) * n) + 1
    ii, jj = rng.integers(0, n, k), rng.integers(0, n, k)
    if lower:
        ii, jj = np.maximum(ii, jj), np.minimum(ii, jj)
    for rep_values in range(2):  # second pass: new values on the same structure through the same solver
        vals = rng.uniform(-1, 1, k) + (1j * rng.uniform(-1, 1, k) if cplx else 0.0)
        dense = np.zeros((n, n), dtype=complex if cplx else float)
        np.add.at(dense, (ii, jj), vals)
        if lower:
            dense = dense + np.tril(dense, -1).T
        diag = (np.abs(dense).sum(axis=1) + 1.0) * (1.0 + rng.random(n))
        dense[np.arange(n), np.arange(n)] += diag
        if rep_values == 0:
            sym = Sym.YesLower if lower else Sym.No
            coo = (ComplexCooMatrix if cplx else CooMatrix)(n, n, k + n, sym)
            solver = (ComplexLinSolver if cplx else LinSolver)(Genie.Hipmf)
        else:
            coo.reset()
        coo.put_many(ii, jj, vals)  # duplicates stay duplicates: summed by the conversion / the device value map
        coo.put_many(np.arange(n), np.arange(n), diag.astype(vals.dtype))
        xs = rng.standard_normal(n) + (1j * rng.standard_normal(n) if cplx else 0.0)
        b = dense @ xs
        solver.actual.factorize(coo, LinSolParams() if rep_values == 0 else None)
        x = solver.actual.solve(b)
        want = np.linalg.solve(dense, b)
        err = float(np.max(np.abs(x - want)) / max(1.0, np.max(np.abs(want))))
        worst = max(worst, err)
        if err > 1e-10:
            print("MISMATCH seed %d n %d complex %s lower %s pass %d: %.3e" % (seed0 + c, n, cplx, lower, rep_values, err))
            sys.exit(1)
print("%d cases x 2 factorizations ok (seeds %d..%d), worst error %.2e" % (cases, seed0, seed0 + cases - 1, worst))
