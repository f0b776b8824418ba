#!/usr/bin/env python3
"""Generates tests/golden/cases.json, tests/golden/bfwb62_x.json and tests/golden/mtx/*.mtx.

Run in the authoring container only (it reads /root/reference when present):

    python tests/golden/make_golden.py

Everything written is DATA: triplets, right-hand sides and expected solutions that the reference's
own tests hold for the solver boundary, plus the tiny MatrixMarket data
files of russell_sparse/data/matrix_market/.  Each case cites the reference lines it is taken
from.  Expected solutions are cross-checked here with a dense numpy solve before being written.
"""
import json
import os
import re
import shutil

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
REF = "/root/reference/russell_sparse"


def dense(n, trip, sym):
    a = np.zeros((n, n))
    for i, j, v in trip:
        a[i, j] += v
        if sym in ("YesLower", "YesUpper") and i != j:
            a[j, i] += v
    return a


CASES = []


def case(name, cite, n, sym, trip, rhs=None, x=None, tol=None, det=None, **extra):
    c = dict(name=name, cite=cite, n=n, sym=sym, triplets=[[int(i), int(j), float(v)] for i, j, v in trip])
    if rhs is not None:
        c["rhs"] = [float(v) for v in rhs]
        c["x"] = [float(v) for v in x]
        c["tol"] = tol
        a = dense(n, trip, sym)
        xs = np.linalg.solve(a, np.array(rhs, dtype=float))
        assert np.allclose(xs, x, rtol=1e-12, atol=1e-12), name
    if det is not None:
        c["det"] = det
        assert abs(np.linalg.det(dense(n, trip, sym)) - det) < 1e-10 * max(1.0, abs(det)), name
    c.update(extra)
    CASES.append(c)


# (1) UMFPACK quick-start 5x5, with the duplicated (0,0) triplet -- samples.rs:564-618;
#     rhs/x from solver_umfpack.rs:660-671 (tol 1e-14); det=114 from solver_umfpack.rs:585-606 (1e-13)
case(
    "umfpack_unsymmetric_5x5",
    "russell_sparse/src/samples.rs:564-618; solver_umfpack.rs:585-606,660-671",
    5, "No",
    [(0, 0, 1.0), (2, 1, -1.0), (1, 0, 3.0), (4, 1, 4.0), (4, 4, 1.0), (0, 1, 3.0), (3, 2, 1.0),
     (2, 2, -3.0), (0, 0, 1.0), (4, 2, 2.0), (2, 3, 2.0), (1, 4, 6.0), (1, 2, 4.0)],
    rhs=[8.0, 45.0, -3.0, 3.0, 19.0], x=[1.0, 2.0, 3.0, 4.0, 5.0], tol=1e-14, det=114.0,
    csc=dict(col_pointers=[0, 2, 5, 9, 10, 12], row_indices=[0, 1, 0, 2, 4, 1, 2, 3, 4, 2, 1, 4],
             values=[2.0, 3.0, 3.0, -1.0, 4.0, 4.0, -3.0, 1.0, 2.0, 2.0, 6.0, 1.0]),
    csr=dict(row_pointers=[0, 2, 5, 8, 9, 12], col_indices=[0, 1, 0, 2, 4, 1, 2, 3, 2, 1, 2, 4],
             values=[2.0, 3.0, 3.0, 4.0, 6.0, -1.0, -3.0, 2.0, 1.0, 4.0, 2.0, 1.0]),
)

# (2) symmetric 5x5, full storage -- samples.rs:1395-1449; solve_works_symmetric solver_umfpack.rs:689-716 (1e-10)
SYM5 = [(0, 0, 9.0), (0, 1, 1.5), (0, 2, 6.0), (0, 3, 0.75), (0, 4, 3.0), (1, 0, 1.5), (1, 1, 0.5), (2, 0, 6.0),
        (2, 2, 12.0), (3, 0, 0.75), (3, 3, 0.625), (4, 0, 3.0), (4, 4, 16.0)]
X_SYM5 = [-979.0 / 3.0, 983.0, 1961.0 / 12.0, 398.0, 123.0 / 2.0]
case("mkl_symmetric_5x5_full", "russell_sparse/src/samples.rs:1395-1449; solver_umfpack.rs:689-716",
     5, "YesFull", SYM5, rhs=[1, 2, 3, 4, 5], x=X_SYM5, tol=1e-10, det=9.0 / 4.0)

# (3) the same matrix, lower storage, flagged positive definite -- samples.rs:913-960; solver_cudss.rs:800-826 (1e-10)
case("mkl_positive_definite_5x5_lower", "russell_sparse/src/samples.rs:913-960; solver_cudss.rs:800-826",
     5, "YesLower",
     [(0, 0, 9.0), (1, 1, 0.5), (2, 2, 12.0), (3, 3, 0.625), (4, 4, 16.0), (1, 0, 1.5), (2, 0, 6.0), (3, 0, 0.75),
      (4, 0, 3.0)],
     rhs=[1, 2, 3, 4, 5], x=X_SYM5, tol=1e-10, positive_definite=True)

# (4) cuDSS SPD example, lower -- solver_cudss.rs:828-857 (1e-10)
case("cudss_simple_spd", "russell_sparse/src/solver_cudss.rs:828-857", 5, "YesLower",
     [(0, 0, 4.0), (1, 1, 3.0), (2, 0, 1.0), (2, 1, 2.0), (2, 2, 5.0), (3, 3, 1.0), (4, 2, 1.0), (4, 4, 2.0)],
     rhs=[7.0, 12.0, 25.0, 4.0, 13.0], x=[1, 2, 3, 4, 5], tol=1e-10, positive_definite=True)

# (5) cuDSS unsymmetric example -- solver_cudss.rs:859-891 (1e-10)
case("cudss_unsymmetric", "russell_sparse/src/solver_cudss.rs:859-891", 5, "No",
     [(0, 0, 5.0), (0, 1, 1.0), (0, 4, 3.0), (1, 0, 2.0), (1, 1, 6.0), (1, 3, 4.0), (2, 2, 7.0), (2, 3, 2.0),
      (3, 1, 1.0), (3, 2, 3.0), (3, 3, 8.0), (4, 0, 4.0), (4, 4, 9.0)],
     rhs=[22.0, 30.0, 29.0, 43.0, 49.0], x=[1, 2, 3, 4, 5], tol=1e-10)

# (6) 3x3 doc example -- lin_solver.rs:80-103 (1e-14)
case("doc_3x3", "russell_sparse/src/lin_solver.rs:80-103", 3, "No",
     [(0, 0, 0.2), (0, 1, 0.2), (1, 0, 0.5), (1, 1, -0.25), (2, 2, 0.25)],
     rhs=[1.0, 1.0, 1.0], x=[3.0, 2.0, 4.0], tol=1e-14)

# (7) 10x10 diagonal -- tests/test_umfpack.rs:6-30 (1e-14)
n = 10
d = n / 10.0
trip = [(k, k, 10.0 + k * d) for k in range(n)]
case("diag_10x10", "russell_sparse/tests/test_umfpack.rs:6-30", n, "No", trip,
     rhs=[(10.0 + k * d) * k for k in range(n)], x=[float(k) for k in range(n)], tol=1e-14)

# (8) singular 2x2 -- solver_umfpack.rs:624-630 ("Error(1): Matrix is singular")
CASES.append(dict(name="singular_2x2", cite="russell_sparse/src/solver_umfpack.rs:624-630", n=2, sym="No",
                  triplets=[[0, 0, 1.0], [1, 1, 0.0]], singular=True,
                  error="Error(1): Matrix is singular"))

# (9) Newton iteration table of tests/test_nonlinear_system.rs:63-110 (iterates @1e-6, exactly 5 iterations).
#     The Jacobian formula is restated in tests/test_reference_cases.py; the table below is the data.
CASES.append(dict(name="nonlinear_4eq", cite="russell_sparse/tests/test_nonlinear_system.rs:63-110", n=4, sym="No",
                  iterates=[[0.000000, 0.000000, 0.000000, 0.000000],
                            [-0.236393, -0.106230, -0.225574, -0.086557],
                            [-0.196773, -0.079071, -0.171604, -0.074904],
                            [-0.194395, -0.077412, -0.168376, -0.074249],
                            [-0.194386, -0.077406, -0.168364, -0.074246],
                            [-0.194386, -0.077406, -0.168364, -0.074246]],
                  tol=1e-6, iterations=5))

with open(os.path.join(HERE, "cases.json"), "w") as fh:
    json.dump(CASES, fh, indent=1)
print("wrote cases.json with", len(CASES), "cases")

# (10) MatrixMarket data files + the bfwb62 golden solution embedded at bin/solve_matrix_market.rs:307-372
if os.path.isdir(REF):
    os.makedirs(os.path.join(HERE, "mtx"), exist_ok=True)
    src_dir = os.path.join(REF, "data", "matrix_market")
    for name in sorted(os.listdir(src_dir)):
        if name.endswith(".mtx"):
            shutil.copyfile(os.path.join(src_dir, name), os.path.join(HERE, "mtx", name))
    src = open(os.path.join(REF, "src", "bin", "solve_matrix_market.rs")).read()
    tail = src[src.index("fn get_bfwb62_correct_x"):]
    xs = [float(v) for v in re.findall(r"(-?\d\.\d+e[+-]\d+)", tail)]
    assert len(xs) == 62
    with open(os.path.join(HERE, "bfwb62_x.json"), "w") as fh:
        json.dump(xs, fh)
    print("copied mtx fixtures and wrote bfwb62_x.json")
