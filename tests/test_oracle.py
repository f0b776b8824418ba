"""The oracle against the reference's golden vectors.  CPU only."""
import json
import os

import numpy as np
import pytest

import oracle_lib as O

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
CASES = json.load(open(os.path.join(GOLD, "cases.json")))
BY_NAME = {c["name"]: c for c in CASES}


def trip(c):
    t = np.array(c["triplets"])
    return t[:, 0].astype(np.int32), t[:, 1].astype(np.int32), t[:, 2].astype(np.float64)


def test_coo_to_csc_and_csr_exact_arrays():
    # csc_matrix.rs:934-1018 / csr tests compare against Samples' exact arrays
    c = BY_NAME["umfpack_unsymmetric_5x5"]
    ai, aj, ax = trip(c)
    cp, ri, vx = O.coo_to_csc(5, 5, ai, aj, ax)
    assert cp.tolist() == c["csc"]["col_pointers"]
    assert ri.tolist() == c["csc"]["row_indices"]
    assert vx.tolist() == c["csc"]["values"]
    rp, cj, vy = O.coo_to_csr(5, 5, ai, aj, ax)
    assert rp.tolist() == c["csr"]["row_pointers"]
    assert cj.tolist() == c["csr"]["col_indices"]
    assert vy.tolist() == c["csr"]["values"]


@pytest.mark.parametrize("name", [c["name"] for c in CASES if "x" in c])
def test_solutions_match_reference_tolerances(name):
    c = BY_NAME[name]
    ai, aj, ax = trip(c)
    x, lu = O.solve_coo(c["n"], ai, aj, ax, np.array(c["rhs"], float), sym=c["sym"])
    assert lu.status == 0
    assert np.max(np.abs(x - np.array(c["x"]))) <= c["tol"] * max(1.0, np.max(np.abs(c["x"])))
    if "det" in c:
        m, e = lu.determinant()
        assert abs(m * 10.0 ** e - c["det"]) <= 1e-13 * abs(c["det"]) * 10


def test_singular_status():
    c = BY_NAME["singular_2x2"]
    ai, aj, ax = trip(c)
    _, lu = O.solve_coo(2, ai, aj, ax, np.ones(2))
    assert lu.status == 1  # "Error(1): Matrix is singular" (solver_umfpack.rs:492,624-630)


def read_mtx_lower(path):
    rows, cols, vals = [], [], []
    with open(path) as fh:
        header = fh.readline()
        assert header.startswith("%%MatrixMarket")
        dims = None
        for line in fh:
            s = line.strip()
            if not s or s.startswith("%"):
                continue
            if dims is None:
                dims = [int(v) for v in s.split()]
                continue
            a = s.split()
            rows.append(int(a[0]) - 1), cols.append(int(a[1]) - 1), vals.append(float(a[2]))
    return dims, rows, cols, vals


def test_bfwb62_golden_solution():
    # bin/solve_matrix_market.rs:217-229,307-372: rhs = ones, |x - x_correct| <= 1e-10
    dims, r, c, v = read_mtx_lower(os.path.join(GOLD, "mtx", "bfwb62.mtx"))
    n = dims[0]
    xg = np.array(json.load(open(os.path.join(GOLD, "bfwb62_x.json"))))
    x, lu = O.solve_coo(n, r, c, v, np.ones(n), sym="YesLower")
    assert lu.status == 0
    assert np.max(np.abs(x - xg)) <= 1e-10 * max(1.0, 1e-5 * np.max(np.abs(xg)))  # golden has 18 digits; |x|~1e5
    res = O.verify(n, r, c, v, x, np.ones(n), sym_triangular=True)
    assert res["relative_error"] < 1e-10


def test_matvec_variants_agree():
    rng = np.random.default_rng(1)
    n = 40
    ai = rng.integers(0, n, 300).astype(np.int32)
    aj = rng.integers(0, n, 300).astype(np.int32)
    ax = rng.standard_normal(300)
    u = rng.standard_normal(n)
    v0 = O.coo_matvec(n, ai, aj, ax, u)
    cp, ri, vx = O.coo_to_csc(n, n, ai, aj, ax)
    rp, cj, vy = O.coo_to_csr(n, n, ai, aj, ax)
    assert np.allclose(O.csc_matvec(n, n, cp, ri, vx, u), v0, atol=1e-12)
    assert np.allclose(O.csr_matvec(n, rp, cj, vy, u), v0, atol=1e-12)
    dense = np.zeros((n, n))
    np.add.at(dense, (ai, aj), ax)
    assert np.allclose(dense @ u, v0, atol=1e-12)


def test_lu_random_vs_dense_with_column_order():
    rng = np.random.default_rng(7)
    n = 60
    dense = np.where(rng.random((n, n)) < 0.08, rng.standard_normal((n, n)), 0.0) + np.diag(rng.standard_normal(n))
    ai, aj = np.nonzero(dense)
    ax = dense[ai, aj]
    cp, ri, vx = O.coo_to_csc(n, n, ai, aj, ax)
    b = rng.standard_normal(n)
    q = rng.permutation(n).astype(np.int32)
    for qq in (None, q):
        lu = O.OracleLU(n, cp, ri, vx, q=qq)
        x = lu.solve(b)
        assert np.allclose(x, np.linalg.solve(dense, b), rtol=1e-9, atol=1e-9)
        m, e = lu.determinant()
        sign, logdet = np.linalg.slogdet(dense)
        assert np.sign(m) == sign
        assert abs(np.log(abs(m)) + e * np.log(10.0) - logdet) < 1e-8
