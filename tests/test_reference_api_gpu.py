"""The reference's own solver tests, replayed through the mirrored host API (LinSolver / Genie::Hipmf) on the GPU.

Each test cites the reference test it follows; data come from tests/golden (generated from those tests)."""
import json
import os

import numpy as np
import pytest

from helpers import BY_NAME, GOLD, triplets
from russell_amd import problems as P
from russell_amd.sparse import (CooMatrix, Genie, LinSolParams, LinSolver, MMsym, Ordering, Scaling, StrError, Sym, VerifyLinSys,
                                read_matrix_market)

pytestmark = pytest.mark.gpu


def coo_from_case(c, sym=None):
    ai, aj, ax = triplets(c)
    coo = CooMatrix(c["n"], c["n"], len(ax), Sym[c["sym"]] if sym is None else sym)
    for i, j, v in zip(ai, aj, ax):
        coo.put(i, j, v)
    return coo


def test_factorize_handles_errors():
    # solver_umfpack.rs:533-582 / solver_cudss.rs (same strings, this backend's symmetry rule)
    solver = LinSolver(Genie.Hipmf)
    with pytest.raises(StrError, match="the COO matrix must have at least one non-zero value"):
        solver.actual.factorize(CooMatrix(1, 1, 1, Sym.No), None)
    rect = CooMatrix(1, 7, 1, Sym.No)
    rect.put(0, 0, 1.0)
    with pytest.raises(StrError, match="the matrix must be square"):
        solver.actual.factorize(rect, None)
    full = CooMatrix(2, 2, 2, Sym.YesFull)
    full.put(0, 0, 1.0)
    with pytest.raises(StrError, match="HIPMF requires Sym::YesLower for symmetric matrices"):
        solver.actual.factorize(full, None)
    coo = CooMatrix(2, 2, 2, Sym.No)
    coo.put(0, 0, 1.0), coo.put(1, 1, 2.0)
    solver.actual.factorize(coo, None)
    other = CooMatrix(2, 2, 2, Sym.YesLower)
    other.put(0, 0, 1.0), other.put(1, 1, 2.0)
    with pytest.raises(StrError, match="subsequent factorizations must use the same matrix \\(symmetric differs\\)"):
        solver.actual.factorize(other, None)
    one = CooMatrix(1, 1, 1, Sym.No)
    one.put(0, 0, 1.0)
    with pytest.raises(StrError, match="subsequent factorizations must use the same matrix \\(ndim differs\\)"):
        solver.actual.factorize(one, None)
    fewer = CooMatrix(2, 2, 1, Sym.No)
    fewer.put(0, 0, 1.0)
    with pytest.raises(StrError, match="subsequent factorizations must use the same matrix \\(nnz differs\\)"):
        solver.actual.factorize(fewer, None)


def test_factorize_works_with_determinant_and_params_rule():
    # solver_umfpack.rs:585-621
    c = BY_NAME["umfpack_unsymmetric_5x5"]
    coo = coo_from_case(c)
    solver = LinSolver(Genie.Hipmf)
    params = LinSolParams()
    params.compute_determinant = True
    params.ordering = Ordering.Amd
    params.scaling = Scaling.Sum
    solver.actual.factorize(coo, params)
    out = solver.actual.outputs()
    det = out["determinant_coefficient"] * 10.0 ** out["determinant_exponent"]
    assert abs(det - 114.0) <= 1e-13 * 114.0 * 10
    solver.actual.factorize(coo, None)  # calling factorize again works
    params.ordering = Ordering.Metis
    with pytest.raises(StrError, match="subsequent factorizations must not change LinSolParams"):
        solver.actual.factorize(coo, params)


def test_factorize_fails_on_singular_matrix():
    # solver_umfpack.rs:624-630
    coo = CooMatrix(2, 2, 2, Sym.No)
    coo.put(0, 0, 1.0), coo.put(1, 1, 0.0)
    with pytest.raises(StrError, match="Error\\(1\\): Matrix is singular"):
        LinSolver(Genie.Hipmf).actual.factorize(coo, None)


def test_solve_handles_errors():
    # solver_umfpack.rs:633-657
    coo = CooMatrix(2, 2, 2, Sym.No)
    coo.put(0, 0, 123.0), coo.put(1, 1, 456.0)
    solver = LinSolver(Genie.Hipmf)
    with pytest.raises(StrError, match="the function factorize must be called before solve"):
        solver.actual.solve(np.zeros(2), x=np.zeros(2))
    solver.actual.factorize(coo, None)
    with pytest.raises(StrError, match="the dimension of the vector of unknown values x is incorrect"):
        solver.actual.solve(np.zeros(2), x=np.zeros(1))
    with pytest.raises(StrError, match="the dimension of the right-hand side vector is incorrect"):
        solver.actual.solve(np.zeros(1), x=np.zeros(2))


def test_solve_works_and_stats():
    # solver_umfpack.rs:660-686
    c = BY_NAME["umfpack_unsymmetric_5x5"]
    coo = coo_from_case(c)
    solver = LinSolver(Genie.Hipmf)
    solver.actual.factorize(coo, None)
    x = solver.actual.solve(c["rhs"])
    assert np.max(np.abs(x - np.array(c["x"]))) <= 1e-14 * 5
    x2 = solver.actual.solve(c["rhs"])  # calling solve again works
    assert np.array_equal(x, x2)
    st = solver.actual.stats(coo, "umfpack_unsymmetric_5x5", x, c["rhs"])
    assert st["main"]["solver"] == "HIPMF" and st["output"]["effective_scaling"] == "Sum"
    assert len(st["time_nanoseconds"]["initialize_array"]) == 1 and len(st["time_nanoseconds"]["solve_array"]) == 1
    assert solver.actual.get_ns_init() > 0 and solver.actual.get_ns_fact() > 0 and solver.actual.get_ns_solve() > 0
    assert st["verify"]["relative_error"] < 1e-14
    assert st["time_nanoseconds"]["total_ifs"] == st["time_nanoseconds"]["initialize"] + st["time_nanoseconds"]["factorize"] + st["time_nanoseconds"]["solve"]


def test_solve_works_symmetric_lower():
    # solver_umfpack.rs:689-716 / solver_cudss.rs:800-826 (1e-10), with this backend's YesLower storage rule
    c = BY_NAME["mkl_positive_definite_5x5_lower"]
    params = LinSolParams()
    params.positive_definite = True
    solver, x = LinSolver.compute(Genie.Hipmf, coo_from_case(c), c["rhs"], params)
    assert np.max(np.abs(x - np.array(c["x"]))) <= 1e-10 * max(1.0, np.max(np.abs(c["x"])))


def test_doc_example_and_diagonal():
    # lin_solver.rs:80-103 (1e-14) and tests/test_umfpack.rs:6-30 (1e-14)
    for name in ("doc_3x3", "diag_10x10", "cudss_unsymmetric", "cudss_simple_spd"):
        c = BY_NAME[name]
        _, x = LinSolver.compute(Genie.Hipmf, coo_from_case(c), c["rhs"])
        assert np.max(np.abs(x - np.array(c["x"]))) <= c["tol"] * max(1.0, np.max(np.abs(c["x"])))


def test_nonlinear_system_newton():
    # tests/test_nonlinear_system.rs:63-110: iterates @1e-6, exactly 5 iterations, re-factorise with None
    c = BY_NAME["nonlinear_4eq"]

    def residual(u):
        d1, d2, d3, d4 = u
        return np.array([2.0 * d1 + d1 ** 4 + d2 + 3.0 * d1 * d2 * d2 - 9.0 * d4 + d4 ** 4 - 0.2,
                         d1 + 3.0 * d1 * d1 * d2 + 10.0 * d2 + 4.0 * d2 * d2 + 2.0 * d2 * d3 - 8.0 * d3 + 7.0 * d4 + 0.1,
                         -8.0 * d2 + d2 * d2 + 3.0 * d3 + d3 * d3 + 2.0 * d4,
                         -9.0 * d1 + 4.0 * d1 * d4 ** 3 + 7.0 * d2 + 2.0 * d3 + 5.0 * d4 - 0.5])

    def jacobian(jj, u):
        d1, d2, d3, d4 = u
        jj.reset()
        rows = [[2.0 + 4.0 * d1 ** 3 + 3.0 * d2 * d2, 1.0 + 6.0 * d1 * d2, 0.0, -9.0 + 4.0 * d4 ** 3],
                [1.0 + 6.0 * d1 * d2, 10.0 + 3.0 * d1 * d1 + 8.0 * d2 + 2.0 * d3, -8.0 + 2.0 * d2, 7.0],
                [0.0, -8.0 + 2.0 * d2, 3.0 + 2.0 * d3, 2.0],
                [-9.0 + 4.0 * d4 ** 3, 7.0, 2.0, 5.0 + 12.0 * d1 * d4 * d4]]
        for i in range(4):
            for j in range(4):
                jj.put(i, j, rows[i][j])

    solver = LinSolver(Genie.Hipmf)
    jj = CooMatrix(4, 4, 16, Sym.No)
    uu = np.zeros(4)
    norm0, it = 1.0, 0
    while it < 10:
        rr = residual(uu)
        err = 1.0 if it == 0 else np.linalg.norm(rr) / norm0
        if it == 0:
            norm0 = np.linalg.norm(rr)
        assert np.max(np.abs(uu - np.array(c["iterates"][it]))) <= c["tol"]
        if err < 1e-13:
            break
        jacobian(jj, uu)
        solver.actual.factorize(jj, None)
        uu = uu - solver.actual.solve(rr)
        it += 1
    assert it == c["iterations"]


def test_bfwb62_from_matrix_market():
    # bin/solve_matrix_market.rs:97-231: read .mtx -> rhs = ones -> factorize -> solve -> verify; golden x @1e-10
    coo = read_matrix_market(os.path.join(GOLD, "mtx", "bfwb62.mtx"), MMsym.LeaveAsLower)
    assert coo.symmetric == Genie.Hipmf.get_sym(True)
    rhs = np.ones(coo.nrow)
    solver, x = LinSolver.compute(Genie.Hipmf, coo, rhs)
    xg = np.array(json.load(open(os.path.join(GOLD, "bfwb62_x.json"))))
    assert np.max(np.abs(x - xg)) <= 1e-10
    assert VerifyLinSys(coo, x, rhs).relative_error < 1e-10


def test_many_rhs_through_host_layer():
    c = BY_NAME["cudss_unsymmetric"]
    solver = LinSolver(Genie.Hipmf)
    solver.actual.factorize(coo_from_case(c), None)
    B = np.vstack([np.array(c["rhs"]), 2.0 * np.array(c["rhs"]), np.ones(5)])
    X = solver.actual.solve_many(B)
    assert np.max(np.abs(X[0] - np.array(c["x"]))) <= 1e-12 and np.max(np.abs(X[1] - 2.0 * np.array(c["x"]))) <= 1e-12


def test_repeat_factorize_refreshes_values_on_device():
    # the Radau5 / Newton pattern (radau5.rs:264-303): same structure, new values every step, factorize(coo, None).
    # After the first call the backend refreshes the CSR values on the device through a map built once
    # (solver_hipmf_set_value_map): triplets incl. duplicates, no host COO -> CSR conversion per step.
    npoint = 12
    n, rp, ci, v0 = P.brusselator_pattern(npoint)
    rows = np.repeat(np.arange(n), np.diff(rp))
    rng = np.random.default_rng(17)
    dup = rng.choice(len(v0), size=len(v0) // 10, replace=False)  # 10 % of the entries arrive as two triplets
    solver = LinSolver(Genie.Hipmf)
    coo = CooMatrix(n, n, len(v0) + len(dup), Sym.No)
    xs = 1.0 + (np.arange(n) % 7) / 7.0
    for step in range(6):
        v = v0 * (1.0 + 0.05 * step) + 0.3 * step * (rows == ci)
        coo.reset()
        isdup = np.zeros(len(v), bool)
        isdup[dup] = True
        for k in range(len(v)):
            if isdup[k]:
                coo.put(int(rows[k]), int(ci[k]), 0.25 * v[k])
                coo.put(int(rows[k]), int(ci[k]), 0.75 * v[k])
            else:
                coo.put(int(rows[k]), int(ci[k]), v[k])
        solver.actual.factorize(coo, None)
        b = P.csr_matvec(n, rp, ci, v, xs)
        x = solver.actual.solve(b)
        assert np.max(np.abs(x - xs)) / np.max(np.abs(xs)) < 1e-10, step


@pytest.mark.gpu
def test_solve_matrix_market_harness_on_device():
    # bin/solve_matrix_market.rs:97-305 against the real HIP library: bfwb62 golden solution (1e-10), JSON record, nrun
    import json
    import subprocess

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    harness = os.path.join(root, "russell_amd", "lib", "solve_matrix_market")
    env = {k: v for k, v in os.environ.items() if k != "RUSSELL_HIPMF_LIB"}
    p = subprocess.run([harness, "-d", "-r", "3", os.path.join(root, "tests", "golden", "mtx", "bfwb62.mtx")], env=env, capture_output=True,
                       text=True, timeout=300)
    assert p.returncode == 0, p.stderr
    assert "BFWB62 FAILED" not in p.stdout
    d = json.loads(p.stdout)
    assert d["matrix"]["nnz_actual"] == 342 and d["verify"]["relative_error"] < 1e-12
    assert len(d["time_nanoseconds"]["total_ifs_array"]) == 3
    p = subprocess.run([harness, os.path.join(root, "tests", "golden", "mtx", "ok_complex_general.mtx")], env=env, capture_output=True, text=True,
                       timeout=300)
    assert p.returncode == 0, p.stderr
    d = json.loads(p.stdout)
    assert d["matrix"]["complex"] is True and d["verify"]["relative_error"] < 1e-13
