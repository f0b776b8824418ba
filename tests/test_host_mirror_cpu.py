"""Host layer (C++ mirror of the Rust layer) against the oracle and the reference's fixtures.  CPU only.

Mirrors, with the reference's data and error strings:
  coo_matrix.rs tests (new/put errors), csc_matrix.rs:934-1018 / csr_matrix.rs (exact arrays of Samples),
  update_from_coo idempotence (csc_matrix.rs:1032), SpMV incl. triangular mirror (csc_matrix.rs:1197-1228),
  verify_lin_sys.rs:160-303, read_matrix_market.rs:479-889, lin_solver.rs dispatch errors."""
import os

import numpy as np
import pytest

import oracle_lib as O
from helpers import BY_NAME, GOLD, triplets
from russell_amd.sparse import (CooMatrix, CscMatrix, CsrMatrix, Genie, LinSolver, MMsym, StrError, Sym, VerifyLinSys,
                                handle_hipmf_error_code, read_matrix_market)

MTX = os.path.join(GOLD, "mtx")


def coo_from_case(c):
    ai, aj, ax = triplets(c)
    coo = CooMatrix(c["n"], c["n"], len(ax), Sym[c["sym"]])
    for i, j, v in zip(ai, aj, ax):
        coo.put(i, j, v)
    return coo


def test_coo_new_and_put_errors():
    for args, msg in [((0, 1, 1, Sym.No), "nrow must be ≥ 1"), ((1, 0, 1, Sym.No), "ncol must be ≥ 1"), ((1, 1, 0, Sym.No), "max_nnz must be ≥ 1"),
                      ((2, 3, 1, Sym.YesLower), "symmetric storage requires a square matrix")]:
        with pytest.raises(StrError, match=msg):
            CooMatrix(*args)
    coo = CooMatrix(2, 2, 1, Sym.YesLower)
    for (i, j), msg in [((2, 0), "COO matrix: index of row is outside range"), ((0, 2), "COO matrix: index of column is outside range"),
                        ((0, 1), "COO matrix: j > i is incorrect for lower triangular storage")]:
        with pytest.raises(StrError, match=msg):
            coo.put(i, j, 1.0)
    coo.put(1, 0, 1.0)
    with pytest.raises(StrError, match="COO matrix: max number of items has been reached"):
        coo.put(1, 1, 1.0)
    up = CooMatrix(2, 2, 2, Sym.YesUpper)
    with pytest.raises(StrError, match="COO matrix: j < i is incorrect for upper triangular storage"):
        up.put(1, 0, 1.0)


def test_csc_csr_from_coo_exact_arrays_and_update_idempotence():
    c = BY_NAME["umfpack_unsymmetric_5x5"]
    coo = coo_from_case(c)
    csc = CscMatrix.from_coo(coo)
    cp, ri, vx = csc.arrays()
    assert cp.tolist() == c["csc"]["col_pointers"] and ri.tolist() == c["csc"]["row_indices"] and vx.tolist() == c["csc"]["values"]
    csr = CsrMatrix.from_coo(coo)
    rp, cj, vy = csr.arrays()
    assert rp.tolist() == c["csr"]["row_pointers"] and cj.tolist() == c["csr"]["col_indices"] and vy.tolist() == c["csr"]["values"]
    for _ in range(2):  # csc_matrix.rs:1032: updating again gives the same arrays
        csc.update_from_coo(coo)
        csr.update_from_coo(coo)
    assert csc.arrays()[2].tolist() == c["csc"]["values"] and csr.arrays()[2].tolist() == c["csr"]["values"]
    # identical to the oracle's restatement
    ai, aj, ax = triplets(c)
    ocp, ori, ovx = O.coo_to_csc(5, 5, ai, aj, ax)
    assert np.array_equal(ocp, cp) and np.array_equal(ori, ri) and np.array_equal(ovx, vx)


def test_conversion_matches_oracle_on_random_duplicates():
    rng = np.random.default_rng(11)
    n, k = 57, 700
    ai, aj, ax = rng.integers(0, n, k), rng.integers(0, n, k), rng.standard_normal(k)
    coo = CooMatrix(n, n, k, Sym.No)
    for i, j, v in zip(ai, aj, ax):
        coo.put(i, j, v)
    for mine, theirs in ((CscMatrix.from_coo(coo).arrays(), O.coo_to_csc(n, n, ai, aj, ax)), (CsrMatrix.from_coo(coo).arrays(), O.coo_to_csr(n, n, ai, aj, ax))):
        for a, b in zip(mine, theirs):
            assert np.array_equal(a, b)  # bit-exact: duplicates are summed in COO order within a row


def test_mat_vec_mul_all_formats_incl_triangular_mirror():
    c = BY_NAME["mkl_positive_definite_5x5_lower"]
    coo = coo_from_case(c)
    u = np.array([1.0, -2.0, 3.0, 0.5, 7.0])
    ai, aj, ax = triplets(c)
    want = O.coo_matvec(5, ai, aj, ax, u, sym_triangular=True, alpha=2.0)
    assert np.allclose(coo.mat_vec_mul(u, alpha=2.0), want, atol=1e-15)
    assert np.allclose(CscMatrix.from_coo(coo).mat_vec_mul(u, 5, alpha=2.0), want, atol=1e-15)
    assert np.allclose(CsrMatrix.from_coo(coo).mat_vec_mul(u, 5, alpha=2.0), want, atol=1e-15)
    with pytest.raises(StrError, match="u.dim\\(\\) must be ≥ the number of columns of the matrix"):
        coo.mat_vec_mul(np.ones(3))


def test_verify_lin_sys():
    c = BY_NAME["umfpack_unsymmetric_5x5"]
    coo = coo_from_case(c)
    v = VerifyLinSys(coo, c["x"], c["rhs"])
    ai, aj, ax = triplets(c)
    o = O.verify(5, ai, aj, ax, np.array(c["x"]), np.array(c["rhs"]))
    assert v.max_abs_a == o["max_abs_a"] and v.max_abs_ax == o["max_abs_ax"] and v.relative_error == o["relative_error"] == 0.0
    with pytest.raises(StrError, match="x.dim\\(\\) must be equal to ncol"):
        VerifyLinSys(coo, np.ones(4), c["rhs"])
    with pytest.raises(StrError, match="rhs.dim\\(\\) must be equal to nrow"):
        VerifyLinSys(coo, c["x"], np.ones(4))


@pytest.mark.parametrize("name,msg", [
    ("__wrong__", "cannot open file"),
    ("bad_empty_file.mtx", "the file is empty"),
    ("bad_wrong_header.mtx", 'after %%MatrixMarket, the first option must be "matrix"'),
    ("bad_wrong_dims.mtx", "found invalid \\(zero or negative\\) dimensions"),
    ("bad_wrong_dims_complex.mtx", "found invalid \\(zero or negative\\) dimensions"),
    ("bad_missing_data.mtx", "not all values have been found"),
    ("bad_missing_data_complex.mtx", "not all values have been found"),
    ("bad_many_lines.mtx", "there are more values than specified"),
    ("bad_many_lines_complex.mtx", "there are more values than specified"),
    ("bad_symmetric_rectangular.mtx", "MatrixMarket data is invalid: the number of rows must equal the number of columns for symmetric matrices"),
    ("bad_symmetric_rectangular_complex.mtx", "MatrixMarket data is invalid: the number of rows must equal the number of columns for symmetric matrices"),
    ("bad_not_complex_hermitian.mtx", '"Hermitian" keyword can only be used with the "complex" type'),
])
def test_read_matrix_market_errors(name, msg):
    # read_matrix_market.rs:611-660
    with pytest.raises(StrError, match=msg):
        read_matrix_market(os.path.join(MTX, name), MMsym.LeaveAsLower)


def test_read_matrix_market_symmetric_handling():
    # read_matrix_market.rs:717-735 (ok_symmetric.mtx, LeaveAsLower) and the SwapToUpper / MakeItFull variants
    path = os.path.join(MTX, "ok_symmetric.mtx")
    coo = read_matrix_market(path, MMsym.LeaveAsLower)
    assert coo.symmetric == Sym.YesLower and (coo.nrow, coo.ncol, coo.nnz) == (5, 5, 15)
    ai, aj, ax = coo.triplets()
    assert ai.tolist() == [0, 1, 2, 3, 4, 1, 2, 3, 4, 2, 3, 4, 3, 4, 4]
    assert aj.tolist() == [0, 1, 2, 3, 4, 0, 0, 0, 0, 1, 1, 1, 2, 2, 3]
    assert ax.tolist() == [2.0, 2.0, 9.0, 7.0, 8.0, 1.0, 1.0, 3.0, 2.0, 2.0, 1.0, 1.0, 1.0, 5.0, 1.0]
    up = read_matrix_market(path, MMsym.SwapToUpper)
    assert up.symmetric == Sym.YesUpper and np.array_equal(up.triplets()[0], aj) and np.array_equal(up.triplets()[1], ai)
    full = read_matrix_market(path, MMsym.MakeItFull)
    assert full.symmetric == Sym.YesFull and full.nnz == 25
    gen = read_matrix_market(os.path.join(MTX, "ok_general.mtx"), MMsym.LeaveAsLower)
    assert gen.symmetric == Sym.No and (gen.nrow, gen.ncol) == (5, 5)


def test_genie_rules_and_unavailable_backends():
    assert Genie.Hipmf.to_string() == "hipmf" and Genie.from_name("UMFPACK") == Genie.Umfpack and Genie.from_name("other") == Genie.Hipmf
    assert Genie.Hipmf.get_sym(True) == Sym.YesLower and Genie.Umfpack.get_sym(True) == Sym.YesFull and Genie.Mumps.get_sym(False) == Sym.No
    for g, msg in [(Genie.Umfpack, "UMFPACK solver is not available"), (Genie.Mumps, "MUMPS solver is not available"), (Genie.Cudss, "cuDSS solver is not available")]:
        with pytest.raises(StrError, match=msg):
            LinSolver(g)
    assert handle_hipmf_error_code(1) == "Error(1): Matrix is singular"  # solver_umfpack.rs:492
    assert "requires initialization" in handle_hipmf_error_code(500000)
    assert "memory" in handle_hipmf_error_code(100)  # OOM strings must be recognisable (stats_lin_sol.rs:334-340)


def test_no_gpu_means_no_solver():
    from russell_amd import _capi
    if _capi.load().hipmf_device_count() > 0:
        pytest.skip("a GPU is visible here")
    with pytest.raises(StrError):
        LinSolver(Genie.Hipmf)


def test_csc_csr_transposition_round_trips():
    # csc_matrix.rs:1048-1100 (from_csr_works), csr_matrix.rs:1036-1090 (from_csc_works): the converted arrays are those of the
    # direct COO conversion, for every sample matrix (duplicates already summed, triangular storage kept as stored)
    from russell_amd.sparse import CscMatrix

    for c in GOLD:
        if "triplets" not in c or not c["triplets"]:
            continue
        coo = coo_from_case(c)
        csc, csr = CscMatrix.from_coo(coo), CsrMatrix.from_coo(coo)
        for got, want in ((CscMatrix.from_csr(csr), csc), (CsrMatrix.from_csc(csc), csr), (CsrMatrix.from_csc(CscMatrix.from_csr(csr)), csr)):
            for a, b in zip(got.arrays(), want.arrays()):
                assert np.array_equal(a, b), c["name"]
    # rectangular 1 x 2 (csc_matrix.rs:1086-1089: [10 20])
    coo = CooMatrix(1, 2, 2, Sym.No)
    coo.put(0, 0, 10.0)
    coo.put(0, 1, 20.0)
    cp, ri, vx = CscMatrix.from_csr(CsrMatrix.from_coo(coo)).arrays()
    assert cp.tolist() == [0, 1, 2] and ri.tolist() == [0, 0] and vx.tolist() == [10.0, 20.0]


def test_coo_update_transpose_assign_add():
    # coo_matrix.rs:590-627 (doc example of mat_vec_mul_update): lower-triangular storage, v = [1000, 2000, 3000] + A [1, 1, 1]
    coo = CooMatrix(3, 3, 6, Sym.No)
    for i, j, a in [(0, 0, 1.0), (1, 0, 2.0), (1, 1, 3.0), (2, 0, 4.0), (2, 1, 5.0), (2, 2, 6.0)]:
        coo.put(i, j, a)
    assert coo.as_dense().tolist() == [[1, 0, 0], [2, 3, 0], [4, 5, 6]]
    v = np.array([1000.0, 2000.0, 3000.0])
    coo.mat_vec_mul_update(v, np.ones(3))
    assert v.tolist() == [1001.0, 2005.0, 3015.0]
    assert coo.mat_t_vec_mul(np.ones(3)).tolist() == [7.0, 8.0, 6.0]
    assert coo.get_actual_nnz() == 6
    # triangular storage mirrors in every product and counts off-diagonals twice (coo_matrix.rs:1160-1170)
    low = CooMatrix(3, 3, 5, Sym.YesLower)
    for i, j, a in [(0, 0, 1.0), (1, 0, 2.0), (1, 1, 3.0), (2, 0, 4.0), (2, 2, 5.0)]:
        low.put(i, j, a)
    assert low.get_actual_nnz() == 7
    dense = low.to_dense()
    assert np.array_equal(dense, dense.T) and dense[0].tolist() == [1.0, 2.0, 4.0]
    u = np.array([1.0, -2.0, 0.5])
    assert np.allclose(low.mat_t_vec_mul(u), dense.T @ u, atol=0) and np.allclose(low.mat_vec_mul(u), dense @ u, atol=0)
    # assign / add: K = gamma M - J on triplets (how radau5.rs / euler_backward.rs build their matrices)
    mm, jj, kk = CooMatrix(2, 2, 2, Sym.No), CooMatrix(2, 2, 3, Sym.No), CooMatrix(2, 2, 5, Sym.No)
    mm.put(0, 0, 1.0), mm.put(1, 1, 1.0)
    jj.put(0, 0, 0.5), jj.put(0, 1, 2.0), jj.put(1, 0, -3.0)
    kk.assign(10.0, mm)
    kk.add(-1.0, jj)
    assert kk.nnz == 5 and kk.to_dense().tolist() == [[9.5, -2.0], [3.0, 10.0]]
    kk.assign(2.0, mm)  # assign resets the triplets first
    assert kk.nnz == 2 and kk.to_dense().tolist() == [[2.0, 0.0], [0.0, 2.0]]
    with pytest.raises(StrError, match="matrices must have the same nrow"):
        kk.assign(1.0, CooMatrix(3, 2, 1, Sym.No))
    with pytest.raises(StrError, match="matrices must have the same symmetric type"):
        kk.add(1.0, CooMatrix(2, 2, 1, Sym.YesLower))
    with pytest.raises(StrError, match="other.ncol must be ≤ this.ncol"):
        kk.add(1.0, CooMatrix(2, 3, 1, Sym.No))
    with pytest.raises(StrError, match="COO matrix: max number of items has been reached"):
        small = CooMatrix(2, 2, 1, Sym.No)
        small.add(1.0, jj)


def test_coo_put_lagrange_block():
    # coo_matrix.rs:823-857: [A B^T; B 0] for full storage, B only for lower, B^T only for upper
    bb = CooMatrix(1, 2, 2, Sym.No)
    bb.put(0, 0, 7.0), bb.put(0, 1, 8.0)
    for sym, want in ((Sym.No, [[1, 0, 7], [0, 2, 8], [7, 8, 0]]), (Sym.YesLower, [[1, 0, 7], [0, 2, 8], [7, 8, 0]])):
        aa = CooMatrix(3, 3, 6, sym)
        aa.put(0, 0, 1.0), aa.put(1, 1, 2.0)
        aa.put_lagrange_block(bb)
        assert aa.nnz == (6 if sym == Sym.No else 4)
        assert aa.to_dense().tolist() == want
    with pytest.raises(StrError, match="the Lagrange block must not be symmetric"):
        CooMatrix(3, 3, 6, Sym.No).put_lagrange_block(CooMatrix(1, 1, 1, Sym.YesLower))
    with pytest.raises(StrError, match="ncol\\(B\\) \\+ nrow\\(B\\) must be ≤ nrow\\(A\\)"):
        CooMatrix(2, 2, 6, Sym.No).put_lagrange_block(bb)


def test_coo_from_arrays():
    # coo_matrix.rs:246-291 and its tests (:990-1030): validation strings, nnz = max_nnz = len
    coo = CooMatrix.from_arrays(3, 3, [0, 1, 2, 0], [0, 1, 2, 0], [1.0, 2.0, 3.0, 0.5])
    assert coo.get_info() == (3, 3, 4, Sym.No)
    assert coo.to_dense().tolist() == [[1.5, 0, 0], [0, 2.0, 0], [0, 0, 3.0]]
    with pytest.raises(StrError, match="COO matrix: max number of items has been reached"):
        coo.put(1, 0, 1.0)
    for args, msg in [((0, 3, [0], [0], [1.0]), "nrow must be ≥ 1"), ((3, 0, [0], [0], [1.0]), "ncol must be ≥ 1"),
                      ((3, 3, [], [], []), "nnz must be ≥ 1"), ((3, 3, [0, 1], [0], [1.0, 2.0]), "col_indices.len\\(\\) must be = nnz"),
                      ((3, 3, [0, 1], [0, 1], [1.0]), "values.len\\(\\) must be = nnz"), ((3, 3, [3], [0], [1.0]), "row index is out-of-range"),
                      ((3, 3, [0], [-1], [1.0]), "col index is out-of-range")]:
        with pytest.raises(StrError, match=msg):
            CooMatrix.from_arrays(*args)


def test_compressed_constructors_validate_like_the_reference():
    # csc_matrix.rs:197-262 / csr_matrix.rs:193-257 and their tests (new_captures_errors): messages, order of the checks
    from russell_amd.sparse import CscMatrix

    good = CscMatrix.new(3, 3, [0, 2, 3, 5], [0, 2, 1, 0, 2], [1.0, 4.0, 3.0, 2.0, 6.0])
    assert good.to_dense().tolist() == [[1.0, 0, 2.0], [0, 3.0, 0], [4.0, 0, 6.0]]
    low = CsrMatrix.new(3, 3, [0, 1, 3, 4], [0, 0, 1, 2], [1.0, 2.0, 3.0, 5.0], Sym.YesLower)
    assert low.to_dense().tolist() == [[1.0, 2.0, 0], [2.0, 3.0, 0], [0, 0, 5.0]]
    assert np.array_equal(CsrMatrix.from_csc(good).to_dense(3, 3), good.to_dense())
    cases = [
        ((0, 3, [0], [], []), "nrow must be ≥ 1"),
        ((3, 0, [0], [], []), "ncol must be ≥ 1"),
        ((3, 3, [0, 1], [0], [1.0]), "col_pointers.len\\(\\) must be = ncol \\+ 1"),
        ((1, 1, [0, 0], [], []), "nnz = col_pointers\\[ncol\\] must be ≥ 1"),
        ((1, 1, [0, 1], [], [1.0]), "row_indices.len\\(\\) must be ≥ nnz"),
        ((1, 1, [0, 1], [0], []), "values.len\\(\\) must be ≥ nnz"),
        ((2, 2, [-1, 0, 1], [0], [1.0]), "col pointers must be ≥ 0"),
        ((2, 2, [2, 1, 1], [0, 0], [1.0, 1.0]), "col pointers must be sorted in ascending order"),
        ((2, 2, [0, 1, 2], [-1, 0], [1.0, 1.0]), "row indices must be ≥ 0"),
        ((2, 2, [0, 1, 2], [2, 0], [1.0, 1.0]), "row indices must be < nrow"),
        ((2, 2, [0, 2, 2], [1, 0], [1.0, 1.0]), "row indices must be sorted in ascending order \\(within their column\\)"),
    ]
    for args, msg in cases:
        with pytest.raises(StrError, match=msg):
            CscMatrix.new(*args)
    with pytest.raises(StrError, match="symmetric storage requires a square matrix"):
        CscMatrix.new(2, 1, [0, 1], [0], [1.0], Sym.YesFull)
    with pytest.raises(StrError, match="column indices must be sorted in ascending order \\(within their row\\)"):
        CsrMatrix.new(2, 2, [0, 2, 2], [1, 0], [1.0, 1.0])
    with pytest.raises(StrError, match="row_pointers.len\\(\\) must be = nrow \\+ 1"):
        CsrMatrix.new(3, 3, [0, 1], [0], [1.0])
