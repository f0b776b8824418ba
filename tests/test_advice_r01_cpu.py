"""Regression tests for the round-1 advisor findings (host logic + emulated kernels; no GPU needed)."""
import ctypes as C
import os

import numpy as np
import pytest

from russell_amd import problems as P
from russell_amd import sparse as RS
from russell_amd.backend import Hipmf

HERE = os.path.dirname(os.path.abspath(__file__))


@pytest.fixture()
def host_on_emu(emu_lib):
    lib = RS._L()
    lib.rh_set_hipmf_library.argtypes = [C.c_char_p]
    lib.rh_set_hipmf_library(emu_lib.encode())
    yield lib
    lib.rh_set_hipmf_library(b"")


def _coo(n, rows, cols, vals):
    coo = RS.CooMatrix(n, n, len(vals))
    coo.put_many(rows.astype(np.int32), cols.astype(np.int32), vals.astype(np.float64))
    return coo


def test_repeat_factorize_with_reordered_triplets_uses_the_new_indices(host_on_emu):
    # host_api.cpp: the value map belongs to the first call's triplet order; the reference re-reads the indices every call
    n, rp, ci, v = P.poisson2d(9, 7)
    rows = np.repeat(np.arange(n), np.diff(rp))
    xs = P.manufactured_solution(n)
    solver = RS.LinSolver(RS.Genie.Hipmf)
    solver.actual.factorize(_coo(n, rows, ci, v))
    rng = np.random.default_rng(3)
    for _ in range(2):
        order = rng.permutation(len(v))
        v2 = v * (1.0 + 0.3 * rng.random(len(v)))
        coo = _coo(n, rows[order], ci[order], v2[order])
        solver.actual.factorize(coo)
        b = P.csr_matvec(n, rp, ci, v2, xs)
        x = solver.actual.solve(b)
        assert np.max(np.abs(x - xs)) < 1e-11
    # same nnz, other pattern: refused like any other structural change
    rows2 = rows.copy()
    cols2 = ci.copy()
    k = int(np.flatnonzero(rows != ci)[0])
    far = (rows2[k] + n // 2) % n
    if far in ci[rp[rows2[k]]:rp[rows2[k] + 1]]:
        far = (far + 1) % n
    cols2[k] = far
    with pytest.raises(RS.StrError, match="sparsity pattern differs"):
        solver.actual.factorize(_coo(n, rows2, cols2, v))


def test_csr_with_duplicates_or_unsorted_rows_is_refused(emu_lib):
    n, rp, ci, v = P.poisson2d(6, 5)
    s = Hipmf(emu_lib)
    ci_dup = ci.copy()
    ci_dup[rp[3] + 1] = ci_dup[rp[3]]  # duplicate column in row 3
    assert s.initialize(n, rp, ci_dup) == 600
    ci_uns = ci.copy()
    ci_uns[rp[3]], ci_uns[rp[3] + 1] = ci[rp[3] + 1], ci[rp[3]]
    assert s.initialize(n, rp, ci_uns) == 600
    assert s.initialize(n, rp, ci) == 0  # a refused initialize leaves the handle usable
    s.close()


def test_malformed_csr_is_refused_before_the_matching_reads_it(emu_lib):
    n, rp, ci, v = P.poisson2d(6, 5)
    v = v.copy()
    v[::5] = 0.0  # weak diagonal somewhere: the matching path would run
    bad_ci = ci.copy()
    bad_ci[7] = n + 1000
    s = Hipmf(emu_lib)
    assert s.initialize(n, rp, bad_ci, values=v) == 600
    bad_rp = rp.copy()
    bad_rp[4] = bad_rp[3] - 1
    assert s.initialize(n, bad_rp, ci, values=v) == 600
    bad_rp = rp.copy()
    bad_rp[0] = 1
    assert s.initialize(n, bad_rp, ci, values=v) == 600
    s.close()
    mrow = np.zeros(n, np.int32)
    dr = np.zeros(n)
    dc = np.zeros(n)
    assert s.lib.hipmf_max_product_matching(n, rp, bad_ci, v, mrow, dr, dc) == 600


def test_matrix_market_entry_on_the_wrong_triangle_is_an_error(tmp_path):
    # read_matrix_market.rs:450-463 unwraps the put result
    path = tmp_path / "upper_in_symmetric.mtx"
    path.write_text("%%MatrixMarket matrix coordinate real symmetric\n3 3 3\n1 1 2.0\n1 2 1.0\n3 3 4.0\n")
    with pytest.raises(RS.StrError):
        RS.read_matrix_market(str(path), RS.MMsym.LeaveAsLower)
    coo = RS.read_matrix_market(str(path), RS.MMsym.MakeItFull)
    assert coo.nnz == 4
