"""Round 6 on the CPU (emulated kernels + host logic): block groups of the many-RHS solves, the buffers prepared ahead of a blocked solve,
the warning for a weak diagonal under a kept L D L^T plan, and the ADVICE r05 finding about a failed repeat factorize."""
import ctypes as C

import numpy as np
import pytest

from russell_amd import problems as P
from russell_amd import sparse as RS
from russell_amd.backend import Hipmf


def _many(emu, monkeypatch, groups, n, rp, ci, v, B, sym=False, prepare=False):
    monkeypatch.setenv("HIPMF_BLOCK_GROUPS", str(groups))
    s = Hipmf(emu)
    if sym:
        lrp, lci, lv = P.lower_triangle(n, rp, ci, v)
        assert s.initialize(n, lrp, lci, general_symmetric=True) == 0
        vals = lv
    else:
        assert s.initialize(n, rp, ci) == 0
        vals = v
    if prepare:
        s.prepare_solve_many(B.shape[0])  # (before the factorisation: a rank that waits for another rank's factor does this meanwhile)
    assert s.factorize(vals) == 0
    X = s.solve_many(B)
    info = (s.counter("block_groups"), s.counter("fused_fallbacks"), s.counter("split_slabs"))
    s.close()
    return X, info


@pytest.mark.parametrize("ncols", [33, 64, 70])
def test_block_groups_give_the_bits_of_one_block_per_launch(emu_lib, monkeypatch, ncols):
    # kernels_solve_fused.hpp, SfGroups: the groups of a launch are independent blocks of sixteen columns -- per column the arithmetic is
    # the one of a launch that carries its block alone
    n, rp, ci, v = P.poisson2d(38, 33)
    B = np.stack([np.random.default_rng([20260927, j]).standard_normal(n) for j in range(ncols)])
    X1, (g1, f1, _) = _many(emu_lib, monkeypatch, 1, n, rp, ci, v, B)
    X4, (g4, f4, _) = _many(emu_lib, monkeypatch, 4, n, rp, ci, v, B, prepare=True)
    assert g1 == 1 and g4 == min(4, (ncols + 15) // 16) and f1 == 0 and f4 == 0
    if ncols % 16 != 1:  # (a last block of ONE column takes the single-column kernels when it travels alone: equal to rounding then)
        assert np.array_equal(X1, X4)
    for j in range(ncols):
        r = P.csr_matvec(n, rp, ci, v, X4[j]) - B[j]
        assert np.max(np.abs(r)) / (np.max(np.abs(v)) + 1.0) <= 1e-13


def test_block_groups_with_split_dot_products_and_ldlt(emu_lib, monkeypatch):
    # a 3D factor whose top levels split the backward slabs' dot products (one scratch set per group), as L D L^T
    monkeypatch.setenv("HIPMF_SPLIT_TASKS", "4000")
    monkeypatch.setenv("HIPMF_SPLIT_MINLEN", "64")
    n, rp, ci, v = P.poisson3d(11)
    B = np.stack([np.random.default_rng([7, j]).standard_normal(n) for j in range(40)])
    X1, (g1, f1, sp1) = _many(emu_lib, monkeypatch, 1, n, rp, ci, v, B, sym=True)
    X3, (g3, f3, sp3) = _many(emu_lib, monkeypatch, 3, n, rp, ci, v, B, sym=True)
    assert g1 == 1 and g3 == 3 and f1 == 0 and f3 == 0 and sp1 == sp3 and sp3 > 0
    assert np.array_equal(X1, X3)
    for j in range(B.shape[0]):
        r = P.csr_matvec(n, rp, ci, v, X3[j]) - B[j]
        assert np.max(np.abs(r)) / (np.max(np.abs(v)) + 1.0) <= 1e-13


def test_prepare_solve_many_needs_an_initialised_handle(emu_lib):
    s = Hipmf(emu_lib)
    assert s.lib.solver_hipmf_prepare_solve_many(s.h, 32) == 500000  # ERROR_NEED_INITIALIZATION (constants.h:10)
    n, rp, ci, v = P.poisson2d(9, 8)
    assert s.initialize(n, rp, ci) == 0
    assert s.lib.solver_hipmf_prepare_solve_many(s.h, -1) != 0
    s.prepare_solve_many(1)  # no-op
    s.prepare_solve_many(20)
    assert s.counter("block_groups") == 2
    assert s.factorize(v) == 0
    x = s.solve(P.csr_matvec(n, rp, ci, v, np.ones(n)))
    assert np.max(np.abs(x - 1.0)) < 1e-12
    s.close()


def test_kept_ldlt_plan_reports_a_weak_diagonal_once(emu_lib):
    # with HIPMF_OPTION_SYM_RECHECK off, a symmetric-lower handle initialised WITHOUT values keeps its L D L^T plan on a
    # saddle-point matrix; the first factorize now says so (counter + last_error), nothing is re-analysed
    from test_sym_indefinite_cpu import _csr, saddle_point
    A, L = saddle_point(10, 20, seed=2)
    n = A.shape[0]
    rp, ci, v = _csr(L)
    s = Hipmf(emu_lib)
    assert s.initialize(n, rp, ci, general_symmetric=True) == 0
    assert s.counter("sym_weak_diagonal") == 0
    s.factorize(v)
    assert s.counter("sym_weak_diagonal") == 1 and s.counter("sym_expanded") == 0 and s.counter("rematch") == 0
    assert b"weak or zero diagonal" in s.lib.solver_hipmf_last_error(s.h)
    s.close()
    # a definite matrix: silent
    n, rp, ci, v = P.poisson2d(10, 9)
    lrp, lci, lv = P.lower_triangle(n, rp, ci, v)
    s = Hipmf(emu_lib)
    assert s.initialize(n, lrp, lci, general_symmetric=True) == 0
    assert s.factorize(lv) == 0 and s.counter("sym_weak_diagonal") == 0
    s.close()


@pytest.fixture()
def host_on_emu(emu_lib):
    lib = RS._L()
    lib.rh_set_hipmf_library.argtypes = [C.c_char_p]
    lib.rh_set_hipmf_library(emu_lib.encode())
    yield lib
    lib.rh_set_hipmf_library(b"")


def test_refused_repeat_factorize_leaves_no_usable_factor(host_on_emu):
    # ADVICE r05 (host_api.cpp): the repeat call factorises through the value map while a host thread compares the triplet order; when the
    # order AND the pattern changed, the call is refused -- and the handle must not keep the factor built from mis-mapped values
    n, rp, ci, v = P.poisson2d(9, 7)
    rows = np.repeat(np.arange(n), np.diff(rp))

    def coo(r, c, vals):
        m = RS.CooMatrix(n, n, len(vals))
        m.put_many(r.astype(np.int32), c.astype(np.int32), vals.astype(np.float64))
        return m

    solver = RS.LinSolver(RS.Genie.Hipmf)
    solver.actual.factorize(coo(rows, ci, v))
    b = P.csr_matvec(n, rp, ci, v, np.ones(n))
    assert np.max(np.abs(solver.actual.solve(b) - 1.0)) < 1e-12
    rows2, cols2 = rows.copy(), ci.copy()
    k = int(np.flatnonzero(rows != ci)[0])
    far = (rows2[k] + n // 2) % n
    if far in ci[rp[rows2[k]]:rp[rows2[k] + 1]]:
        far = (far + 1) % n
    cols2[k] = far
    with pytest.raises(RS.StrError, match="sparsity pattern differs"):
        solver.actual.factorize(coo(rows2, cols2, v))  # (map still set: the speculative factorisation ran on mis-mapped values)
    with pytest.raises(RS.StrError, match="factorize must be called before solve"):
        solver.actual.solve(b)
    # the handle recovers with a valid call
    solver.actual.factorize(coo(rows, ci, v))
    assert np.max(np.abs(solver.actual.solve(b) - 1.0)) < 1e-12


def _pm1(n, k, rng):
    # random +-1 entries, k per row plus a permutation: every transversal has the same product -- the maximum-product matching has nothing
    # to prefer, and elimination in exact +-1 arithmetic runs into EXACTLY zero pivots inside the pivot blocks of the static order
    import scipy.sparse as sp
    rows = np.repeat(np.arange(n), k)
    A = sp.csr_matrix((rng.choice([-1.0, 1.0], n * k), (rows, rng.integers(0, n, n * k))), shape=(n, n))
    A = A + sp.csr_matrix((rng.choice([-1.0, 1.0], n), (np.arange(n), rng.permutation(n))), shape=(n, n))
    A.sum_duplicates()
    A.eliminate_zeros()
    A.sort_indices()
    return A.tocsr()


@pytest.mark.parametrize("seed", [100, 101, 104])
def test_static_pivot_failures_are_rescued_not_reported_as_singular(emu_lib, monkeypatch, seed):
    # what the matching cannot fix.  UMFPACK pivots dynamically (interface_umfpack.c:167) and solves these; here the
    # zero pivots are replaced by sqrt(eps) max|a| (kernels_common.hpp, pivot_replacement), refinement + the Krylov rescue finish the solve,
    # and the probe solve of factorize tells "unlucky order" from "singular" (status 0, not UMFPACK's 1)
    import scipy.sparse.linalg as spla
    rng = np.random.default_rng(seed)
    A = _pm1(700 + 100 * (seed % 7), 4 + seed % 3, rng)
    n = A.shape[0]
    rp, ci, v = A.indptr.astype(np.int32), A.indices.astype(np.int32), A.data.astype(np.float64)
    xs = rng.standard_normal(n)
    b = A @ xs
    e_ref = np.max(np.abs(spla.splu(A.tocsc()).solve(b) - xs)) / np.max(np.abs(xs))
    s = Hipmf(emu_lib)
    assert s.initialize(n, rp, ci, values=v) == 0
    assert s.factorize(v) == 0
    assert s.num_perturbed > 0  # (the family does need replaced pivots: that is what the test is about)
    x = s.solve(b)
    assert np.max(np.abs(x - xs)) / np.max(np.abs(xs)) <= 10.0 * e_ref + 1e-12
    assert np.max(np.abs(A @ x - b)) / (np.max(np.abs(v)) + 1.0) <= 1e-10  # VerifyLinSys relative_error
    s.close()
    # without the rescue: the old verdict (status 1 for an exactly zero pivot)
    monkeypatch.setenv("HIPMF_KRYLOV", "0")
    s = Hipmf(emu_lib)
    assert s.initialize(n, rp, ci, values=v) == 0
    assert s.factorize(v) in (0, 1)
    s.close()


def test_singular_matrix_is_still_singular(emu_lib):
    # solver_umfpack.rs:624-630: Error(1) "Matrix is singular" -- the probe solve must not talk a singular matrix into status 0
    import scipy.sparse as sp
    for dense in (np.array([[1.0, 2.0], [2.0, 4.0]]), np.array([[1.0, 0.0, 2.0], [0.0, 0.0, 0.0], [3.0, 0.0, 1.0]])):
        A = sp.csr_matrix(dense)
        A = sp.csr_matrix((np.where(dense[dense != 0] != 0, dense[dense != 0], 0.0), A.indices, A.indptr), shape=A.shape) if False else A
        # (keep explicit zeros of the middle row out: give the row its diagonal entry with value 0)
        M = sp.lil_matrix(dense)
        r, c = np.nonzero(dense)
        rows, cols, vals = list(r), list(c), list(dense[r, c])
        for i in range(dense.shape[0]):
            if not np.any(dense[i]):
                rows.append(i), cols.append(i), vals.append(0.0)
        order = np.lexsort((cols, rows))
        rows, cols, vals = np.array(rows)[order], np.array(cols)[order], np.array(vals)[order]
        rp = np.concatenate([[0], np.cumsum(np.bincount(rows, minlength=dense.shape[0]))]).astype(np.int32)
        s = Hipmf(emu_lib)
        assert s.initialize(dense.shape[0], rp, cols.astype(np.int32)) == 0
        assert s.factorize(vals.astype(np.float64)) == 1
        s.close()
