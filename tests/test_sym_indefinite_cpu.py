"""Symmetric INDEFINITE input through the symmetric-lower boundary (general_symmetric = 1, lower triangle only): saddle-point / KKT
matrices as CooMatrix::put_lagrange_block builds them (reference: russell_sparse/src/coo_matrix.rs:823-857), with a ZERO (2,2) block.

The L D L^T fronts never interchange rows, so such a matrix used to rest on perturbed pivots + refinement.  With values at initialize
the C-ABI now mirrors it to general storage and takes the general path (maximum-product matching, LU): no perturbed pivots, and the
caller keeps handing over lower-triangle values (factorize, factorize_mapped).  Here on the CPU emulator; tests/test_round3_gpu.py
holds the device twin."""
import numpy as np
import pytest
import scipy.sparse as sp

from russell_amd.backend import Hipmf


def saddle_point(nx, m, seed=5, c22=0.0):
    """[[K, B^T], [B, -c22 I]]: K = 5-point Laplacian (nx x nx), B = m constraint rows of 3 entries each.  Returns full CSR + lower CSR."""
    rng = np.random.default_rng(seed)
    T = sp.diags([-1.0, 2.0, -1.0], [-1, 0, 1], shape=(nx, nx))
    K = sp.kron(sp.identity(nx), T) + sp.kron(T, sp.identity(nx))
    nk = nx * nx
    rows = np.repeat(np.arange(m), 3)
    cols = np.concatenate([rng.choice(nk, 3, replace=False) for _ in range(m)])
    B = sp.csr_matrix((rng.uniform(0.5, 1.5, 3 * m) * rng.choice([-1.0, 1.0], 3 * m), (rows, cols)), shape=(m, nk))
    A = sp.bmat([[K, B.T], [B, (-c22 * sp.identity(m)) if c22 else None]], format="csr")
    A.sum_duplicates()
    A.sort_indices()
    L = sp.tril(A, format="csr")
    L.sort_indices()
    return A, L


def _csr(M):
    return M.indptr.astype(np.int32), M.indices.astype(np.int32), M.data.astype(np.float64)


def test_saddle_point_lower_triangle_takes_the_matched_general_path(emu_lib):
    A, L = saddle_point(12, 30)
    n = A.shape[0]
    rp, ci, v = _csr(L)
    assert np.all(L.diagonal()[144:] == 0.0)  # the Lagrange block: no diagonal entries at all
    rng = np.random.default_rng(1)
    xs = rng.standard_normal(n)
    b = A @ xs
    s = Hipmf(emu_lib)
    assert s.initialize(n, rp, ci, general_symmetric=True, values=v) == 0
    assert s.counter("sym_expanded") == 1 and s.counter("symmetric_ldlt") == 0
    st = s.stats()
    assert st["matched"] == 1
    assert s.factorize(v) == 0
    assert s.num_perturbed == 0
    x = s.solve(b)
    assert np.max(np.abs(x - xs)) <= 1e-10 * np.max(np.abs(xs))
    # new values, same pattern (a Newton iteration): still lower-triangle values through the same entry point
    A2 = A.copy()
    A2.data = A.data * (1.0 + 0.2 * np.sin(np.arange(A.nnz)))
    A2 = ((A2 + A2.T) * 0.5).tocsr()
    A2.sort_indices()
    L2 = sp.tril(A2, format="csr")
    L2.sort_indices()
    assert np.array_equal(L2.indices, L.indices)
    assert s.factorize(L2.data) == 0
    x2 = s.solve(A2 @ xs)
    assert np.max(np.abs(x2 - xs)) <= 1e-10 * np.max(np.abs(xs))
    # mat_vec_mul multiplies by the SYMMETRIC matrix
    assert np.allclose(s.mat_vec_mul(xs), A2 @ xs, rtol=1e-13, atol=1e-13)
    s.close()


def test_saddle_point_value_map_speaks_of_the_lower_triangle(emu_lib):
    # COO triplets (lower triangle, with duplicates) -> the caller's map is built for ITS CSR; the handle composes it with the mirror
    A, L = saddle_point(10, 20, seed=9)
    n = A.shape[0]
    rp, ci, v = _csr(L)
    rng = np.random.default_rng(3)
    # every lower entry split into two triplets, shuffled
    parts = rng.uniform(0.2, 0.8, v.size)
    trip_val = np.concatenate([v * parts, v * (1.0 - parts)])
    trip_ent = np.concatenate([np.arange(v.size), np.arange(v.size)])
    order = rng.permutation(trip_val.size)
    trip_val, trip_ent = trip_val[order], trip_ent[order]
    seg_idx = np.argsort(trip_ent, kind="stable").astype(np.int32)
    seg_ptr = np.concatenate([[0], np.cumsum(np.bincount(trip_ent, minlength=v.size))]).astype(np.int32)
    xs = rng.standard_normal(n)
    s = Hipmf(emu_lib)
    assert s.initialize(n, rp, ci, general_symmetric=True, values=v) == 0
    assert s.counter("sym_expanded") == 1
    assert s.set_value_map(seg_ptr, seg_idx) == 0
    assert s.factorize_mapped(trip_val) == 0
    assert s.num_perturbed == 0
    x = s.solve(A @ xs)
    assert np.max(np.abs(x - xs)) <= 1e-10 * np.max(np.abs(xs))
    s.close()


def test_definite_and_valueless_symmetric_input_keep_ldlt(emu_lib, monkeypatch):
    A, L = saddle_point(10, 20, seed=2)
    n = A.shape[0]
    rp, ci, v = _csr(L)
    # no values at initialize: nothing to judge the diagonal by, L D L^T as before
    s = Hipmf(emu_lib)
    assert s.initialize(n, rp, ci, general_symmetric=True) == 0
    assert s.counter("sym_expanded") == 0
    s.close()
    # HIPMF_SYM_EXPAND=0 keeps L D L^T (static pivoting + refinement) for an indefinite matrix, too
    monkeypatch.setenv("HIPMF_SYM_EXPAND", "0")
    s = Hipmf(emu_lib)
    assert s.initialize(n, rp, ci, general_symmetric=True, values=v) == 0
    assert s.counter("sym_expanded") == 0
    s.close()
    monkeypatch.delenv("HIPMF_SYM_EXPAND")
    # a strong diagonal everywhere: stays symmetric
    T = sp.diags([-1.0, 2.0, -1.0], [-1, 0, 1], shape=(12, 12))
    K = (sp.kron(sp.identity(12), T) + sp.kron(T, sp.identity(12))).tocsr()
    Lk = sp.tril(K, format="csr")
    Lk.sort_indices()
    rpk, cik, vk = _csr(Lk)
    s = Hipmf(emu_lib)
    assert s.initialize(144, rpk, cik, general_symmetric=True, values=vk) == 0
    assert s.counter("sym_expanded") == 0
    assert s.factorize(vk) == 0
    xs = np.linspace(1.0, 2.0, 144)
    assert np.max(np.abs(s.solve(K @ xs) - xs)) < 1e-11
    s.close()


def test_host_mirror_put_lagrange_block_lower_storage(emu_lib):
    # the reference's own route to such a matrix: CooMatrix(Sym::YesLower) + put_lagrange_block (coo_matrix.rs:823-857), then
    # LinSolTrait::factorize twice (the second call refreshes the values through the triplet map on the device)
    import ctypes as C

    from russell_amd import sparse as RS

    lib = RS._L()
    lib.rh_set_hipmf_library.argtypes = [C.c_char_p]
    lib.rh_set_hipmf_library(emu_lib.encode())
    try:
        A, L = saddle_point(9, 15, seed=4)
        nk, m = 81, 15
        n = nk + m
        K = sp.tril(A[:nk, :nk], format="coo")
        B = A[nk:, :nk].tocoo()
        xs = np.cos(np.arange(n))
        for scale in (1.0, 3.0):  # second pass: new values, same triplets
            coo = RS.CooMatrix(n, n, K.nnz + B.nnz, RS.Sym.YesLower)
            coo.put_many(K.row.astype(np.int32), K.col.astype(np.int32), scale * K.data)
            bb = RS.CooMatrix(m, nk, B.nnz, RS.Sym.No)
            bb.put_many(B.row.astype(np.int32), B.col.astype(np.int32), B.data.astype(np.float64))
            coo.put_lagrange_block(bb)
            if scale == 1.0:
                solver = RS.LinSolver(RS.Genie.Hipmf)
            solver.actual.factorize(coo, None)
            As = sp.bmat([[scale * A[:nk, :nk], A[:nk, nk:]], [A[nk:, :nk], None]], format="csr")
            x = solver.actual.solve(As @ xs)
            assert np.max(np.abs(x - xs)) <= 1e-10
    finally:
        lib.rh_set_hipmf_library(b"")


def test_saddle_point_without_values_at_initialize_is_sent_to_the_general_path_by_the_first_factorize(emu_lib):
    # round 4: initialize(values = NULL) analyses the lower triangle for L D L^T; the first factorize sees the zero (2,2) block and redoes
    # the analysis on the mirrored matrix with the matching -- inside the call, once per handle
    A, L = saddle_point(12, 30)
    n = A.shape[0]
    rp, ci, v = _csr(L)
    rng = np.random.default_rng(1)
    xs = rng.standard_normal(n)
    # (opt-in since round 5, HIPMF_OPTION_SYM_RECHECK: by default the handle keeps the plan its peers hold -- ADVICE r04)
    s = Hipmf(emu_lib)
    assert s.initialize(n, rp, ci, general_symmetric=True) == 0
    s.factorize(v)  # (static pivoting on a zero (2,2) block: perturbed pivots or status 1 -- what the recheck is for)
    assert s.counter("sym_expanded") == 0 and s.counter("symmetric_ldlt") == 1
    s.close()
    s = Hipmf(emu_lib)
    assert s.set_option("sym_recheck", 1) == 0
    assert s.initialize(n, rp, ci, general_symmetric=True) == 0
    assert s.counter("sym_expanded") == 0 and s.counter("symmetric_ldlt") == 1
    assert s.factorize(v, compute_determinant=True) == 0
    assert s.counter("sym_expanded") == 1 and s.counter("symmetric_ldlt") == 0 and s.stats()["matched"] == 1
    assert s.num_perturbed == 0
    sign, logdet = np.linalg.slogdet(A.toarray())
    assert np.sign(s.det_coefficient) == sign and abs(np.log10(abs(s.det_coefficient)) + s.det_exponent - logdet / np.log(10.0)) < 1e-9
    x = s.solve(A @ xs)
    assert np.max(np.abs(x - xs)) <= 1e-10 * np.max(np.abs(xs))
    assert s.factorize(v * 1.5) == 0  # (the handle stays where it is: lower-triangle values as before)
    assert np.max(np.abs(s.solve(1.5 * (A @ xs)) - xs)) <= 1e-10 * np.max(np.abs(xs))
    s.close()
    # the device-values entry point takes the same decision (every factorize entry point or none)
    s = Hipmf(emu_lib)
    assert s.set_option("sym_recheck", 1) == 0
    assert s.initialize(n, rp, ci, general_symmetric=True) == 0
    d_v = s.dev_alloc(v.nbytes)
    s.h2d(d_v, v)
    assert s.factorize_device(d_v) == 0
    assert s.counter("sym_expanded") == 1 and s.stats()["matched"] == 1
    assert np.max(np.abs(s.solve(A @ xs) - xs)) <= 1e-10 * np.max(np.abs(xs))
    s.dev_free(d_v)
    s.close()
    # a definite matrix handed over the same way keeps its L D L^T fronts
    K = sp.tril(A[:144, :144], format="csr")
    K.sort_indices()
    s = Hipmf(emu_lib)
    assert s.set_option("sym_recheck", 1) == 0
    assert s.initialize(144, *_csr(K)[:2], general_symmetric=True) == 0
    assert s.factorize(_csr(K)[2]) == 0 and s.counter("sym_expanded") == 0 and s.counter("symmetric_ldlt") == 1
    s.close()
