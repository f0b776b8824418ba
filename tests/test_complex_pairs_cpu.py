"""Round 4: the complex twin's PAIRED mode on the CPU emulator (interface_complex_hipmf.cpp, NumericOptions.complex_pairs).  The
real-equivalent form of a complex matrix is factorised with pivot searches that take the two rows of a complex row together
(tile_lu32_z, lds_lu_blocked, tile_inv32: kernels_factor.hpp, kernels_factor_front.hpp), the ordering and the matching run on the
complex matrix's graph / moduli -- a complex LU with partial pivoting in real arithmetic whose complex pivots give the determinant
umfpack_zi_get_determinant hands to /root/reference/russell_sparse/c_code/interface_complex_umfpack.c:187-195 (the reference's own
check: complex_solver_umfpack.rs:520-545, det against a known value).  Oracle here: numpy's dense complex LU (slogdet, solve)."""
import ctypes as C
import os

import numpy as np
import pytest
import scipy.sparse as sp

from russell_amd._capi import load


def _handle(lib_path):
    lib = load(lib_path)
    h = lib.complex_solver_hipmf_new()
    assert h
    return lib, h


def _zcsr(A):
    A = sp.csr_matrix(A)
    A.sort_indices()
    return A.indptr.astype(np.int32), A.indices.astype(np.int32), np.ascontiguousarray(np.stack([A.data.real, A.data.imag], axis=1).ravel())


def _factor_det_solve(lib_path, A, env=None, symmetric=False, values_at_initialize=True, ordering=0):
    env = env or {}
    old = {k: os.environ.get(k) for k in env}
    os.environ.update(env)
    try:
        lib, h = _handle(lib_path)
        n = A.shape[0]
        S = sp.tril(A).tocsr() if symmetric else sp.csr_matrix(A)
        rp, ci, zv = _zcsr(S)
        assert lib.complex_solver_hipmf_initialize(h, ordering, 1, -1.0, -1, 0, int(symmetric), n, rp, ci, zv.ctypes.data if values_at_initialize else None) == 0
        npert, rc = C.c_int32(), C.c_double()
        dre, dim, dex = C.c_double(7.0), C.c_double(7.0), C.c_double(7.0)
        code = lib.complex_solver_hipmf_factorize(h, None, None, C.byref(npert), C.byref(rc), C.byref(dre), C.byref(dim), C.byref(dex), 1, 0, zv)
        rng = np.random.default_rng(3)
        xs = rng.standard_normal(n) + 1j * rng.standard_normal(n)
        b = sp.csr_matrix(A) @ xs
        x = np.zeros(2 * n)
        if code == 0:
            assert lib.complex_solver_hipmf_solve(h, x, np.ascontiguousarray(np.stack([b.real, b.imag], axis=1).ravel()), 0) == 0
        g = (C.c_double(), C.c_double(), C.c_double())
        code2 = lib.complex_solver_hipmf_get_determinant(h, C.byref(g[0]), C.byref(g[1]), C.byref(g[2]))
        lib.complex_solver_hipmf_drop(h)
        return code, complex(dre.value, dim.value), dex.value, x[0::2] + 1j * x[1::2], xs, npert.value, (code2, complex(g[0].value, g[1].value), g[2].value)
    finally:
        for k, val in old.items():
            if val is None:
                os.environ.pop(k, None)
            else:
                os.environ[k] = val


def _check_det(A, m, e):
    sign, logabs = np.linalg.slogdet(sp.csr_matrix(A).toarray())
    assert 1.0 <= abs(m) < 10.0
    assert abs(np.log10(abs(m)) + e - logabs / np.log(10.0)) < 1e-9
    assert abs(m / abs(m) - sign) < 1e-9, (m / abs(m), sign)


def _random_complex(n, density, seed, diag):
    rng = np.random.default_rng(seed)
    A = sp.random(n, n, density=density, random_state=seed, format="csr") * (1.0 + 0.0j)
    A = A + 1j * sp.random(n, n, density=density, random_state=seed + 1, format="csr")
    d = diag * np.exp(2j * np.pi * rng.random(n))  # diagonal entries of modulus `diag` and any phase (purely imaginary ones included)
    d[::7] = 1j * diag
    return sp.csr_matrix(A + sp.diags(d))


def _helmholtz2d(nx, ny, k2=3.0 + 0.7j):
    """5-point Laplacian minus a complex shift: the shape of Radau5's K_comp = (alpha + i beta) M - J on a 2D mesh (tiled fronts at the root)"""
    n = nx * ny
    T = lambda m: sp.diags([-np.ones(m - 1), 2.0 * np.ones(m), -np.ones(m - 1)], [-1, 0, 1])
    L = sp.kron(sp.identity(ny), T(nx)) + sp.kron(T(ny), sp.identity(nx))
    return sp.csr_matrix(L.astype(np.complex128) - k2 * sp.identity(n) + 0.3j * sp.diags(np.ones(n - 1), 1))


def test_reference_known_answer(emu_lib):
    # complex_solver_umfpack.rs:496-545 (complex_solver_umfpack_handle ... works): the 5 x 5 matrix of the reference's own test --
    # [2+1i 3+1i 0 0 0; 3-1i 0 4+2i 0 6+3i; 0 -1+1i -3-1i 2+2i 0; 0 0 1 0 0; 0 4 2 0 1+1i]; determinant checked against numpy here
    A = np.zeros((5, 5), dtype=complex)
    A[0, 0], A[0, 1] = 2 + 1j, 3 + 1j
    A[1, 0], A[1, 2], A[1, 4] = 3 - 1j, 4 + 2j, 6 + 3j
    A[2, 1], A[2, 2], A[2, 3] = -1 + 1j, -3 - 1j, 2 + 2j
    A[3, 2] = 1
    A[4, 1], A[4, 2], A[4, 4] = 4, 2, 1 + 1j
    code, m, e, x, xs, _, again = _factor_det_solve(emu_lib, sp.csr_matrix(A))
    assert code == 0
    _check_det(A, m, e)
    assert again == (0, m, e)
    assert np.max(np.abs(x - xs)) < 1e-12


@pytest.mark.parametrize("n,density,diag", [(40, 0.1, 4.0), (200, 0.03, 4.0), (300, 0.03, 0.05)])
def test_determinant_and_solution_of_random_complex_matrices(emu_lib, n, density, diag):
    # strong diagonals of any phase, and weak ones (0.05: the matching on the moduli permutes complex rows -- its sign enters the determinant;
    # the pivot searches interchange pairs inside the pivot blocks)
    A = _random_complex(n, density, seed=n + int(100 * diag), diag=diag)
    code, m, e, x, xs, npert, _ = _factor_det_solve(emu_lib, A)
    assert code == 0 and npert == 0
    _check_det(A, m, e)
    assert np.max(np.abs(x - xs)) < 1e-9 * max(1.0, np.max(np.abs(xs)))


def test_tiled_fronts_and_one_workgroup_fronts(emu_lib):
    # a 2D mesh large enough for tiled fronts (k_panel / k_update look-ahead / first tiles in the extend-add) and k_front_lu fronts
    A = _helmholtz2d(36, 34)
    ref = None
    for env in ({}, {"HIPMF_MID_FRONT": "0", "HIPMF_EA_LDS": "0"}, {"HIPMF_UPD32_MAXF": "0", "HIPMF_EA_LU": "0"}):
        code, m, e, x, xs, npert, _ = _factor_det_solve(emu_lib, A, env)
        assert code == 0 and npert == 0, env
        _check_det(A, m, e)
        assert np.max(np.abs(x - xs)) < 1e-10, env
        if ref is None:
            ref = (m, e)
        assert e == ref[1] and abs(m - ref[0]) < 1e-10, env


def test_complex_symmetric_lower_storage_and_no_values_at_initialize(emu_lib):
    A = _helmholtz2d(18, 17)
    A = sp.csr_matrix((A + A.T) * 0.5)
    for kw in ({"symmetric": True}, {"values_at_initialize": False}, {"ordering": 2}):
        code, m, e, x, xs, _, _ = _factor_det_solve(emu_lib, A, **kw)
        assert code == 0, kw
        _check_det(A, m, e)
        assert np.max(np.abs(x - xs)) < 1e-11, kw


def test_singular_complex_matrix_has_zero_determinant(emu_lib):
    A = _random_complex(30, 0.15, seed=5, diag=3.0).tolil()
    A[7, :] = 0.0
    A[:, 7] = 0.0
    # (row and column 7 are empty: no perfect matching, a zero pivot is met -- UMFPACK's "matrix is singular" case, solver_umfpack.rs:492)
    code, m, e, *_ = _factor_det_solve(emu_lib, sp.csr_matrix(A))
    assert code != 0 or (m == 0 and e == 0)


def test_plain_real_equivalent_mode_has_no_determinant(emu_lib):
    A = _random_complex(40, 0.1, seed=9, diag=4.0)
    code, m, e, *_ = _factor_det_solve(emu_lib, A, {"HIPMF_COMPLEX_PAIRS": "0"})
    assert code == 400000  # ERROR_NOT_AVAILABLE


def _two_complex_leaves_and_a_root(p, m, seed):
    """Natural order: two leaf supernodes of p complex columns with WEAK diagonals (the pivot search has to interchange complex rows inside
    the pivot block), both coupled densely to the same m later rows / columns, then the m x m remainder."""
    rng = np.random.default_rng(seed)
    n = 2 * p + m
    cz = lambda *s: rng.standard_normal(s) + 1j * rng.standard_normal(s)
    A = np.zeros((n, n), dtype=complex)
    for g in range(2):
        a0 = g * p
        A[a0:a0 + p, a0:a0 + p] = cz(p, p) * 0.3 + 0.05 * np.eye(p)
        A[a0:a0 + p, 2 * p:] = cz(p, m) * 0.3
        A[2 * p:, a0:a0 + p] = cz(m, p) * 0.3
    A[2 * p:, 2 * p:] = np.diag((4.0 + rng.random(m)) * np.exp(2j * np.pi * rng.random(m))) + np.diag(cz(m - 1) * 0.2, 1) + np.diag(cz(m - 1) * 0.2, -1)
    return sp.csr_matrix(A)


@pytest.mark.parametrize("p,m", [(3, 20), (8, 30), (16, 40), (16, 90), (40, 60)])
def test_interchanges_of_complex_rows_inside_the_pivot_blocks(emu_lib, p, m):
    # real-equivalent fronts of 2 p pivots and 2 m rows: k_small_factor (f <= 64), k_front_lu (2 p <= 32), the tiled path (2 p > 32: several
    # panel steps, look-ahead tiles); no matching, natural order -- the interchanges happen in the kernels' paired pivot searches
    A = _two_complex_leaves_and_a_root(p, m, seed=10 * p + m)
    for env in ({"HIPMF_MATCHING": "0"}, {"HIPMF_MATCHING": "0", "HIPMF_MID_FRONT": "0"}):
        code, mant, e, x, xs, npert, _ = _factor_det_solve(emu_lib, A, env, ordering=2)
        assert code == 0 and npert == 0, env
        _check_det(A, mant, e)
        assert np.max(np.abs(x - xs)) < 1e-9 * max(1.0, np.max(np.abs(xs))), env
