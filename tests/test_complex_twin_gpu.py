"""Complex twin (ComplexLinSolTrait for Genie::Hipmf): the reference's complex known-answer tests, replayed through the
host mirror.  The backend solves the real-equivalent system of order 2n on the same device path."""
import numpy as np
import pytest

from russell_amd.sparse import ComplexCooMatrix, ComplexLinSolver, Genie, LinSolParams, StrError, Sym

pytestmark = pytest.mark.gpu


def complex_symmetric_3x3(sym):
    # samples.rs:220-370 (complex_symmetric_3x3_lower / _full): entries incl. the duplicated (1,0)
    if sym == Sym.YesLower:
        coo = ComplexCooMatrix(3, 3, 6, Sym.YesLower)
        for (i, j, a) in [(1, 0, -0.5 - 0.5j), (0, 0, 2 + 1j), (2, 2, 2 - 1j), (1, 0, -0.5 - 0.5j), (1, 1, 2 + 2j), (2, 1, -1 + 1j)]:
            coo.put(i, j, a)
    else:
        coo = ComplexCooMatrix(3, 3, 8, Sym.No)
        for (i, j, a) in [(1, 0, -0.5 - 0.5j), (0, 0, 2 + 1j), (2, 2, 2 - 1j), (1, 0, -0.5 - 0.5j), (1, 1, 2 + 2j), (2, 1, -1 + 1j),
                          (0, 1, -1 - 1j), (1, 2, -1 + 1j)]:
            coo.put(i, j, a)
    return coo


@pytest.mark.parametrize("sym", [Sym.No, Sym.YesLower])
def test_solve_works(sym):
    # complex_solver_umfpack.rs:598-610: x = [1+1i, 2-2i, 3+3i] @1e-14, solve twice
    solver = ComplexLinSolver(Genie.Hipmf)
    coo = complex_symmetric_3x3(sym)
    rhs = np.array([-3 + 3j, 2 - 2j, 9 + 7j])
    solver.actual.factorize(coo, None)
    for _ in range(2):
        x = solver.actual.solve(rhs)
        assert np.max(np.abs(x - np.array([1 + 1j, 2 - 2j, 3 + 3j]))) <= 1e-14
    # calling factorize again works (complex_solver_umfpack.rs:542-543)
    solver.actual.factorize(coo, None)
    assert np.max(np.abs(coo.mat_vec_mul(x) - rhs)) <= 1e-13


def test_complex_diagonal_10x10():
    # tests/test_complex_umfpack.rs:5-30: a_kk = (10 + k d) + (10 - k d) i, x_k = k + 0.5 i, @1e-14
    n, d = 10, 1.0
    coo = ComplexCooMatrix(n, n, n, Sym.No)
    xc = np.array([k + 0.5j for k in range(n)])
    rhs = np.zeros(n, complex)
    for k in range(n):
        akk = (10.0 + k * d) + (10.0 - k * d) * 1j
        coo.put(k, k, akk)
        rhs[k] = akk * xc[k]
    solver = ComplexLinSolver(Genie.Hipmf)
    solver.actual.factorize(coo, None)
    assert np.max(np.abs(solver.actual.solve(rhs) - xc)) <= 1e-14


def test_errors_follow_the_reference():
    # complex_solver_umfpack.rs:473-521, 571-597
    solver = ComplexLinSolver(Genie.Hipmf)
    rect = ComplexCooMatrix(4, 3, 1, Sym.No)
    rect.put(0, 0, 1.0)
    with pytest.raises(StrError, match="the matrix must be square"):
        solver.actual.factorize(rect, None)
    with pytest.raises(StrError, match="the COO matrix must have at least one non-zero value"):
        solver.actual.factorize(ComplexCooMatrix(1, 1, 1, Sym.No), None)
    full = ComplexCooMatrix(2, 2, 2, Sym.YesFull)
    full.put(0, 0, 1.0), full.put(1, 1, 2.0)
    with pytest.raises(StrError, match="HIPMF requires Sym::YesLower for symmetric matrices"):
        solver.actual.factorize(full, None)
    with pytest.raises(StrError, match="the function factorize must be called before solve"):
        solver.actual.solve(np.zeros(2, complex))
    coo = ComplexCooMatrix(2, 2, 2, Sym.No)
    coo.put(0, 0, 123.0 + 1j), coo.put(1, 1, 456.0 + 2j)
    solver.actual.factorize(coo, None)
    with pytest.raises(StrError, match="the dimension of the vector of unknown values x is incorrect"):
        solver.actual.solve(np.zeros(2, complex), x=np.zeros(1, complex))
    with pytest.raises(StrError, match="the dimension of the right-hand side vector is incorrect"):
        solver.actual.solve(np.zeros(1, complex), x=np.zeros(2, complex))
    one = ComplexCooMatrix(1, 1, 1, Sym.No)
    one.put(0, 0, 1.0)
    with pytest.raises(StrError, match="ndim differs"):
        solver.actual.factorize(one, None)
    with pytest.raises(StrError, match="must not change LinSolParams"):
        solver.actual.factorize(coo, LinSolParams())


def test_radau5_like_complex_shifted_system():
    # (alpha + beta i) M - J with J a real Jacobian pattern (radau5.rs:264-266): random sparse J, M = I
    import scipy.sparse as sp
    n = 400
    rng = np.random.default_rng(9)
    J = (sp.random(n, n, density=5.0 / n, random_state=3, format="coo") - sp.diags(2.0 + rng.random(n))).tocoo()
    alpha, beta = 3.6, 3.05
    coo = ComplexCooMatrix(n, n, J.nnz + n, Sym.No)
    for i, j, a in zip(J.row, J.col, J.data):
        coo.put(int(i), int(j), -a)
    for k in range(n):
        coo.put(k, k, alpha + beta * 1j)
    zs = rng.standard_normal(n) + 1j * rng.standard_normal(n)
    A = (alpha + beta * 1j) * sp.identity(n) - J.tocsr()
    rhs = A @ zs
    solver = ComplexLinSolver(Genie.Hipmf)
    solver.actual.factorize(coo, None)
    z = solver.actual.solve(rhs)
    assert np.max(np.abs(z - zs)) / np.max(np.abs(zs)) < 1e-12


def _check_complex_det(A_dense, m, e, tol=1e-9):
    sign, logabs = np.linalg.slogdet(A_dense)
    assert 1.0 <= abs(m) < 10.0
    assert abs(np.log10(abs(m)) + e - logabs / np.log(10.0)) < tol
    assert abs(m / abs(m) - sign) < 10.0 * tol, (m / abs(m), sign)


def test_determinant_works():
    # complex_solver_umfpack.rs:523-545: compute_determinant = true on the reference's 5 x 5 sample, det = mantissa x 10^exponent
    # (round 4: the paired pivot searches of the real-equivalent factorisation leave the complex pivots; oracle: numpy's dense complex LU)
    A = np.zeros((5, 5), dtype=complex)
    A[0, 0], A[0, 1] = 2 + 1j, 3 + 1j
    A[1, 0], A[1, 2], A[1, 4] = 3 - 1j, 4 + 2j, 6 + 3j
    A[2, 1], A[2, 2], A[2, 3] = -1 + 1j, -3 - 1j, 2 + 2j
    A[3, 2] = 1
    A[4, 1], A[4, 2], A[4, 4] = 4, 2, 1 + 1j
    coo = ComplexCooMatrix(5, 5, 13, Sym.No)
    for i, j in zip(*np.nonzero(A)):
        coo.put(int(i), int(j), A[i, j])
    par = LinSolParams()
    par.compute_determinant = True
    solver = ComplexLinSolver(Genie.Hipmf)
    solver.actual.factorize(coo, par)
    out = solver.actual.outputs()
    _check_complex_det(A, out["determinant_coefficient"], out["determinant_exponent"])
    # calling factorize again works, values through the device-side refresh (the determinant is fetched after it)
    solver.actual.factorize(coo, None)
    out2 = solver.actual.outputs()
    assert out2["determinant_coefficient"] == out["determinant_coefficient"] and out2["determinant_exponent"] == out["determinant_exponent"]
    # without the request the triple is zero (interface_complex_umfpack.c:196-200)
    plain = ComplexLinSolver(Genie.Hipmf)
    plain.actual.factorize(coo, None)
    o = plain.actual.outputs()
    assert o["determinant_coefficient"] == 0 and o["determinant_exponent"] == 0


@pytest.mark.parametrize("case", ["helmholtz 60x50", "random weak diagonal 600", "symmetric lower helmholtz 30x28", "weak pivot blocks, natural order"])
def test_determinant_of_larger_complex_matrices(case):
    # tiled fronts, one-workgroup fronts, small fronts; matching on the moduli (weak diagonal); interchanges of complex rows inside the
    # pivot blocks (natural order, no matching) -- through the C-ABI
    import ctypes as C
    import os
    import scipy.sparse as sp
    from russell_amd._capi import load
    from test_complex_pairs_cpu import _helmholtz2d, _random_complex, _two_complex_leaves_and_a_root, _zcsr
    symmetric, ordering, env, tol = False, 0, {}, 1e-9
    if case.startswith("helmholtz"):
        A = _helmholtz2d(60, 50)
    elif case.startswith("random"):
        A = _random_complex(600, 0.01, seed=77, diag=0.005)  # (below 1 % of the rows' largest entries: the matching on the moduli is applied)
    elif case.startswith("symmetric"):
        A = _helmholtz2d(30, 28)
        A = sp.csr_matrix((A + A.T) * 0.5)
        symmetric = True
    else:
        A = _two_complex_leaves_and_a_root(40, 60, seed=460)
        ordering, env = 2, {"HIPMF_MATCHING": "0"}
    old = {k: os.environ.get(k) for k in env}
    os.environ.update(env)
    try:
        lib = load()
        h = lib.complex_solver_hipmf_new()
        n = A.shape[0]
        rp, ci, zv = _zcsr(sp.tril(A).tocsr() if symmetric else A)
        assert lib.complex_solver_hipmf_initialize(h, ordering, 1, -1.0, -1, 0, int(symmetric), n, rp, ci, zv.ctypes.data) == 0
    finally:
        for k, val in old.items():
            os.environ.pop(k, None) if val is None else os.environ.__setitem__(k, val)
    npert, dre, dim, dex = C.c_int32(), C.c_double(), C.c_double(), C.c_double()
    assert lib.complex_solver_hipmf_factorize(h, None, None, C.byref(npert), None, C.byref(dre), C.byref(dim), C.byref(dex), 1, 0, zv) == 0
    assert npert.value == 0
    _check_complex_det(sp.csr_matrix(A).toarray(), complex(dre.value, dim.value), dex.value, tol)
    rng = np.random.default_rng(3)
    xs = rng.standard_normal(n) + 1j * rng.standard_normal(n)
    b = sp.csr_matrix(A) @ xs
    x = np.zeros(2 * n)
    assert lib.complex_solver_hipmf_solve(h, x, np.ascontiguousarray(np.stack([b.real, b.imag], axis=1).ravel()), 0) == 0
    assert np.max(np.abs(x[0::2] + 1j * x[1::2] - xs)) < max(1e-9, tol) * max(1.0, np.max(np.abs(xs)))
    lib.complex_solver_hipmf_drop(h)
