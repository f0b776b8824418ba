"""Round-2 device paths: symmetric L D L^T on the tiled fronts, the arena of working blocks, the stream SpMV, the BASELINE
configurations that had no -m gpu coverage (config 3 stand-ins at the published sizes, config 4 at >= 160^3 on one GPU)."""
import json
import os
import subprocess

import numpy as np
import pytest

import oracle_lib as O
from russell_amd import problems as P
from russell_amd.backend import Hipmf

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _solve(n, rp, ci, v, b, **kw):
    s = Hipmf()
    assert s.initialize(n, rp, ci, **kw) == 0
    code = s.factorize(v, compute_determinant=True)
    x = s.solve(b)
    st = s.stats()
    det = (s.det_coefficient, s.det_exponent)
    s.close()
    return code, x, st, det


def test_ldlt_vs_lu_vs_oracle_3d_22():
    # 3D 7-point Poisson 22^3 (tiled fronts up to ~500 rows): symmetric-lower input (L D L^T), the same matrix in general storage (LU),
    # and the CPU oracle with the same column order; no refinement, so the factors themselves are compared
    n, rp, ci, v = P.poisson3d(22)
    xs = P.manufactured_solution(n)
    b = P.csr_matvec(n, rp, ci, v, xs)
    lrp, lci, lv = P.lower_triangle(n, rp, ci, v)
    c1, x_sym, st1, det1 = _solve(n, lrp, lci, lv, b, general_symmetric=True, refinement_nstep=0)
    c2, x_spd, st2, det2 = _solve(n, lrp, lci, lv, b, positive_definite=True, refinement_nstep=0)
    c3, x_lu, st3, det3 = _solve(n, rp, ci, v, b, refinement_nstep=0)
    assert c1 == 0 and c2 == 0 and c3 == 0 and st1["max_front"] > 64
    assert np.array_equal(x_sym, x_spd)  # either flag selects the same path (interface_cudss.cu:324-333)
    s = Hipmf()
    assert s.initialize(n, rp, ci) == 0
    perm = s.permutation()
    s.close()
    rows = np.repeat(np.arange(n), np.diff(rp)).astype(np.int32)
    cp, ri, vx = O.coo_to_csc(n, n, rows, ci, v)
    xo = O.OracleLU(n, cp, ri, vx, q=perm).solve(b, nrefine=0)
    tol = 1e-10 * max(1.0, np.max(np.abs(xs)))
    assert np.max(np.abs(x_sym - xo)) < tol and np.max(np.abs(x_lu - xo)) < tol and np.max(np.abs(x_sym - xs)) < tol
    # same determinant (mantissa x 10^exponent) from D as from diag(U)
    assert det1[1] == det3[1] and abs(det1[0] - det3[0]) < 1e-8 * abs(det3[0])
    # the symmetric factor keeps E only: the persistent part of the pool shrinks
    assert st1["pool_bytes"] < 0.85 * st3["pool_bytes"]  # (0.80 with the supernode partition of late round 4, 0.78 before)


def test_symmetric_indefinite_lower_storage_still_meets_tolerance():
    # general_symmetric with an INDEFINITE matrix (shifted Laplacian): no interchanges on the tiled fronts, static pivoting + refinement
    n, rp, ci, v = P.poisson2d(90, 80)
    v = v.copy()
    rows = np.repeat(np.arange(n), np.diff(rp))
    v[rows == ci] -= 1.37  # eigenvalues of both signs
    xs = P.manufactured_solution(n)
    b = P.csr_matvec(n, rp, ci, v, xs)
    lrp, lci, lv = P.lower_triangle(n, rp, ci, v)
    code, x, st, det = _solve(n, lrp, lci, lv, b, general_symmetric=True)
    assert code == 0
    assert np.max(np.abs(x - xs)) < 1e-9 * np.max(np.abs(xs))
    code, x_lu, _, det_lu = _solve(n, rp, ci, v, b)
    assert det[1] == det_lu[1] and abs(det[0] - det_lu[0]) < 1e-7 * abs(det_lu[0])  # incl. the sign


def test_arena_reuse_gives_bit_identical_results(monkeypatch):
    # the working blocks of the tiled fronts share an arena (static lifetime plan); with re-use switched off every block has its own
    # storage: same arithmetic, same bits -- for LU and for L D L^T
    n, rp, ci, v = P.poisson3d(20)
    b = P.csr_matvec(n, rp, ci, v, P.manufactured_solution(n))
    lrp, lci, lv = P.lower_triangle(n, rp, ci, v)
    res = {}
    for reuse in ("1", "0"):
        monkeypatch.setenv("HIPMF_ARENA_REUSE", reuse)
        res[reuse] = (_solve(n, rp, ci, v, b, refinement_nstep=0), _solve(n, lrp, lci, lv, b, general_symmetric=True, refinement_nstep=0))
    for k in (0, 1):
        assert np.array_equal(res["1"][k][1], res["0"][k][1])
        assert res["1"][k][2]["pool_bytes"] < res["0"][k][2]["pool_bytes"]


def test_repeat_factorizations_reuse_the_arena_cleanly():
    # every factorisation zero-fills and re-fills the shared blocks: new values must not see leftovers of the previous ones
    n, rp, ci, v = P.poisson3d(18)
    xs = P.manufactured_solution(n)
    lrp, lci, lv = P.lower_triangle(n, rp, ci, v)
    s = Hipmf()
    assert s.initialize(n, lrp, lci, general_symmetric=True, refinement_nstep=0) == 0
    first = None
    for scale in (1.0, 3.0, 1.0):
        assert s.factorize(lv * scale) == 0
        x = s.solve(P.csr_matvec(n, rp, ci, v * scale, xs))
        assert np.max(np.abs(x - xs)) < 1e-11
        if scale == 1.0:
            first = x if first is None else first
            assert np.array_equal(x, first)
    s.close()


@pytest.mark.parametrize("case", ["general", "symmetric_lower", "fe_45_per_row", "long_row"])
def test_stream_spmv_matches_oracle(case):
    rng = np.random.default_rng(7)
    sym = False
    if case == "general":
        n, rp, ci, v = P.convection_diffusion2d(70, 60, scale_decades=1.0)
    elif case == "symmetric_lower":
        n, rp, ci, v = P.poisson3d(17, 15, 13)
        full = (n, rp, ci, v)
        rp, ci, v = P.lower_triangle(n, rp, ci, v)
        sym = True
    elif case == "fe_45_per_row":
        n, rp, ci, v = P.fe_block2d(24, 20, 5, symmetric=False, scale_decades=1.0)
        assert np.max(np.diff(rp)) == 45
    else:
        # one row with more than SPMV_CAP = 1024 stored entries (an arrow matrix), the rest short
        import scipy.sparse as sp
        n = 3000
        A = sp.diags([4.0], [0], shape=(n, n)).tolil()
        A[7, :] = rng.standard_normal(n)
        A[:, 7] = rng.standard_normal((n, 1))
        A[7, 7] = 50.0
        A = A.tocsr()
        A.sort_indices()
        rp, ci, v = A.indptr.astype(np.int32), A.indices.astype(np.int32), A.data.astype(np.float64)
    u = rng.standard_normal(n)
    s = Hipmf()
    assert s.initialize(n, rp, ci, general_symmetric=sym) == 0
    assert s.factorize(v) == 0
    y = s.mat_vec_mul(u, alpha=-2.5)
    if sym:
        yo = O.csr_matvec(full[0], full[1], full[2], full[3], u, alpha=-2.5)
    else:
        yo = O.csr_matvec(n, rp, ci, v, u, alpha=-2.5)
    assert np.max(np.abs(y - yo)) <= 1e-13 * np.max(np.abs(yo))
    # the residual of the refinement uses the same kernel: a solve must still converge to the refinement tolerance
    xs = P.manufactured_solution(n)
    b = O.csr_matvec(*full, xs) if sym else O.csr_matvec(n, rp, ci, v, xs)
    x = s.solve(b)
    assert np.max(np.abs(x - xs)) < 1e-8 * np.max(np.abs(xs))
    s.close()


def test_config3_bbmat_standin_through_the_matrix_market_harness(tmp_path):
    # BASELINE config 3, unsymmetric member: bbmat is not in the tree (no network); FE-like stand-in at its published size
    # (n = 38 720 ~ 38 744, 45 ~ 46 entries per row, dense 5 x 5 node blocks) with row scaling 10^U(-6, 6), written as MatrixMarket
    # and run through the reference's harness (bin/solve_matrix_market.rs) on the device, against SuperLU on the same file
    import scipy.io
    import scipy.sparse as sp
    import scipy.sparse.linalg as spla
    n, rp, ci, v = P.fe_block2d(88, 88, 5, symmetric=False, scale_decades=6.0)
    assert n == 38720 and 44.0 < rp[-1] / n < 46.0
    A = sp.csr_matrix((v, ci, rp), shape=(n, n))
    path = str(tmp_path / "bbmat_standin.mtx")
    scipy.io.mmwrite(path, A.tocoo(), precision=17)
    harness = os.path.join(ROOT, "russell_amd", "lib", "solve_matrix_market")
    env = {k: val for k, val in os.environ.items() if k != "RUSSELL_HIPMF_LIB"}
    p = subprocess.run([harness, "-g", "hipmf", path], env=env, capture_output=True, text=True, timeout=600)
    assert p.returncode == 0, p.stderr[-2000:]
    d = json.loads(p.stdout)
    assert d["matrix"]["nrow"] == n and d["matrix"]["nnz"] == int(rp[-1])
    assert d["verify"]["relative_error"] <= 1e-10
    # a system with a known solution through the C-ABI: forward error against SuperLU's on the same matrix (the matrix is
    # ill-conditioned by construction, so the yardstick is the CPU solver's own error, not an absolute number)
    xs = P.manufactured_solution(n)
    b = A @ xs
    s = Hipmf()
    assert s.initialize(n, rp, ci, values=v) == 0
    assert s.factorize(v) == 0
    x = s.solve(b)
    s.close()
    xo = spla.splu(A.tocsc(), permc_spec="COLAMD").solve(b)
    err, err_slu = np.max(np.abs(x - xs)), np.max(np.abs(xo - xs))
    assert err <= max(10.0 * err_slu, 1e-10 * np.max(np.abs(xs))), (err, err_slu)
    r = A @ x - b
    assert np.max(np.abs(r)) / (np.max(np.abs(v)) + 1.0) <= 1e-10


def test_config3_af_shell10_standin_symmetric_lower():
    # BASELINE config 3, symmetric member: shell-like stand-in at af_shell10's published size (n = 1 498 176 ~ 1 508 065,
    # 36 ~ 35 entries per row, dense 4 x 4 node blocks), lower triangle with positive_definite = 1 -> L D L^T.
    # (Written to MatrixMarket it would be a 1.5 GB text file: the C-ABI is called directly; the .mtx reader path is covered by
    # the bbmat stand-in above and by tests/test_reference_api_gpu.py.)
    n, rp, ci, v = P.fe_block2d(612, 612, 4, symmetric=True)
    assert n == 1498176 and 35.0 < rp[-1] / n < 36.0
    xs = P.manufactured_solution(n)
    import scipy.sparse as sp
    A = sp.csr_matrix((v, ci, rp), shape=(n, n))
    b = A @ xs
    lrp, lci, lv = P.lower_triangle(n, rp, ci, v)
    s = Hipmf()
    assert s.initialize(n, lrp, lci, positive_definite=True) == 0
    assert s.factorize(lv) == 0
    x = s.solve(b)
    st = s.stats()
    s.close()
    assert st["n_perturbed"] == 0
    r = A @ x - b
    assert np.max(np.abs(r)) / (np.max(np.abs(v)) + 1.0) <= 1e-10
    assert np.max(np.abs(x - xs)) <= 1e-8 * np.max(np.abs(xs))


def test_config4_matrix_160_cubed_on_one_gpu():
    # BASELINE config 4's matrix family (3D 7-point Poisson) at 160^3 = 4.1 M unknowns on ONE GPU, lower triangle -> L D L^T,
    # eight right-hand sides through the blocked solve; round 1 stopped at 144^3 (194 GB of fronts)
    N = 160
    n, rp, ci, v = P.poisson3d(N)
    xs = P.manufactured_solution(n)
    b = P.csr_matvec(n, rp, ci, v, xs)
    lrp, lci, lv = P.lower_triangle(n, rp, ci, v)
    s = Hipmf()
    assert s.initialize(n, lrp, lci, general_symmetric=True) == 0
    st = s.stats()
    assert st["pool_bytes"] < 120e9
    assert s.factorize(lv) == 0
    # independent random columns (a column-mixing bug that preserves direction would survive scalar multiples of one vector)
    XS = np.stack([np.random.default_rng([20260927, j]).standard_normal(n) for j in range(8)])
    B = np.stack([P.csr_matvec(n, rp, ci, v, XS[j]) for j in range(8)])
    X = s.solve_many(B)
    s.close()
    assert np.max(np.abs(X - XS)) < 1e-9 * np.max(np.abs(XS))


def test_config5_radau5_brusselator_reference_test_on_device():
    # the reference's own Radau5 test (russell_ode/tests/test_radau5_brusselator_pde.rs:31-44) with real + complex handles factorised
    # and solved on two threads (radau5.rs:270-296), repeat factorisations through the device-side value refresh
    from test_radau5_brusselator_cpu import check_reference_test, run
    d = run(None, "--npoint", "9", "--first-book", "--neg-exp-tol", "3", "--t1", "0.1")
    check_reference_test(d)
    assert d["concurrent"] is True
    d2 = run(None, "--npoint", "9", "--first-book", "--neg-exp-tol", "3", "--t1", "0.1", "--serial")
    assert d2["u_mid"] == d["u_mid"] and d2["v_mid"] == d["v_mid"] and d2["n_function"] == 24


def test_config5_radau5_brusselator_second_book_npoint_129():
    # the benchmark problem of bin/brusselator_pde.rs at npoint = 129 (ndim = 33 282), tolerance 1e-4, t1 = 1.5: a full run with
    # step rejections and Jacobian / factorisation re-use; conservation-free sanity: finite, bounded solution and consistent counters
    from test_radau5_brusselator_cpu import run
    d = run(None, "--npoint", "129")
    assert d["ndim"] == 2 * 129 * 129 and d["jac_nnz"] == 14 * 129 * 129
    assert d["n_accepted"] + d["n_rejected"] <= d["n_steps"] and d["n_factor"] <= d["n_steps"]
    assert d["n_lin_sol"] >= d["n_steps"] and 0.0 < d["u_mid"] < 10.0 and 0.0 < d["v_mid"] < 10.0


def test_config5_npoint_513_reproduces_the_reference_log():
    # data/logs/brus_pde_2nd_umfpack_24.txt (UMFPACK, npoint = 513, ndim = 526 338): the step / Newton / factorisation counters of the
    # whole integration and the last step size must come out the same with the HIPMF backend -- every linear solve of the run feeds
    # the step-size controller, so the counters only match when the solves are accurate to the controller's resolution
    from test_radau5_brusselator_cpu import run
    d = run(None, "--npoint", "513")
    assert d["ndim"] == 526338 and d["jac_nnz"] == 3684366
    assert (d["n_function"], d["n_jacobian"], d["n_factor"], d["n_lin_sol"]) == (266, 23, 44, 75)
    assert (d["n_steps"], d["n_accepted"], d["n_rejected"], d["n_iterations_max"]) == (44, 35, 9, 4)
    assert abs(d["h_accepted"] - 0.267457533813765) < 1e-9


def _complex_handle():
    import ctypes as C
    from russell_amd import _capi
    lib = _capi.load()
    h = lib.complex_solver_hipmf_new()
    assert h
    return lib, h


@pytest.mark.parametrize("symmetric", [False, True])
def test_complex_cabi_direct(symmetric):
    # complex_solver_hipmf_* (include/russell_hipmf.h; replaces interface_complex_umfpack.c:82-248): general and symmetric-lower complex
    # CSR, interleaved (re, im) values, against numpy's dense complex solve; value refresh through a triplet map with duplicates
    import ctypes as C
    import scipy.sparse as sp
    rng = np.random.default_rng(12)
    n = 300
    A = sp.random(n, n, density=0.03, random_state=3, format="csr") * (1.0 + 0.0j)
    A = A + 1j * sp.random(n, n, density=0.03, random_state=4, format="csr") + sp.diags((4.0 + 1.5j) * np.ones(n))
    if symmetric:
        A = (A + A.T) * 0.5
    A = sp.csr_matrix(A)
    A.sort_indices()
    S = sp.tril(A).tocsr() if symmetric else A
    S.sort_indices()
    rp, ci = S.indptr.astype(np.int32), S.indices.astype(np.int32)
    zv = np.ascontiguousarray(np.stack([S.data.real, S.data.imag], axis=1).ravel())
    xs = rng.standard_normal(n) + 1j * rng.standard_normal(n)
    b = A @ xs
    rhs = np.ascontiguousarray(np.stack([b.real, b.imag], axis=1).ravel())
    lib, h = _complex_handle()
    assert lib.complex_solver_hipmf_initialize(h, 0, 1, -1.0, -1, 0, int(symmetric), n, rp, ci, zv.ctypes.data) == 0
    assert lib.complex_solver_hipmf_initialize(h, 0, 1, -1.0, -1, 0, int(symmetric), n, rp, ci, None) == 700000
    eo, es, npert, rc = C.c_int32(), C.c_int32(), C.c_int32(), C.c_double()
    assert lib.complex_solver_hipmf_factorize(h, C.byref(eo), C.byref(es), C.byref(npert), C.byref(rc), None, None, None, 0, 0, zv) == 0
    x = np.zeros(2 * n)
    assert lib.complex_solver_hipmf_solve(h, x, rhs, 0) == 0
    xz = x[0::2] + 1j * x[1::2]
    xd = np.linalg.solve(A.toarray(), b)
    assert np.max(np.abs(xz - xd)) < 1e-11 and np.max(np.abs(xz - xs)) < 1e-11
    # every stored entry split into two triplets (a + ib = (0.25 a + i 2 b) + (0.75 a - i b)), shuffled
    nnz = ci.size
    trip_of = np.concatenate([np.arange(nnz), np.arange(nnz)])
    order = rng.permutation(2 * nnz)
    seg_idx = np.argsort(trip_of[order], kind="stable").astype(np.int32)
    seg_ptr = (2 * np.arange(nnz + 1)).astype(np.int32)
    scale = 1.7 - 0.4j
    tv = np.empty(2 * nnz, complex)
    first = np.zeros(2 * nnz, bool)
    first[seg_idx[0::2]] = True
    src = trip_of[order]
    vals = S.data * scale
    tv[first] = 0.25 * vals.real[src[first]] + 2.0j * vals.imag[src[first]]
    tv[~first] = 0.75 * vals.real[src[~first]] - 1.0j * vals.imag[src[~first]]
    tin = np.ascontiguousarray(np.stack([tv.real, tv.imag], axis=1).ravel())
    assert lib.complex_solver_hipmf_set_value_map(h, 2 * nnz, seg_ptr, seg_idx) == 0
    assert lib.complex_solver_hipmf_factorize_mapped(h, None, None, None, None, 0, tin) == 0
    assert lib.complex_solver_hipmf_solve(h, x, rhs, 0) == 0
    xz2 = x[0::2] + 1j * x[1::2]
    assert np.max(np.abs(xz2 - xd / scale)) < 1e-11
    # plain CSR values again: the identity map comes back
    assert lib.complex_solver_hipmf_factorize(h, None, None, None, None, None, None, None, 0, 0, zv) == 0
    assert lib.complex_solver_hipmf_solve(h, x, rhs, 0) == 0
    assert np.array_equal(x[0::2] + 1j * x[1::2], xz)
    lib.complex_solver_hipmf_drop(h)
    lib.complex_solver_hipmf_drop(None)
