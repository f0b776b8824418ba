"""Finite-difference Laplacian assembled on the device (hipmf_fdm_*) against the reference's own expected
matrices (tests/golden/fdm2d_reference_cases.json <- russell_pde/src/fdm_2d.rs tests) and against the CPU oracle.

CPU part: the oracle restatement is pinned on the golden matrices.  GPU part: the device triplets equal the oracle's bit for bit
(indices, order and values), and a Poisson problem is solved end to end with values that never leave the device.
"""
import json
import os

import numpy as np
import pytest

import oracle_lib as O

HERE = os.path.dirname(os.path.abspath(__file__))
CASES = json.load(open(os.path.join(HERE, "golden", "fdm2d_reference_cases.json")))["cases"]


def dense_from(n_rows, n_cols, trip, sym=0):
    i, j, v = trip
    a = np.zeros((n_rows, n_cols))
    np.add.at(a, (i, j), v)  # duplicates (mirrored ghost nodes) add up, as in CooMatrix::as_dense
    if sym in (1, 2):
        off = i != j
        np.add.at(a, (j[off], i[off]), v[off])
    return a


def mask_of(case):
    m = np.zeros(case["nx"] * case["ny"], np.uint8)
    m[case["prescribed"]] = 1
    return m if case["prescribed"] else None


@pytest.mark.parametrize("case", CASES, ids=[c["name"] for c in CASES])
@pytest.mark.parametrize("sym", [0, 1, 2])
def test_oracle_reproduces_the_reference_matrices(case, sym):
    r = O.fdm_sps(case["nx"], case["ny"], 1, (case["periodic_x"], case["periodic_y"], False), sym, mask_of(case),
                  (case["dx"], case["dy"], 1.0), (case["kx"], case["ky"], 0.0), case["alpha"])
    assert (r["nu"], r["np"]) == (case["nu"], case["np"])
    kb = dense_from(r["nu"], r["nu"], r["bar"], sym)
    assert np.array_equal(kb, np.array(case["kk_bar_dense"]))  # exact: small integers
    if case["kk_check_dense"] is not None:
        kc = dense_from(r["nu"], r["np"], r["check"])
        assert np.array_equal(kc, np.array(case["kk_check_dense"]))
    else:
        assert len(r["check"][0]) == 0
    # triplet budget of the reference: band * nu entries allocated (fdm_2d.rs:608-609)
    assert len(r["bar"][0]) <= (3 if sym else 5) * r["nu"]


def test_oracle_molecule_and_halving_follow_the_reference():
    # loop_over_molecule(0, ..) of get_matrices_work: 800, -100, -300 (fdm_2d.rs:1030-1038); K-bar halves boundary rows
    r = O.fdm_sps(4, 3, 1, (False, False, False), 0, None, (1.0, 1.0, 1.0), (100.0, 300.0, 0.0), 0.0)
    i, j, v = r["bar"]
    row5 = {int(c): float(x) for c, x in zip(j[i == 5], v[i == 5])}  # interior node 5: the bare molecule
    assert row5 == {5: 800.0, 4: -100.0, 6: -100.0, 1: -300.0, 9: -300.0}
    assert v[(i == 0) & (j == 0)].sum() == 200.0  # corner: halved twice
    # Helmholtz term lands on the diagonal only (fdm_2d.rs:622-624), before the halving
    r2 = O.fdm_sps(4, 3, 1, (False, False, False), 0, None, (1.0, 1.0, 1.0), (100.0, 300.0, 0.0), 8.0)
    assert r2["bar"][2][(i == 5) & (j == 5)].sum() == 808.0 and r2["bar"][2][(i == 0) & (j == 0)].sum() == 202.0


def _device_vs_oracle(lib_path, nx, ny, nz, periodic, sym, mask, d, k, alpha):
    from russell_amd.pde import FdmDevice
    want = O.fdm_sps(nx, ny, nz, periodic, sym, mask, d, k, alpha)
    f = FdmDevice(nx, ny, nz, periodic, sym, mask, lib_path=lib_path)
    assert (f.nu, f.np, f.nnz_bar, f.nnz_check) == (want["nu"], want["np"], len(want["bar"][0]), len(want["check"][0]))
    bi, bj, ci, cj = f.structure_device()
    bv, cv = f.values_device(d, k, alpha)
    assert np.array_equal(f.to_host(bi, f.nnz_bar, np.int32), want["bar"][0])
    assert np.array_equal(f.to_host(bj, f.nnz_bar, np.int32), want["bar"][1])
    assert np.array_equal(f.to_host(bv, f.nnz_bar, np.float64), want["bar"][2])  # bit-exact
    if f.nnz_check:
        assert np.array_equal(f.to_host(ci, f.nnz_check, np.int32), want["check"][0])
        assert np.array_equal(f.to_host(cj, f.nnz_check, np.int32), want["check"][1])
        assert np.array_equal(f.to_host(cv, f.nnz_check, np.float64), want["check"][2])
    f.close()


def test_device_kernels_match_oracle_in_the_emulator(emu_lib):
    # (development-only CPU emulation of the HIP kernels: the same source, small grids)
    rng = np.random.default_rng(5)
    for case in CASES:
        for sym in (0, 1, 2):
            _device_vs_oracle(emu_lib, case["nx"], case["ny"], 1, (case["periodic_x"], case["periodic_y"], False), sym, mask_of(case),
                              (case["dx"], case["dy"], 1.0), (case["kx"], case["ky"], 0.0), 0.0)
    mask = (rng.random(23 * 17) < 0.2).astype(np.uint8)
    _device_vs_oracle(emu_lib, 23, 17, 1, (False, True, False), 1, mask, (0.5, 0.25, 1.0), (3.0, 7.0, 0.0), 1.5)
    mask3 = (rng.random(9 * 7 * 5) < 0.15).astype(np.uint8)
    _device_vs_oracle(emu_lib, 9, 7, 5, (True, False, False), 0, mask3, (0.5, 0.25, 2.0), (3.0, 7.0, 11.0), 0.0)


@pytest.mark.gpu
@pytest.mark.parametrize("shape", [(4, 3, 1), (3, 4, 1), (257, 129, 1), (1000, 1000, 1), (41, 37, 29), (100, 100, 100)])
def test_device_assembly_equals_oracle_bitwise(shape):
    nx, ny, nz = shape
    rng = np.random.default_rng(nx * 7 + ny)
    ntot = nx * ny * nz
    for sym, periodic, dens in ((0, (False, False, False), 0.0), (1, (False, True, False), 0.1), (2, (True, False, nz > 1), 0.3)):
        mask = None if dens == 0.0 else (rng.random(ntot) < dens).astype(np.uint8)
        _device_vs_oracle(None, nx, ny, nz, periodic, sym, mask, (0.1, 0.3, 0.7), (2.0, 5.0, 11.0), 0.25)


@pytest.mark.gpu
def test_reference_matrices_on_the_device():
    from russell_amd.pde import FdmDevice
    for case in CASES:
        for sym in (0, 1, 2):
            f = FdmDevice(case["nx"], case["ny"], 1, (case["periodic_x"], case["periodic_y"], False), sym, mask_of(case))
            bi, bj, ci, cj = f.structure_device()
            bv, cv = f.values_device((case["dx"], case["dy"], 1.0), (case["kx"], case["ky"], 0.0), case["alpha"])
            trip = (f.to_host(bi, f.nnz_bar, np.int32), f.to_host(bj, f.nnz_bar, np.int32), f.to_host(bv, f.nnz_bar, np.float64))
            assert np.array_equal(dense_from(f.nu, f.nu, trip, sym), np.array(case["kk_bar_dense"]))
            if case["kk_check_dense"] is not None:
                tc = (f.to_host(ci, f.nnz_check, np.int32), f.to_host(cj, f.nnz_check, np.int32), f.to_host(cv, f.nnz_check, np.float64))
                assert np.array_equal(dense_from(f.nu, f.np, tc), np.array(case["kk_check_dense"]))
            f.close()


@pytest.mark.gpu
def test_poisson_solved_with_device_assembled_values():
    # Dirichlet problem on a 200 x 150 grid: phi = 0 prescribed on the whole boundary; K-bar assembled on the device as the lower
    # triangle (Sym::YesLower, the cuDSS / MUMPS contract), handed to the solver through the value map without leaving HBM;
    # two coefficient sets re-use the structure (repeat factorisation).  Checked against the oracle's LU on the same triplets.
    from russell_amd.backend import Hipmf
    from russell_amd.pde import FdmDevice, SYM_LOWER
    nx, ny = 200, 150
    mask = np.zeros((ny, nx), np.uint8)
    mask[0, :] = mask[-1, :] = mask[:, 0] = mask[:, -1] = 1
    f = FdmDevice(nx, ny, 1, (False, False, False), SYM_LOWER, mask.ravel())
    bi_d, bj_d, _, _ = f.structure_device()
    bi, bj = f.to_host(bi_d, f.nnz_bar, np.int32), f.to_host(bj_d, f.nnz_bar, np.int32)
    # structure -> CSR of the lower triangle + value map (CSR entry <- the triplets that fall on it), once
    order = np.lexsort((bj, bi))
    ri, rj = bi[order], bj[order]
    first = np.concatenate([[True], (ri[1:] != ri[:-1]) | (rj[1:] != rj[:-1])])
    ci = rj[first].astype(np.int32)
    rp = np.concatenate([[0], np.cumsum(np.bincount(ri[first], minlength=f.nu))]).astype(np.int32)
    seg_ptr = np.concatenate([np.flatnonzero(first), [len(order)]]).astype(np.int32)
    s = Hipmf()
    assert s.initialize(f.nu, rp, ci, general_symmetric=True) == 0
    assert s.set_value_map(seg_ptr, order.astype(np.int32)) == 0
    xs = 1.0 + np.sin(0.01 * np.arange(f.nu))
    vals = None
    for d, k, alpha in (((1.0 / nx, 1.0 / ny, 1.0), (1.0, 1.0, 0.0), 0.0), ((1.0 / nx, 1.0 / ny, 1.0), (2.5, 0.5, 0.0), 3.0)):
        vals = f.values_device(d, k, alpha, out=vals)
        assert s.factorize_mapped_device(vals[0]) == 0
        want = O.fdm_sps(nx, ny, 1, (False, False, False), 1, mask.ravel(), d, k, alpha)
        wi, wj, wv = want["bar"]
        b = O.coo_matvec(f.nu, wi, wj, wv, xs, sym_triangular=True)
        x = s.solve(b)
        assert np.max(np.abs(x - xs)) < 1e-9
    s.close()
    f.close()


# ---- Lagrange-multiplier form: M = [K C^T; C 0] of Fdm2d::get_matrices_lmm (fdm_2d.rs:672-748) --------------------------------------
@pytest.mark.parametrize("case", CASES, ids=[c["name"] for c in CASES])
@pytest.mark.parametrize("sym", [0, 1, 2])
def test_oracle_reproduces_the_reference_lmm_matrices(case, sym):
    # the reference asserts the same dense M for Sym::No, YesLower, YesUpper and YesFull (fdm_2d.rs:1094-1131) and C apart
    r = O.fdm_lmm(case["nx"], case["ny"], 1, (case["periodic_x"], case["periodic_y"], False), sym, mask_of(case),
                  (case["dx"], case["dy"], 1.0), (case["kx"], case["ky"], 0.0), case["alpha"])
    assert (r["neq"], r["nlag"], r["ndim"]) == (case["nx"] * case["ny"], case["np"], case["nx"] * case["ny"] + case["np"])
    mm = dense_from(r["ndim"], r["ndim"], r["mm"], sym)
    assert np.array_equal(mm, np.array(case["lmm_mm_dense"]))  # exact: small integers
    if case["lmm_cc_dense"] is not None:
        i, j, v = r["mm"]
        sel = (i >= r["neq"]) if sym != 2 else (j >= r["neq"])  # the C entries (the C^T ones for upper storage)
        ci, cj = (i[sel] - r["neq"], j[sel]) if sym != 2 else (j[sel] - r["neq"], i[sel])
        assert np.array_equal(dense_from(r["nlag"], r["neq"], (ci, cj, v[sel])), np.array(case["lmm_cc_dense"]))
    # triplet budget of the reference: band * neq + 2 nlag entries allocated (fdm_2d.rs:688-689)
    assert len(r["mm"][0]) <= (3 if sym else 5) * r["neq"] + 2 * r["nlag"]


def _device_lmm_vs_oracle(lib_path, nx, ny, nz, periodic, sym, mask, d, k, alpha):
    from russell_amd.pde import FdmDevice
    want = O.fdm_lmm(nx, ny, nz, periodic, sym, mask, d, k, alpha)
    f = FdmDevice(nx, ny, nz, periodic, sym, mask, lib_path=lib_path)
    neq, nlag, nnz = f.lmm_dims()
    assert (neq, nlag, nnz) == (want["neq"], want["nlag"], len(want["mm"][0]))
    di, dj = f.lmm_structure_device()
    dv = f.lmm_values_device(d, k, alpha)
    assert np.array_equal(f.to_host(di, nnz, np.int32), want["mm"][0])
    assert np.array_equal(f.to_host(dj, nnz, np.int32), want["mm"][1])
    assert np.array_equal(f.to_host(dv, nnz, np.float64), want["mm"][2])  # bit-exact
    f.close()


def test_device_lmm_kernels_match_oracle_in_the_emulator(emu_lib):
    rng = np.random.default_rng(6)
    for case in CASES:
        for sym in (0, 1, 2):
            _device_lmm_vs_oracle(emu_lib, case["nx"], case["ny"], 1, (case["periodic_x"], case["periodic_y"], False), sym, mask_of(case),
                                  (case["dx"], case["dy"], 1.0), (case["kx"], case["ky"], 0.0), 0.0)
    mask = (rng.random(23 * 17) < 0.2).astype(np.uint8)
    _device_lmm_vs_oracle(emu_lib, 23, 17, 1, (False, True, False), 1, mask, (0.5, 0.25, 1.0), (3.0, 7.0, 0.0), 1.5)
    _device_lmm_vs_oracle(emu_lib, 23, 17, 1, (True, False, False), 2, mask, (0.5, 0.25, 1.0), (3.0, 7.0, 0.0), 0.0)
    mask3 = (rng.random(9 * 7 * 5) < 0.15).astype(np.uint8)
    _device_lmm_vs_oracle(emu_lib, 9, 7, 5, (True, False, False), 0, mask3, (0.5, 0.25, 2.0), (3.0, 7.0, 11.0), 0.0)


def _solve_lmm(lib_path, nx, ny):
    """Dirichlet problem in its Lagrange-multiplier form, assembled on the device as the lower triangle and solved as the saddle-point
    system it is (weak diagonal in the multiplier rows: the handle mirrors it to general storage and matches); against the oracle's LU."""
    from russell_amd.backend import Hipmf
    from russell_amd.pde import FdmDevice, SYM_LOWER
    mask = np.zeros((ny, nx), np.uint8)
    mask[0, :] = mask[-1, :] = mask[:, 0] = mask[:, -1] = 1
    f = FdmDevice(nx, ny, 1, (False, False, False), SYM_LOWER, mask.ravel(), lib_path=lib_path)
    neq, nlag, nnz = f.lmm_dims()
    di, dj = f.lmm_structure_device()
    dv = f.lmm_values_device((1.0 / (nx - 1), 1.0 / (ny - 1), 1.0), (1.0, 1.0, 0.0), 0.0)
    ti, tj, tv = f.to_host(di, nnz, np.int32), f.to_host(dj, nnz, np.int32), f.to_host(dv, nnz, np.float64)
    f.close()
    import scipy.sparse as sp
    ndim = neq + nlag
    L = sp.coo_matrix((tv, (ti, tj)), shape=(ndim, ndim)).tocsr()  # duplicates (mirrored ghost nodes) summed, as COO -> CSR does
    L.sort_indices()
    assert (sp.triu(L, 1)).nnz == 0
    rng = np.random.default_rng(3)
    xs = rng.uniform(-1.0, 1.0, ndim)
    full = (L + sp.tril(L, -1).T).tocsr()
    b = full @ xs
    s = Hipmf(lib_path)
    assert s.initialize(ndim, L.indptr.astype(np.int32), L.indices.astype(np.int32), general_symmetric=True, values=L.data) == 0
    assert s.factorize(L.data) == 0
    x = s.solve(b)
    st = s.stats()
    s.close()
    assert st["n_perturbed"] == 0
    assert np.max(np.abs(x - xs)) <= 1e-9 * np.max(np.abs(xs))
    return ndim


def test_lagrange_form_is_solved_as_a_saddle_point_system_in_the_emulator(emu_lib):
    assert _solve_lmm(emu_lib, 14, 11) == 14 * 11 + 2 * (14 + 11) - 4


@pytest.mark.gpu
@pytest.mark.parametrize("shape", [(4, 3, 1), (257, 129, 1), (1000, 1000, 1), (41, 37, 29)])
def test_device_lmm_assembly_equals_oracle_bitwise(shape):
    nx, ny, nz = shape
    rng = np.random.default_rng(nx * 5 + ny)
    ntot = nx * ny * nz
    for sym, periodic, dens in ((0, (False, False, False), 0.05), (1, (False, True, False), 0.1), (2, (True, False, nz > 1), 0.3)):
        mask = (rng.random(ntot) < dens).astype(np.uint8)
        _device_lmm_vs_oracle(None, nx, ny, nz, periodic, sym, mask, (0.1, 0.3, 0.7), (2.0, 5.0, 11.0), 0.25)


@pytest.mark.gpu
def test_reference_lmm_matrices_on_the_device_and_a_saddle_point_solve():
    from russell_amd.pde import FdmDevice
    for case in CASES:
        for sym in (0, 1, 2):
            f = FdmDevice(case["nx"], case["ny"], 1, (case["periodic_x"], case["periodic_y"], False), sym, mask_of(case))
            neq, nlag, nnz = f.lmm_dims()
            di, dj = f.lmm_structure_device()
            dv = f.lmm_values_device((case["dx"], case["dy"], 1.0), (case["kx"], case["ky"], 0.0), case["alpha"])
            trip = (f.to_host(di, nnz, np.int32), f.to_host(dj, nnz, np.int32), f.to_host(dv, nnz, np.float64))
            assert np.array_equal(dense_from(neq + nlag, neq + nlag, trip, sym), np.array(case["lmm_mm_dense"]))
            f.close()
    _solve_lmm(None, 300, 200)
