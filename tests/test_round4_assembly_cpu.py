"""Round 4, second half, on the CPU emulator: the extend-add with first touch (k_extend_add_lds, kernels_extend_add.hpp: the task builds its
tile of the parent in LDS and writes it once; the task that holds a tiled front's first diagonal tile also factorises it; LU and L D L^T
fronts) against the
zero-fill + scatter + read-modify-write launches it replaces -- the order of the additions is the same, so factors, pivots and
determinants agree bit for bit; and the one-launch tiled step with the inverse of the diagonal tile (kernels_factor_binv.hpp, an
opt-in) against the two-launch step: another elimination order, so solutions and determinants agree to rounding."""
import numpy as np
import pytest
import scipy.sparse as sp

from russell_amd import problems as P
from test_mid_fronts_cpu import _run


def _cases():
    yield "poisson2d 52x48", P.poisson2d(52, 48), {}
    yield "convection-diffusion 66 (interchanges)", P.convection_diffusion2d(66, peclet=30.0, scale_decades=0.0), {}
    n, rp, ci, v = P.poisson3d(8)
    yield "poisson3d 8, lower triangle given, factorised as LU (mirrored entries)", (n,) + tuple(P.lower_triangle(n, rp, ci, v)), {}


@pytest.mark.parametrize("case", list(_cases()), ids=lambda c: c[0])
def test_extend_add_with_first_touch_gives_the_same_factor_bit_for_bit(emu_lib, case):
    _, (n, rp, ci, v), kw = case
    ref = _run(emu_lib, n, rp, ci, v, {"HIPMF_EA_LDS": "0"}, nrhs=3, **kw)
    for env in ({"HIPMF_EA_LDS": "1", "HIPMF_EA_LU": "0"},  # tiles only
                {"HIPMF_EA_LDS": "1", "HIPMF_EA_LU": "1"}):  # + the first diagonal tiles (the default)
        got = _run(emu_lib, n, rp, ci, v, env, nrhs=3, **kw)
        assert np.array_equal(ref[0], got[0]), env
        assert ref[2:4] == got[2:4] and ref[5] == got[5], env


def test_first_touch_extend_add_of_symmetric_fronts_gives_the_same_factor_bit_for_bit(emu_lib):
    # L D L^T fronts: only entries on or below the parent's diagonal are added, tiles strictly above it have no task (k_extend_add_lds<true>)
    n, rp, ci, v = P.poisson3d(8)
    low = (n,) + tuple(P.lower_triangle(n, rp, ci, v))
    a = _run(emu_lib, *low, {"HIPMF_EA_LDS": "0"}, general_symmetric=True)
    for env in ({"HIPMF_EA_LDS": "1", "HIPMF_EA_LU": "0"}, {"HIPMF_EA_LDS": "1", "HIPMF_EA_LU": "1"}):
        b = _run(emu_lib, *low, env, general_symmetric=True)
        assert np.array_equal(a[0], b[0]) and a[2:4] == b[2:4], env
    n2, rp2, ci2, v2 = P.poisson2d(52, 48)  # (a front with more than one tile row: tiles above the diagonal are skipped)
    low2 = (n2,) + tuple(P.lower_triangle(n2, rp2, ci2, v2))
    a2 = _run(emu_lib, *low2, {"HIPMF_EA_LDS": "0"}, general_symmetric=True)
    b2 = _run(emu_lib, *low2, {"HIPMF_EA_LDS": "1"}, general_symmetric=True)
    assert np.array_equal(a2[0], b2[0]) and a2[2:4] == b2[2:4]
    xo = np.linalg.solve(sp.csr_matrix((v, ci, rp), shape=(n, n)).toarray(), a[1][0])
    assert np.max(np.abs(a[0][0] - xo)) <= 1e-11 * np.max(np.abs(xo))


@pytest.mark.parametrize("case", [c for c in _cases() if "lower" not in c[0]], ids=lambda c: c[0])
def test_one_launch_steps_with_the_inverse_of_the_diagonal_tile(emu_lib, case):
    _, (n, rp, ci, v), kw = case
    A = sp.csr_matrix((v, ci, rp), shape=(n, n))
    for extra in ({"HIPMF_MID_LU": "0", "HIPMF_MID_FRONT": "0", "HIPMF_UPD32_MAXF": "0"},):  # (every big front tiled, 64 x 64 tiles)
        ref = _run(emu_lib, n, rp, ci, v, dict(extra, HIPMF_BLOCK_INV="0"), nrhs=2, **kw)
        got = _run(emu_lib, n, rp, ci, v, dict(extra, HIPMF_BLOCK_INV="1"), nrhs=2, **kw)
        for j in range(2):
            r = A @ got[0][j] - got[1][j]
            assert np.max(np.abs(r)) <= 1e-10 * (np.max(np.abs(v)) * np.max(np.abs(got[0][j])) + np.max(np.abs(got[1][j])))
            assert np.max(np.abs(got[0][j] - ref[0][j])) <= 1e-8 * np.max(np.abs(ref[0][j]))
        assert got[3] == ref[3] and abs(got[2] - ref[2]) <= 1e-9 * abs(ref[2])  # the same pivots: the same determinant
        assert got[5] == ref[5]
