"""kernels_factor_front.hpp: ONE workgroup carries a front with f > 64 through its whole partial factorisation in one launch.  Two
kernels: k_front_lu (the default, at most 32 pivots: the pivot block is inverted in the registers of one wavefront, the rest are
products with the inverse) and k_front (opt-in, HIPMF_MID_MMAX: up to 64 pivots, Gauss-Jordan over the pivot rows on eight
wavefronts).  Both leave another (E, E') pair than the tiled launches (E_top = inv(F11), E'_left = I) and the same pivots.  On the CPU
emulator: against the tiled launches (HIPMF_MID_FRONT=0), against SuperLU, determinants included; fronts with very few pivots (the
lanes beyond the pivot block must not store: the bug the device found in round 4), row interchanges inside the pivot block, and the
many-right-hand-side solves through the dense-top flag.  tests/test_round4_gpu.py repeats it on the device."""
import os

import numpy as np
import pytest
import scipy.sparse as sp
import scipy.sparse.linalg as spla

from russell_amd import problems as P
from russell_amd.backend import Hipmf


def _run(lib, n, rp, ci, v, env, nrhs=1, **kw):
    old = {k: os.environ.get(k) for k in env}
    os.environ.update(env)
    try:
        s = Hipmf(lib)
        assert s.initialize(n, rp, ci, refinement_nstep=0, **kw) == 0
        assert s.factorize(v, compute_determinant=True) == 0
        rng = np.random.default_rng(5)
        B = rng.standard_normal((nrhs, n))
        X = s.solve_many(B) if nrhs > 1 else s.solve(B[0])[None, :]
        out = (X, B, s.det_coefficient, s.det_exponent, s.counter("mid_fronts"), s.num_perturbed)
        s.close()
        return out
    finally:
        for k, val in old.items():
            if val is None:
                os.environ.pop(k, None)
            else:
                os.environ[k] = val


def _two_leaves_and_a_root(p, m, seed, weak=False):
    """Natural order: two leaf supernodes of p columns each, both coupled densely to the same m later rows / columns (their fronts have p
    pivots and m off-diagonal rows: k_front's fronts when p + m > 64), then the m x m remainder (root)."""
    rng = np.random.default_rng(seed)
    n = 2 * p + m
    A = np.zeros((n, n))
    for g in range(2):
        a0 = g * p
        A[a0:a0 + p, a0:a0 + p] = rng.standard_normal((p, p)) * 0.3
        A[a0:a0 + p, 2 * p:] = rng.standard_normal((p, m)) * 0.3
        A[2 * p:, a0:a0 + p] = rng.standard_normal((m, p)) * 0.3
        for i in range(p):
            A[a0 + i, a0 + i] += 0.05 if weak else 3.0  # (weak: the pivot search has to interchange rows inside the pivot block)
    A[2 * p:, 2 * p:] = np.diag(4.0 + rng.random(m)) + np.diag(rng.standard_normal(m - 1) * 0.2, 1) + np.diag(rng.standard_normal(m - 1) * 0.2, -1)
    M = sp.csr_matrix(A)
    M.sort_indices()
    return n, M.indptr.astype(np.int32), M.indices.astype(np.int32), M.data.astype(np.float64), M


LU32 = {"HIPMF_MID_FRONT": "1"}                                                   # the default: k_front_lu where p <= 32
GJ8 = {"HIPMF_MID_FRONT": "1", "HIPMF_MID_LU": "0", "HIPMF_MID_MMAX": "80"}      # k_front (eight wavefronts) for everything it takes
HAND = [(2, 70, False, LU32), (5, 75, False, LU32), (6, 62, True, LU32), (20, 60, True, LU32), (32, 120, True, LU32), (2, 70, False, GJ8), (6, 62, True, GJ8),
        (20, 60, True, GJ8), (33, 80, False, GJ8), (64, 40, True, GJ8)]


@pytest.mark.parametrize("p,m,weak,env", HAND, ids=lambda v: ("gj8" if v is GJ8 else "lu32") if isinstance(v, dict) else str(v))
def test_one_workgroup_fronts_by_hand(emu_lib, p, m, weak, env):
    n, rp, ci, v, M = _two_leaves_and_a_root(p, m, seed=100 * p + m, weak=weak)
    kw = {"ordering": 2}  # HIPMF_ORDERING_NONE: the supernodes are the ones built above
    got = _run(emu_lib, n, rp, ci, v, env, **kw)
    ref = _run(emu_lib, n, rp, ci, v, {"HIPMF_MID_FRONT": "0"}, **kw)
    assert got[4] >= 1 and ref[4] == 0  # the leaves take the new kernel (the relaxed amalgamation may merge a leaf of a few columns into the root)
    xo = spla.splu(M.tocsc()).solve(got[1][0])
    scale = np.max(np.abs(xo))
    assert np.max(np.abs(got[0][0] - xo)) <= 1e-11 * max(scale, 1.0), (p, m)
    assert np.max(np.abs(ref[0][0] - xo)) <= 1e-11 * max(scale, 1.0), (p, m)
    # the same pivots: determinant mantissa / exponent of the LU and of the Gauss-Jordan elimination agree
    assert got[3] == ref[3] and abs(got[2] - ref[2]) <= 1e-10 * abs(ref[2])
    sign, logdet = np.linalg.slogdet(M.toarray())
    assert np.sign(got[2]) == sign and abs(np.log10(abs(got[2])) + got[3] - logdet / np.log(10.0)) < 1e-9


@pytest.mark.parametrize("env", [LU32, GJ8], ids=["lu32", "gj8"])
@pytest.mark.parametrize("case", ["poisson2d 60x52", "convection-diffusion 50", "fe blocks 12x12x3"])
def test_tree_with_one_workgroup_fronts_against_the_tiled_launches(emu_lib, case, env):
    if case.startswith("poisson"):
        n, rp, ci, v = P.poisson2d(60, 52)
    elif case.startswith("convection"):
        n, rp, ci, v = P.convection_diffusion2d(50, peclet=30.0, scale_decades=2.0)
    else:
        n, rp, ci, v = P.fe_block2d(12, 12, 3, symmetric=False, scale_decades=1.0)
    got = _run(emu_lib, n, rp, ci, v, env, nrhs=9)   # (9 columns: the blocked forward slabs read the dense E_top too)
    ref = _run(emu_lib, n, rp, ci, v, {"HIPMF_MID_FRONT": "0"}, nrhs=9)
    assert got[4] > 0 and ref[4] == 0
    A = sp.csr_matrix((v, ci, rp), shape=(n, n)).tocsc()
    lu = spla.splu(A)
    for j in range(9):
        xo = lu.solve(got[1][j])
        tol = 1e-10 * max(1.0, np.max(np.abs(xo)))
        assert np.max(np.abs(got[0][j] - xo)) <= tol and np.max(np.abs(ref[0][j] - xo)) <= tol
    assert got[3] == ref[3] and abs(got[2] - ref[2]) <= 1e-9 * abs(ref[2])
    assert got[5] == ref[5] == 0
