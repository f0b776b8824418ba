"""Dependency-driven triangular solves (kernels_solve_fused.hpp, one launch per direction) against the
level-set launches (kernels_solve.hpp): same factor, same right-hand side.  With identical slab shapes
(HIPMF_SOLVE_SLAB64=1) the two paths add the same numbers in the same order, so a stale or torn hand-off
between workgroups shows up as a bit difference; the test repeats the solve to catch intermittent ones."""
import os

import numpy as np
import pytest

from russell_amd import problems as P
from russell_amd.backend import Hipmf

pytestmark = pytest.mark.gpu


def _solve(n, rp, ci, v, b, fused, slab64, reps=1, nrefine=0):
    old = {k: os.environ.get(k) for k in ("HIPMF_FUSED_SOLVE", "HIPMF_SOLVE_SLAB64")}
    os.environ["HIPMF_FUSED_SOLVE"] = "1" if fused else "0"
    os.environ["HIPMF_SOLVE_SLAB64"] = "1" if slab64 else "0"
    try:
        s = Hipmf()
        assert s.initialize(n, rp, ci, refinement_nstep=nrefine) == 0
        assert s.factorize(v) == 0
        outs = []
        for _ in range(reps):
            outs.append(s.solve(b))
        st = s.stats()
        s.close()
        return outs, st
    finally:
        for k, val in old.items():
            if val is None:
                os.environ.pop(k, None)
            else:
                os.environ[k] = val


@pytest.mark.parametrize("grid", [37, 300, 1000])
def test_fused_equals_level_set_bitwise(grid):
    n, rp, ci, v = P.poisson2d(grid)
    xs = P.manufactured_solution(n)
    b = P.csr_matvec(n, rp, ci, v, xs)
    (ref,), st_l = _solve(n, rp, ci, v, b, fused=False, slab64=True)
    outs, st_f = _solve(n, rp, ci, v, b, fused=True, slab64=True, reps=8)
    assert st_f["solve_launches"] <= 6 < st_l["solve_launches"] or grid < 64
    for x in outs:
        assert np.array_equal(ref, x)
    assert np.max(np.abs(ref - xs)) < 1e-9


def test_fused_default_slabs_unsymmetric():
    # convection-diffusion (unsymmetric values, row interchanges inside the pivot blocks), default slab shapes
    n, rp, ci, v = P.convection_diffusion2d(160, peclet=30.0, scale_decades=0.0)
    xs = P.manufactured_solution(n)
    b = P.csr_matvec(n, rp, ci, v, xs)
    (ref,), _ = _solve(n, rp, ci, v, b, fused=False, slab64=False, nrefine=2)
    outs, _ = _solve(n, rp, ci, v, b, fused=True, slab64=False, reps=4, nrefine=2)
    for x in outs:
        assert np.array_equal(outs[0], x)  # run-to-run reproducible
        assert np.max(np.abs(x - xs)) / np.max(np.abs(xs)) < 1e-10
    assert np.max(np.abs(outs[0] - ref)) / np.max(np.abs(xs)) < 1e-12


def test_fused_poisson3d():
    n, rp, ci, v = P.poisson3d(22)
    xs = P.manufactured_solution(n)
    b = P.csr_matvec(n, rp, ci, v, xs)
    (ref,), _ = _solve(n, rp, ci, v, b, fused=False, slab64=True)
    outs, _ = _solve(n, rp, ci, v, b, fused=True, slab64=True, reps=4)
    for x in outs:
        assert np.array_equal(ref, x)


def test_poisson3d_fronts_beyond_the_lds_staging_limit():
    # 88^3: the root front has > 7 936 rows, more than the level-set solve kernels can stage in LDS; the
    # dependency-driven solves sweep such fronts in chunks (needs ~26 GB of HBM)
    n, rp, ci, v = P.poisson3d(88)
    xs = P.manufactured_solution(n)
    b = P.csr_matvec(n, rp, ci, v, xs)
    s = Hipmf()
    assert s.initialize(n, rp, ci) == 0
    assert s.stats()["max_front"] > 7936
    assert s.factorize(v) == 0
    x = s.solve(b)
    assert np.max(np.abs(x - xs)) / np.max(np.abs(xs)) < 1e-11
    s.close()


@pytest.mark.parametrize("grid,nrhs", [(60, 6), (300, 9), (200, 20)])
def test_many_rhs_blocks_agree_with_single_solves(grid, nrhs):
    # solve_many sends blocks of SF_KMAX (8) right-hand sides through the dependency-driven kernels together (the factor is read
    # once per block).  The small fronts use the single-column arithmetic per column; the slabs of the big fronts run on MFMA tiles
    # (E tile x 8 vectors), i.e. with another summation order: blocked and single solves agree to rounding, not bit for bit.
    n, rp, ci, v = P.poisson2d(grid)
    rng = np.random.default_rng(grid)
    XS = rng.standard_normal((nrhs, n))
    B = np.array([P.csr_matvec(n, rp, ci, v, XS[j]) for j in range(nrhs)])
    s = Hipmf()
    assert s.initialize(n, rp, ci) == 0
    assert s.factorize(v) == 0
    X = s.solve_many(B)
    for j in range(nrhs):
        xj = s.solve(B[j])
        assert np.max(np.abs(X[j] - xj)) <= 1e-12 * np.max(np.abs(xj))
    assert np.max(np.abs(X - XS)) / np.max(np.abs(XS)) < 1e-10
    s.close()


@pytest.mark.gpu
def test_assemble_once_tasks_give_the_same_bits(monkeypatch):
    # forward pass of the largest fronts: the front's vector (right-hand side + children's updates) is assembled once by tasks of
    # their own instead of by every slab (numeric.cpp, kind-1 SfTask).  Forced onto every tiled front of a small problem: the
    # single-column results are bit-identical (same sums in the same order), the blocked ones agree to rounding.
    n, rp, ci, v = P.poisson2d(150, 140)
    rng = np.random.default_rng(11)
    v = v * (1.0 + 0.3 * rng.uniform(-1, 1, v.size))
    xs = P.manufactured_solution(n)
    b = P.csr_matvec(n, rp, ci, v, xs)
    B = np.array([b * (1.0 + 0.1 * j) for j in range(11)])
    got = []
    for env in ({"HIPMF_SF_ASM_FRONT": "0", "HIPMF_SF_BIG_ROWS": "0"}, {"HIPMF_SF_ASM_FRONT": "65", "HIPMF_SF_BIG_FRONT": "65"},
                {"HIPMF_SF_ASM_FRONT": "65", "HIPMF_SF_BIG_ROWS": "0"}):
        for k, val in env.items():
            monkeypatch.setenv(k, val)
        s = Hipmf()
        assert s.initialize(n, rp, ci) == 0
        assert s.factorize(v) == 0
        got.append((s.solve(b), s.solve_many(B)))
        s.close()
        for k in env:
            monkeypatch.delenv(k)
    assert np.max(np.abs(got[0][0] - xs)) < 1e-10
    # same slab shapes (third configuration against the first): the same sums in the same order
    assert np.array_equal(got[2][0], got[0][0])
    # with the 64-row slabs of the largest fronts the dot products are grouped differently: equal to rounding
    assert np.max(np.abs(got[1][0] - got[0][0])) <= 1e-10 * np.max(np.abs(got[0][0]))
    for _, X in got[1:]:
        assert np.max(np.abs(X - got[0][1])) <= 1e-10 * np.max(np.abs(got[0][1]))
