"""Chained tiled steps (k_chain, kernels_factor_chain.hpp: all steps of a level in one launch, in-launch hand-offs) against one launch per
step, on the CPU emulator: the tile bodies are shared, so factors, pivots and determinants must agree bit for bit -- for LU (row
interchanges inside the tiles), for L D L^T, with the fine-grained waits, and whatever levels the knobs select.  The emulator runs the
workgroups of a launch one after the other: this checks the task lists (every piece present once, right front / step / tile), not the
hand-offs -- tests/test_round3_gpu.py repeats it on the device."""
import os

import numpy as np
import pytest

from russell_amd import problems as P
from russell_amd.backend import Hipmf


def _run(lib, n, rp, ci, v, env, **kw):
    old = {k: os.environ.get(k) for k in env}
    os.environ.update(env)
    try:
        s = Hipmf(lib)
        assert s.initialize(n, rp, ci, refinement_nstep=0, **kw) == 0
        assert s.factorize(v, compute_determinant=True) == 0
        x = s.solve(np.cos(np.arange(n)))
        out = (x, s.det_coefficient, s.det_exponent, s.permutation(), s.stats()["factor_launches"], s.counter("chain_fallbacks"))
        s.close()
        return out
    finally:
        for k, val in old.items():
            if val is None:
                os.environ.pop(k, None)
            else:
                os.environ[k] = val


def _cases():
    yield "poisson2d 52x48", P.poisson2d(52, 48), {}
    yield "convection-diffusion 46 (interchanges)", P.convection_diffusion2d(46, peclet=30.0, scale_decades=0.0), {}
    n, rp, ci, v = P.poisson3d(10)
    yield "poisson3d 10 lower (L D L^T)", (n,) + tuple(P.lower_triangle(n, rp, ci, v)), {"general_symmetric": True}


@pytest.mark.parametrize("case", list(_cases()), ids=lambda c: c[0])
def test_chained_steps_give_the_same_factor_bit_for_bit(emu_lib, case):
    _, (n, rp, ci, v), kw = case
    ref = _run(emu_lib, n, rp, ci, v, {"HIPMF_FACTOR_CHAIN": "0"}, **kw)
    variants = [{"HIPMF_FACTOR_CHAIN": "1"},  # its default selection: levels with at most 8 steps
                {"HIPMF_FACTOR_CHAIN": "1", "HIPMF_CHAIN_MAX_STEPS": "1000", "HIPMF_CHAIN_FINE": "1"}]  # every level, fine-grained waits
    for env in variants:
        got = _run(emu_lib, n, rp, ci, v, env, **kw)
        assert np.array_equal(ref[0], got[0]), env
        assert ref[1:3] == got[1:3], env
        assert np.array_equal(ref[3], got[3])
        assert got[5] == 0
    assert _run(emu_lib, n, rp, ci, v, variants[1], **kw)[4] <= ref[4]  # (fewer launches as soon as a level has more than one step)


@pytest.mark.parametrize("case", list(_cases()), ids=lambda c: c[0])
def test_small_update_tiles_give_the_same_factor_bit_for_bit(emu_lib, case):
    # k_update32: 32 x 32 trailing-update tiles, one wavefront per tile, on the levels of mid-size fronts (default for LU up to 256
    # rows; HIPMF_UPD32_MAXF).  A wavefront does the same 2 x 2 MFMA tiles in the same k order as a quarter of a 64 x 64 tile.
    _, (n, rp, ci, v), kw = case
    ref = _run(emu_lib, n, rp, ci, v, {"HIPMF_UPD32_MAXF": "0"}, **kw)
    for mf in ("100000",):
        got = _run(emu_lib, n, rp, ci, v, {"HIPMF_UPD32_MAXF": mf}, **kw)
        assert np.array_equal(ref[0], got[0]), mf
        assert ref[1:3] == got[1:3], mf
        assert np.array_equal(ref[3], got[3])


def _split_cases():  # separators of more than 64 vertices: a group of panels with a follower
    yield "poisson2d 66x68", P.poisson2d(66, 68), {}
    yield "convection-diffusion 66 (interchanges)", P.convection_diffusion2d(66, peclet=30.0, scale_decades=0.0), {}


@pytest.mark.parametrize("case", list(_split_cases()), ids=lambda c: c[0])
def test_split_full_updates_give_the_same_factor_bit_for_bit(emu_lib, case):
    # HIPMF_UPD_SPLIT: the last update of a group of panels goes out as two launches -- first block column / row (what the next group's
    # panels touch) and the other tiles (on a side stream on the device).  Same tiles, same arithmetic: the same bits.  64 x 64 tiles
    # only (HIPMF_UPD32_MAXF=0), every full step with a follower (threshold 1), no one-workgroup fronts so that the tiled path has work.
    _, (n, rp, ci, v), kw = case
    # (HIPMF_RELAX / _BIG: the amalgamation of rounds 1 - 4, which gives these small problems fronts with enough full steps)
    base = {"HIPMF_UPD32_MAXF": "0", "HIPMF_MID_LU": "0", "HIPMF_MID_FRONT": "0", "HIPMF_RELAX": "4,16,48,0.8,0.1,0.05", "HIPMF_RELAX_BIG": "0"}
    ref = _run(emu_lib, n, rp, ci, v, dict(base, HIPMF_UPD_SPLIT="0"), **kw)
    got = _run(emu_lib, n, rp, ci, v, dict(base, HIPMF_UPD_SPLIT="1"), **kw)
    assert np.array_equal(ref[0], got[0])
    assert ref[1:3] == got[1:3]
    assert np.array_equal(ref[3], got[3])
    assert got[4] > ref[4]  # (some step was split: one launch more for each)
