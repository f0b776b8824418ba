"""Ordering::Amd (enums.rs:71-155): the backend's own approximate minimum degree -- emulated kernels and host mirror, no device."""
import ctypes as C
import hashlib

import numpy as np
import pytest

from russell_amd import problems as P
from russell_amd import sparse as RS
from russell_amd.backend import Hipmf

ORDERING_DEFAULT, ORDERING_AMD = 0, 3  # include/russell_hipmf.h


@pytest.fixture()
def host_on_emu(emu_lib):
    lib = RS._L()
    lib.rh_set_hipmf_library.argtypes = [C.c_char_p]
    lib.rh_set_hipmf_library(emu_lib.encode())
    yield lib
    lib.rh_set_hipmf_library(b"")


def _random_pattern(n, seed, hubs=0):
    """Unsymmetric pattern without small separators, diagonally dominant values; a few hub rows when asked for."""
    rng = np.random.default_rng(seed)
    A = np.zeros((n, n))
    for i in range(n):
        for j in rng.integers(0, n, size=rng.integers(1, 5)):
            A[i, j] = rng.uniform(-1.0, 1.0)
    for h in range(hubs):
        A[h, rng.integers(0, n, size=n // 2)] = rng.uniform(-1.0, 1.0)
    A[np.arange(n), np.arange(n)] = np.sum(np.abs(A), axis=1) + 1.0
    rp = np.zeros(n + 1, dtype=np.int32)
    ci, v = [], []
    for i in range(n):
        nzj = np.nonzero(A[i])[0]
        ci.extend(nzj.tolist()), v.extend(A[i, nzj].tolist())
        rp[i + 1] = len(ci)
    return A, rp, np.array(ci, dtype=np.int32), np.array(v)


@pytest.mark.parametrize("case", ["poisson2d 31x29", "poisson3d 9", "random 300", "random 260 with hubs"])
def test_amd_ordering_factorises_and_solves(emu_lib, case):
    if case.startswith("poisson2d"):
        n, rp, ci, v = P.poisson2d(31, 29)
    elif case.startswith("poisson3d"):
        n, rp, ci, v = P.poisson3d(9)
    else:
        n = int(case.split()[1])
        _, rp, ci, v = _random_pattern(n, 42, hubs=2 if "hubs" in case else 0)
    xs = P.manufactured_solution(n)
    b = P.csr_matvec(n, rp, ci, v, xs)
    s = Hipmf(emu_lib)
    assert s.initialize(n, rp, ci, ordering=ORDERING_AMD) == 0
    p = np.ascontiguousarray(s.permutation(), dtype=np.int64)
    assert np.array_equal(np.sort(p), np.arange(n))  # a permutation
    assert s.factorize(v) == 0
    x = s.solve(b)
    s.close()
    assert np.max(np.abs(x - xs)) <= 1e-10 * np.max(np.abs(xs))


def test_amd_beats_the_dissection_on_a_pattern_without_separators_and_is_close_on_a_grid(emu_lib):
    # a random pattern has no small separators: level-structure dissection cuts it badly, minimum degree does not care
    _, rp, ci, _ = _random_pattern(900, 7)
    fill = {}
    for name, o in (("nd", ORDERING_DEFAULT), ("amd", ORDERING_AMD)):
        s = Hipmf(emu_lib)
        assert s.initialize(900, rp, ci, ordering=o) == 0
        fill[name] = s.stats()["nnz_l"]
        s.close()
    assert fill["amd"] < 0.8 * fill["nd"], fill
    # on a grid the dissection is the better ordering, minimum degree stays within a small factor
    n, rp, ci, _ = P.poisson2d(60, 60)
    for name, o in (("nd", ORDERING_DEFAULT), ("amd", ORDERING_AMD)):
        s = Hipmf(emu_lib)
        assert s.initialize(n, rp, ci, ordering=o) == 0
        fill[name] = s.stats()["nnz_l"]
        s.close()
    assert fill["amd"] < 1.5 * fill["nd"], fill
    # known size of the fill of a minimum-degree ordering on the 60 x 60 five-point grid: about 8 n log2 n / 3.1 ... bounded loosely
    assert fill["amd"] < 25 * n


def test_amd_permutation_is_reproducible(emu_lib):
    # no threads, no pointer hashing: two handles give the same permutation, and it is pinned like the dissection's
    n, rp, ci, _ = P.poisson2d(48, 40)
    got = []
    for _ in range(2):
        s = Hipmf(emu_lib)
        assert s.initialize(n, rp, ci, ordering=ORDERING_AMD) == 0
        got.append(hashlib.sha256(np.ascontiguousarray(s.permutation(), dtype=np.int32).tobytes()).hexdigest()[:16])
        s.close()
    assert got[0] == got[1]
    assert got[0] == GOLDEN_AMD_48x40, got[0]


GOLDEN_AMD_48x40 = "321b30c5963366f7"


def test_host_mirror_maps_the_minimum_degree_family_and_reports_what_ran(host_on_emu):
    n, rp, ci, v = P.poisson2d(12, 10)
    rows = np.repeat(np.arange(n), np.diff(rp))
    xs = P.manufactured_solution(n)
    b = P.csr_matvec(n, rp, ci, v, xs)
    want = {RS.Ordering.Amd: "Amd", RS.Ordering.Amf: "Amd", RS.Ordering.Qamd: "Amd", RS.Ordering.Metis: "Nd", RS.Ordering.Auto: "Nd",
            RS.Ordering.Colamd: "Nd", RS.Ordering.No: "No"}
    for o, name in want.items():
        coo = RS.CooMatrix(n, n, len(v))
        coo.put_many(rows.astype(np.int32), ci.astype(np.int32), v.astype(np.float64))
        par = RS.LinSolParams()
        par.ordering = o
        solver = RS.LinSolver(RS.Genie.Hipmf)
        solver.actual.factorize(coo, par)
        x = solver.actual.solve(b)
        assert np.max(np.abs(x - xs)) < 1e-11
        assert solver.actual.stats()["output"]["effective_ordering"] == name, (o, solver.actual.stats()["output"])


def test_ordering_best_keeps_the_sparser_of_the_two(emu_lib, host_on_emu):
    # Ordering::Best (UMFPACK_ORDERING_BEST: try several orderings, keep the sparsest): dissection on the grid, minimum degree on the
    # pattern without separators; the effective ordering says which
    ORDERING_BEST = 4
    n, rp, ci, v = P.poisson2d(60, 60)
    _, rrp, rci, rv = _random_pattern(900, 7)
    for (nn, p, c, vals, want) in ((n, rp, ci, v, ORDERING_DEFAULT), (900, rrp, rci, rv, ORDERING_AMD)):
        fill = {}
        for o in (ORDERING_DEFAULT, ORDERING_AMD, ORDERING_BEST):
            s = Hipmf(emu_lib)
            assert s.initialize(nn, p, c, ordering=o) == 0
            fill[o] = s.stats()["nnz_l"]
            if o == ORDERING_BEST:
                xs = P.manufactured_solution(nn)
                assert s.factorize(vals) == 0
                assert np.max(np.abs(s.solve(P.csr_matvec(nn, p, c, vals, xs)) - xs)) < 1e-10
            s.close()
        assert fill[ORDERING_BEST] == fill[want], fill
    # through the mirror: the reported name is the winner's
    rows = np.repeat(np.arange(900), np.diff(rrp))
    coo = RS.CooMatrix(900, 900, len(rv))
    coo.put_many(rows.astype(np.int32), rci.astype(np.int32), rv.astype(np.float64))
    par = RS.LinSolParams()
    par.ordering = RS.Ordering.Best
    solver = RS.LinSolver(RS.Genie.Hipmf)
    solver.actual.factorize(coo, par)
    assert solver.actual.stats()["output"]["effective_ordering"] == "Amd"
