"""Round 5 on the device: the data-tagged hand-offs and the wave fronts of the triangular solves at BASELINE config 2's size, the
per-device gate under two handles on two host threads (the way russell_ode's Radau5 drives a backend, radau5.rs:270-296,306-326),
and the cross-stream events with and without system-scope fences."""
import ctypes as C
import threading
import time

import numpy as np
import pytest

from russell_amd import _capi
from russell_amd import problems as P
from russell_amd.backend import Hipmf

pytestmark = pytest.mark.gpu


def _metric(n, rp, ci, v, x, b):
    import scipy.sparse as sp
    A = sp.csr_matrix((v, ci, rp), shape=(n, n))
    return float(np.max(np.abs(A @ x - b)) / (np.max(np.abs(v)) + 1.0))


def test_tagged_handoffs_and_wave_fronts_at_config2_size(monkeypatch):
    # the default solve path of the 1M-DOF matrix: tagged hand-offs above the wave-subtrees, wave fronts on the levels right above them;
    # against the completion counters (bit for bit with the same task shapes) and against the slab tasks (to rounding)
    n, rp, ci, v = P.poisson2d(1000)
    b = np.random.default_rng(5).standard_normal(n)
    got = {}
    for tag, env in (("default", {}), ("counters", {"HIPMF_TAG_SOLVE": "0"}), ("slabs", {"HIPMF_WAVE_FRONTS": "0"})):
        for k, val in env.items():
            monkeypatch.setenv(k, val)
        s = Hipmf()
        assert s.initialize(n, rp, ci, refinement_nstep=0) == 0
        assert s.factorize(v) == 0
        x = s.solve(b)
        x2 = s.solve(b)
        got[tag] = (x, s.counter("tagged_solve"), s.counter("wave_fronts"), s.counter("fused_fallbacks"))
        assert np.array_equal(x, x2)
        s.close()
        for k in env:
            monkeypatch.delenv(k)
    assert got["default"][1:] == (1, got["default"][2], 0) and got["default"][2] > 2000
    assert got["counters"][1] == 0 and got["slabs"][2] == 0
    assert got["counters"][3] == 0 and got["slabs"][3] == 0
    assert np.array_equal(got["default"][0], got["counters"][0])
    assert np.max(np.abs(got["default"][0] - got["slabs"][0])) <= 1e-12 * np.max(np.abs(got["slabs"][0]))
    assert _metric(n, rp, ci, v, got["default"][0], b) <= 1e-10


def _complex_shifted_laplacian(nx):
    """K = (alpha + i beta) I - J on an nx x nx grid (the shape of Radau5's K_comp, radau5.rs:206-262), 0-based complex CSR"""
    n, rp, ci, v = P.poisson2d(nx)
    rows = np.repeat(np.arange(n), np.diff(rp))
    z = -v.astype(complex)
    z[rows == ci] += (3.0 + 2.0j)
    return n, rp, ci, z


def test_two_handles_on_two_threads_share_the_device_without_fallbacks():
    # nothing serialised two dependency-driven launches on one device, and two handles on two threads is how Radau5
    # drives this backend.  Now a per-device gate lets one handle's solve in at a time: 200 solves each of a real 1M-DOF system and of a
    # complex 250k-unknown system (real-equivalent order 500k: task lists that do NOT fit the device whole), concurrently; no solve may
    # fall back, none may stall.
    lib = _capi.load()
    n, rp, ci, v = P.poisson2d(1000)
    br = np.random.default_rng(1).standard_normal(n)
    nz, zrp, zci, zv = _complex_shifted_laplacian(500)
    zb = np.random.default_rng(2).standard_normal(2 * nz)
    NSOLVE = 200
    out = {}

    def real_side():
        s = Hipmf()
        assert s.initialize(n, rp, ci) == 0
        assert s.factorize(v) == 0
        ts = []
        for it in range(NSOLVE):
            if it % 50 == 25:
                assert s.factorize(v) == 0  # (factorisations of one handle run beside the other handle's solves: not gated)
            t0 = time.perf_counter()
            x = s.solve(br)
            ts.append(time.perf_counter() - t0)
        out["real"] = (ts, s.counter("fused_fallbacks"), s.counter("chain_fallbacks"), s.counter("gate_waits"), _metric(n, rp, ci, v, x, br))
        s.close()

    def complex_side():
        h = lib.complex_solver_hipmf_new()
        assert h
        vals = np.ascontiguousarray(np.stack([zv.real, zv.imag], axis=1).ravel())
        assert lib.complex_solver_hipmf_initialize(h, 0, 1, -1.0, -1, 0, 0, nz, np.ascontiguousarray(zrp, dtype=np.int32),
                                                   np.ascontiguousarray(zci, dtype=np.int32), vals.ctypes.data_as(C.c_void_p)) == 0
        i32 = C.c_int32
        eo, es, npert = i32(0), i32(0), i32(0)
        rc, dr, di, de = C.c_double(0), C.c_double(0), C.c_double(0), C.c_double(0)
        assert lib.complex_solver_hipmf_factorize(h, C.byref(eo), C.byref(es), C.byref(npert), C.byref(rc), C.byref(dr), C.byref(di), C.byref(de), 0, 0, vals) == 0
        x = np.zeros(2 * nz)
        ts = []
        for _ in range(NSOLVE):
            t0 = time.perf_counter()
            assert lib.complex_solver_hipmf_solve(h, x, zb, 0) == 0
            ts.append(time.perf_counter() - t0)
        import scipy.sparse as sp
        A = sp.csr_matrix((zv, zci, zrp), shape=(nz, nz))
        xc, bc = x[0::2] + 1j * x[1::2], zb[0::2] + 1j * zb[1::2]
        res = float(np.max(np.abs(A @ xc - bc)) / (np.max(np.abs(zv)) + 1.0))
        out["complex"] = (ts, lib.complex_solver_hipmf_get_counter(h, 2), lib.complex_solver_hipmf_get_counter(h, 7),
                          lib.complex_solver_hipmf_get_counter(h, 11), res)
        lib.complex_solver_hipmf_drop(h)

    th = [threading.Thread(target=real_side), threading.Thread(target=complex_side)]
    for t in th:
        t.start()
    for t in th:
        t.join()
    assert sorted(out) == ["complex", "real"]
    waits = 0
    for tag, (ts, fused_fb, chain_fb, gate_waits, res) in out.items():
        assert fused_fb == 0 and chain_fb == 0, (tag, fused_fb, chain_fb)
        assert res <= 1e-10, (tag, res)
        med = float(np.median(ts))
        assert max(ts) < 20.0 * med, (tag, max(ts), med)
        waits += gate_waits
    print("concurrent solves: real median %.2f ms max %.2f ms, complex median %.2f ms max %.2f ms, gate waits %d" %
          (1e3 * np.median(out["real"][0]), 1e3 * max(out["real"][0]), 1e3 * np.median(out["complex"][0]), 1e3 * max(out["complex"][0]), waits))
    assert waits > 0  # (the two threads did meet at the gate: the test exercised what it is about)


def test_stream_events_with_and_without_system_fences_give_the_same_factor(monkeypatch):
    # the events that order a handle's streams among themselves are created without system-scope fences
    # (hipEventDisableSystemFence; HIPMF_EVENT_FENCE=1 restores the default flags).  Same factors, same solutions, bit for bit.
    n, rp, ci, v = P.poisson2d(400, 380)
    b = np.random.default_rng(3).standard_normal(n)
    got = []
    for fence in ("0", "1"):
        monkeypatch.setenv("HIPMF_EVENT_FENCE", fence)
        s = Hipmf()
        assert s.initialize(n, rp, ci, refinement_nstep=0) == 0
        xs_, dets = [], []
        for rep in range(3):
            assert s.factorize(v * (1.0 + 0.25 * rep), compute_determinant=True) == 0
            xs_.append(s.solve(b))
            dets.append((s.det_coefficient, s.det_exponent))
        got.append((xs_, dets, s.num_perturbed))
        s.close()
    monkeypatch.delenv("HIPMF_EVENT_FENCE")
    for a, bb in zip(got[0][0], got[1][0]):
        assert np.array_equal(a, bb)
    assert got[0][1] == got[1][1] and got[0][2] == got[1][2] == 0


def test_blocked_solves_of_tiny_and_disconnected_matrices():
    # found by tools/fuzz.py on the device (round 5): a leaf WITHOUT off-diagonal rows (a front that is root and leaf at once: matrices of
    # <= 16 unknowns, disconnected blocks) made the backward leaf kernel read the row structure past its end -- a memory fault on the GPU,
    # invisible on the CPU emulator.  Blocks of 2 ... 20 right-hand sides on such matrices, against dense LAPACK.
    import scipy.sparse as sp
    rng = np.random.default_rng(77)
    for n, blocks in ((1, 1), (2, 1), (5, 1), (16, 1), (12, 3), (40, 8), (96, 6), (300, 30)):
        A = sp.lil_matrix((n, n))
        size = n // blocks
        for bidx in range(blocks):
            lo, hi = bidx * size, (n if bidx == blocks - 1 else (bidx + 1) * size)
            A[lo:hi, lo:hi] = rng.uniform(-1, 1, (hi - lo, hi - lo)) + 4.0 * np.eye(hi - lo)
        A = sp.csr_matrix(A)
        A.sort_indices()
        rp, ci, v = A.indptr.astype(np.int32), A.indices.astype(np.int32), A.data.astype(np.float64)
        s = Hipmf()
        assert s.initialize(n, rp, ci) == 0
        assert s.factorize(v) == 0
        for nrhs in (2, 9, 20):
            XS = rng.standard_normal((nrhs, n))
            B = np.array([A @ XS[j] for j in range(nrhs)])
            X = s.solve_many(B)
            assert np.max(np.abs(X - XS)) <= 1e-10 * max(1.0, np.max(np.abs(XS))), (n, blocks, nrhs)
        assert s.counter("fused_fallbacks") == 0
        s.close()


def test_split_dot_products_of_blocked_backward_slabs_on_the_device(monkeypatch):
    # levels of few slabs with long dot products (the top of a 3D factor): a backward slab of the blocked instances is dealt to Q tasks, the
    # last one to arrive adds the partial sums in the order of the parts (k_bwd_fused).  60^3 with the default thresholds (the top levels
    # qualify) and with every level forced: equal to rounding with the unsplit tasks, the same bits from solve to solve (whoever arrives
    # last), no fallback; L D L^T fronts too.
    n, rp, ci, v = P.poisson3d(60)
    lrp, lci, lv = P.lower_triangle(n, rp, ci, v)
    rng = np.random.default_rng(60)
    XS = rng.standard_normal((32, n))
    B = np.array([P.csr_matvec(n, rp, ci, v, XS[j]) for j in range(32)])
    for arrays, kw in (((rp, ci, v), {}), ((lrp, lci, lv), {"general_symmetric": True})):
        got = {}
        # ("default" = the thresholds this test was written with, one block per launch: since round 6 the row threshold is 2 048 x the planned
        #  block groups -- with four groups per launch a 60^3 factor has no front that qualifies)
        for tag, env in (("off", {"HIPMF_SPLIT_TASKS": "0"}), ("default", {"HIPMF_SPLIT_MINLEN": "2048"}), ("forced", {"HIPMF_SPLIT_TASKS": "1000000", "HIPMF_SPLIT_MINLEN": "256"})):
            for k, val in env.items():
                monkeypatch.setenv(k, val)
            s = Hipmf()
            assert s.initialize(n, arrays[0], arrays[1], refinement_nstep=0, **kw) == 0
            assert s.factorize(arrays[2]) == 0
            X = s.solve_many(B)
            for _ in range(3):
                assert np.array_equal(X, s.solve_many(B))
            X7 = s.solve_many(B[:7])  # (the 8-column instance: chunks of 512 positions)
            assert np.array_equal(X7, s.solve_many(B[:7]))
            assert np.max(np.abs(X7 - XS[:7])) <= 1e-10 * np.max(np.abs(XS))
            got[tag] = (X, s.counter("split_slabs"), s.counter("fused_fallbacks"))
            s.close()
            for k in env:
                monkeypatch.delenv(k)
        assert got["off"][1] == 0 and got["default"][1] > 0 and got["forced"][1] > got["default"][1]
        for tag in ("off", "default", "forced"):
            assert got[tag][2] == 0
            assert np.max(np.abs(got[tag][0] - XS)) <= 1e-10 * np.max(np.abs(XS))
            assert np.max(np.abs(got[tag][0] - got["off"][0])) <= 1e-12 * np.max(np.abs(got["off"][0]))


def test_host_solves_with_the_same_buffers_skip_the_staging_copy():
    # a caller that comes back with the SAME (x, rhs) host buffers as in its last call is copied directly, any other call goes through
    # the pinned staging buffer (numeric.cpp, Solver::solve): the same bits either way, and a change of buffers in between is harmless
    n, rp, ci, v = P.poisson2d(300, 280)
    rng = np.random.default_rng(9)
    b1, b2 = rng.standard_normal(n), rng.standard_normal(n)
    s = Hipmf()
    assert s.initialize(n, rp, ci) == 0
    assert s.factorize(v) == 0
    lib = s.lib
    x1, x2 = np.zeros(n), np.zeros(n)
    ref1, ref2 = s.solve(b1), s.solve(b2)  # (fresh buffers: staged)
    for rep in range(3):  # first call of a pair staged, the following ones direct
        assert lib.solver_hipmf_solve(s.h, x1, b1, 0) == 0
        assert np.array_equal(x1, ref1)
    assert lib.solver_hipmf_solve(s.h, x2, b2, 0) == 0 and np.array_equal(x2, ref2)
    b1[:] = b2  # same buffers, new contents
    assert lib.solver_hipmf_solve(s.h, x1, b1, 0) == 0 and np.array_equal(x1, ref2)
    assert lib.solver_hipmf_solve(s.h, x1, b1, 0) == 0 and np.array_equal(x1, ref2)
    assert _metric(n, rp, ci, v, x1, b2) <= 1e-10
    s.close()
