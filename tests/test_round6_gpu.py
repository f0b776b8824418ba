"""Round 6 on the device, through the C-ABI: block groups of the many-RHS solves (same bits as one block per launch), the block buffers
prepared ahead of the first blocked solve, the event flags of the validated device, the warning for a kept L D L^T plan."""
import numpy as np
import pytest

from helpers import relative_error_metric
from russell_amd import problems as P
from russell_amd.backend import Hipmf

pytestmark = pytest.mark.gpu


def _solve_many(monkeypatch, groups, n, rp, ci, v, B, sym=False, plain=None, prepare=False):
    monkeypatch.setenv("HIPMF_BLOCK_GROUPS", str(groups))
    if plain is not None:
        monkeypatch.setenv("HIPMF_PLAIN_BAND", str(plain))
    s = Hipmf()
    if sym:
        lrp, lci, lv = P.lower_triangle(n, rp, ci, v)
        assert s.initialize(n, lrp, lci, general_symmetric=True) == 0
        vals = lv
    else:
        assert s.initialize(n, rp, ci) == 0
        vals = v
    if prepare:
        s.prepare_solve_many(B.shape[0])
    assert s.factorize(vals) == 0
    X = s.solve_many(B)
    info = (s.counter("block_groups"), s.counter("fused_fallbacks"))
    s.close()
    if plain is not None:
        monkeypatch.delenv("HIPMF_PLAIN_BAND")
    return X, info


@pytest.mark.parametrize("kind,size,ncols,sym", [("2d", 300, 70, False), ("2d", 257, 64, True), ("3d", 40, 48, False), ("3d", 36, 33, True)])
def test_block_groups_bitwise_on_the_device(monkeypatch, kind, size, ncols, sym):
    # kernels_solve_fused.hpp, SfGroups: the groups of a launch are independent sixteen-column blocks -- per column the bits of a launch
    # that carries the block alone; the plain per-level band against the dependency-driven one: the same arithmetic
    n, rp, ci, v = P.poisson2d(size) if kind == "2d" else P.poisson3d(size)
    B = np.stack([np.random.default_rng([20260927, j]).standard_normal(n) for j in range(ncols)])
    X1, (g1, f1) = _solve_many(monkeypatch, 1, n, rp, ci, v, B, sym=sym, plain=0)
    X4, (g4, f4) = _solve_many(monkeypatch, 4, n, rp, ci, v, B, sym=sym, plain=1, prepare=True)
    assert g1 == 1 and g4 == min(4, (ncols + 15) // 16) and f1 == 0 and f4 == 0
    if ncols % 16 != 1:
        assert np.array_equal(X1, X4)
    else:  # (a last block of ONE column takes the single-column kernels when it travels alone)
        assert np.max(np.abs(X1 - X4)) <= 1e-12 * np.max(np.abs(X1))
    for j in range(0, ncols, 7):
        assert relative_error_metric(n, rp, ci, v, X4[j], B[j]) <= 1e-12


def test_prepare_solve_many_then_first_solve(monkeypatch):
    n, rp, ci, v = P.poisson3d(30)
    s = Hipmf()
    assert s.lib.solver_hipmf_prepare_solve_many(s.h, 32) == 500000  # ERROR_NEED_INITIALIZATION
    assert s.initialize(n, rp, ci) == 0
    s.prepare_solve_many(32)
    assert s.counter("block_groups") == 2
    assert s.factorize(v) == 0
    B = np.stack([np.random.default_rng([5, j]).standard_normal(n) for j in range(32)])
    X = s.solve_many(B)
    for j in range(32):
        assert relative_error_metric(n, rp, ci, v, X[j], B[j]) <= 1e-12
    # a single right-hand side afterwards takes the single-column path and buffers as before
    x = s.solve(B[3])
    assert np.max(np.abs(x - X[3])) <= 1e-12 * np.max(np.abs(x))
    s.close()


def test_event_flags_are_fence_free_only_on_the_validated_device():
    # numeric.cpp: hipEventDisableSystemFence between a handle's streams only on gfx950 under a HIP 7 runtime (this box); HIPMF_EVENT_FENCE=1 -> default flags
    n, rp, ci, v = P.poisson2d(40)
    s = Hipmf()
    assert s.initialize(n, rp, ci) == 0
    assert s.counter("event_fence_free") == 1
    assert s.factorize(v) == 0
    x = s.solve(P.csr_matvec(n, rp, ci, v, np.ones(n)))
    assert np.max(np.abs(x - 1.0)) < 1e-12
    s.close()


def test_kept_ldlt_plan_reports_weak_diagonal_on_the_device():
    from test_sym_indefinite_cpu import _csr, saddle_point
    A, L = saddle_point(40, 80, seed=4)
    n = A.shape[0]
    rp, ci, v = _csr(L)
    s = Hipmf()
    assert s.initialize(n, rp, ci, general_symmetric=True) == 0  # no values: L D L^T plan
    s.factorize(v)
    assert s.counter("sym_weak_diagonal") == 1 and s.counter("sym_expanded") == 0
    s.close()
    s = Hipmf()
    assert s.initialize(n, rp, ci, general_symmetric=True, values=v) == 0  # with values: mirrored + matched
    assert s.counter("sym_expanded") == 1 and s.factorize(v) == 0 and s.counter("sym_weak_diagonal") == 0
    x = s.solve(A @ np.ones(n))
    assert np.max(np.abs(x - 1.0)) < 1e-9
    s.close()


def test_two_processes_share_the_device_behind_the_gate():
    # an in-process mutex does not order two PROCESSES on one GPU.  Round 6: the gate continues as an advisory file
    # lock keyed by the device's PCI bus id; both workers finish without a hand-off time-out, and they did meet at the gate.
    import json
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    out = subprocess.run([sys.executable, os.path.join(root, "tools", "soak_two_processes.py"), "400", "600"], capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stderr[-500:]
    d = json.loads([ln for ln in out.stdout.splitlines() if ln.startswith("{")][-1])
    assert len(d["workers"]) == 2
    for w in d["workers"]:
        assert "error" not in w, d
        assert w["fallbacks"] == 0 and w["max_error"] < 1e-9 and w["solves"] >= 100
    assert sum(w["gate_waits"] for w in d["workers"]) > 0
