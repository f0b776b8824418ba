"""Ordering::Amd on the device: the same kernels under a minimum-degree tree (deep, thin) -- a circuit-like pattern without small
separators where it beats the dissection, the 5-point grid where it does not, and the reference's golden 5 x 5 system."""
import numpy as np
import pytest
import scipy.sparse as sp
import scipy.sparse.linalg as spla

from russell_amd import problems as P
from russell_amd.backend import Hipmf

pytestmark = pytest.mark.gpu

ORDERING_DEFAULT, ORDERING_AMD = 0, 3


def _circuit_like(n, seed):
    """Sparse random couplings (1 - 4 per row), a handful of hub rows / columns, strictly diagonally dominant."""
    rng = np.random.default_rng(seed)
    rows = np.repeat(np.arange(n), 3)
    cols = rng.integers(0, n, size=3 * n)
    vals = rng.uniform(-1.0, 1.0, size=3 * n)
    hub = rng.integers(0, n, size=(4, n // 50))
    for h in range(4):
        rows = np.concatenate([rows, np.full(hub.shape[1], h), hub[h]])
        cols = np.concatenate([cols, hub[h], np.full(hub.shape[1], h)])
        vals = np.concatenate([vals, rng.uniform(-1.0, 1.0, size=2 * hub.shape[1])])
    A = sp.coo_matrix((vals, (rows, cols)), shape=(n, n)).tocsr()
    A.setdiag(0.0)
    A.eliminate_zeros()
    A = (A + sp.diags(np.asarray(abs(A).sum(axis=1)).ravel() + 1.0)).tocsr()
    A.sort_indices()
    return A


def test_amd_on_a_circuit_like_pattern_needs_less_fill_than_the_dissection_and_solves():
    n = 40000
    A = _circuit_like(n, 11)
    rp, ci, v = A.indptr.astype(np.int32), A.indices.astype(np.int32), A.data.astype(np.float64)
    xs = P.manufactured_solution(n)
    b = A @ xs
    out = {}
    for name, o in (("nd", ORDERING_DEFAULT), ("amd", ORDERING_AMD)):
        s = Hipmf()
        code = s.initialize(n, rp, ci, ordering=o)
        if code != 0:  # (the dissection of such a pattern may be refused for its size: that is the point of having the alternative)
            out[name] = None
            s.close()
            continue
        assert s.factorize(v) == 0
        x = s.solve(b)
        st = s.stats()
        s.close()
        assert np.max(np.abs(x - xs)) <= 1e-9 * np.max(np.abs(xs)), name
        out[name] = st["nnz_l"]
    assert out["amd"] is not None
    if out["nd"] is not None:
        assert out["amd"] < out["nd"], out
    print("circuit-like n = %d: nnz(L) amd %s, nested dissection %s" % (n, out["amd"], out["nd"]))


def test_amd_on_the_grid_matches_superlu():
    n, rp, ci, v = P.poisson2d(300)
    A = sp.csr_matrix((v, ci, rp), shape=(n, n))
    xs = P.manufactured_solution(n)
    b = A @ xs
    s = Hipmf()
    assert s.initialize(n, rp, ci, ordering=ORDERING_AMD) == 0
    assert s.factorize(v) == 0
    x = s.solve(b)
    s.close()
    xo = spla.splu(A.tocsc()).solve(b)
    assert np.max(np.abs(x - xo)) <= 1e-10 * np.max(np.abs(xo))
    assert np.max(np.abs(x - xs)) <= 1e-9 * np.max(np.abs(xs))


def test_amd_on_the_reference_5x5_system():
    # solver_umfpack.rs:660-671 / the doc example: x = (1, 2, 3, 4, 5)
    rp = np.array([0, 2, 5, 8, 9, 12], dtype=np.int32)
    ci = np.array([0, 1, 0, 2, 4, 1, 2, 3, 2, 1, 2, 4], dtype=np.int32)
    v = np.array([2.0, 3.0, 3.0, 4.0, 6.0, -1.0, -3.0, 2.0, 1.0, 4.0, 2.0, 1.0])
    b = np.array([8.0, 45.0, -3.0, 3.0, 19.0])
    s = Hipmf()
    assert s.initialize(5, rp, ci, ordering=ORDERING_AMD) == 0
    assert s.factorize(v) == 0
    x = s.solve(b)
    s.close()
    assert np.max(np.abs(x - np.array([1.0, 2.0, 3.0, 4.0, 5.0]))) < 1e-13
