"""Kernel-logic regression on the CPU: the HIP kernels compiled against tools/hipemu (a development-only emulator:
fibers for __syncthreads, wave shuffles, the MFMA lane maps) and driven through the same C-ABI.

This is NOT parity evidence (parity is measured on a real MI355X by the -m gpu tests) and the emulated library is never
part of the product; it catches indexing / barrier / assembly-map mistakes before a GPU run."""
import numpy as np

from russell_amd import problems as P
from russell_amd.backend import Hipmf


def _solve(lib, n, rp, ci, v, b, **kw):
    s = Hipmf(lib)
    assert s.initialize(n, rp, ci, **kw) == 0
    code = s.factorize(v, compute_determinant=True)
    x = s.solve(b)
    st = s.stats()
    return s, code, x, st


def test_emulated_small_front_path(emu_lib):
    n, rp, ci, v = P.poisson2d(12, 9)
    xs = P.manufactured_solution(n)
    s, code, x, st = _solve(emu_lib, n, rp, ci, v, P.csr_matvec(n, rp, ci, v, xs))
    assert code == 0 and st["max_front"] <= 64
    assert np.max(np.abs(x - xs)) < 1e-12
    s.close()


def test_emulated_tiled_augmented_path(emu_lib):
    n, rp, ci, v = P.poisson2d(44, 40)  # top separators give fronts > 64: k_panel / k_update (MFMA) / k_fwd_big / k_bwd_big
    xs = P.manufactured_solution(n)
    s, code, x, st = _solve(emu_lib, n, rp, ci, v, P.csr_matvec(n, rp, ci, v, xs))
    assert code == 0 and st["max_front"] > 64
    assert np.max(np.abs(x - xs)) < 1e-11
    s.close()


def test_emulated_pivoting_and_determinant(emu_lib):
    dense = np.array([[2.0, 3.0, 0, 0, 0], [3.0, 0, 4.0, 0, 6.0], [0, -1.0, -3.0, 2.0, 0], [0, 0, 1.0, 0, 0], [0, 4.0, 2.0, 0, 1.0]])
    r, c = np.nonzero(dense)
    rp = np.concatenate([[0], np.cumsum(np.bincount(r, minlength=5))]).astype(np.int32)
    s, code, x, st = _solve(emu_lib, 5, rp, c.astype(np.int32), dense[r, c], np.array([8.0, 45.0, -3.0, 3.0, 19.0]))
    assert code == 0 and np.max(np.abs(x - np.arange(1, 6))) < 1e-13
    assert abs(s.det_coefficient * 10.0 ** s.det_exponent - 114.0) < 1e-10
    s.close()


def test_emulated_value_map_refresh_with_duplicates(emu_lib):
    # COO triplets with duplicates -> CSR values on the "device": same factor as handing over the summed CSR values
    n, rp, ci, v = P.poisson2d(10, 8)
    rng = np.random.default_rng(5)
    nnz = int(rp[-1])
    # every CSR entry is split into 1..3 triplets, shuffled
    owner = np.concatenate([np.full(rng.integers(1, 4), j) for j in range(nnz)])
    rng.shuffle(owner)
    order = np.argsort(owner, kind="stable")
    seg_ptr = np.concatenate([[0], np.cumsum(np.bincount(owner, minlength=nnz))]).astype(np.int32)
    seg_idx = order.astype(np.int32)
    xs = P.manufactured_solution(n)
    s = Hipmf(emu_lib)
    assert s.initialize(n, rp, ci) == 0
    assert s.set_value_map(seg_ptr, seg_idx) == 0
    for step in range(3):
        vals = v * (1.0 + 0.1 * step) + (step > 0) * 0.01 * rng.standard_normal(nnz) * (ci == np.repeat(np.arange(n), np.diff(rp)))
        parts = rng.standard_normal(owner.size)
        # make the triplets of every entry sum to its value (last triplet of each segment takes the remainder)
        trip = parts.copy()
        for j in range(nnz):
            idx = seg_idx[seg_ptr[j]:seg_ptr[j + 1]]
            trip[idx[-1]] = vals[j] - np.sum(trip[idx[:-1]])
        summed = np.array([np.sum(np.concatenate([[0.0], trip[seg_idx[seg_ptr[j]:seg_ptr[j + 1]]]])) for j in range(nnz)])
        b = P.csr_matvec(n, rp, ci, summed, xs)
        assert s.factorize_mapped(trip) == 0
        x1 = s.solve(b)
        assert s.factorize(summed) == 0
        x2 = s.solve(b)
        assert np.max(np.abs(x1 - xs)) < 1e-9
        assert np.max(np.abs(x1 - x2)) < 1e-12
    # invalid maps are refused
    bad = seg_ptr.copy()
    bad[-1] += 1
    assert s.set_value_map(bad, seg_idx) == 803
    s.close()


def test_emulated_grouped_updates_and_chain_split(emu_lib, monkeypatch):
    # tiled path with 4 / 8 / 16 panels per pass over the trailing matrix, and big supernodes split into chains: same solution
    # (to rounding) and same determinant as the default two-panel schedule
    # (the amalgamation of rounds 1 - 4: large supernodes merged whatever the front size, so that this small problem has a front with
    #  more than 128 pivots -- the default since late round 4 merges them only into fronts of 2 048 rows and more)
    monkeypatch.setenv("HIPMF_RELAX", "4,16,48,0.8,0.1,0.05")
    monkeypatch.setenv("HIPMF_RELAX_BIG", "0")
    n, rp, ci, v = P.poisson3d(13)
    xs = P.manufactured_solution(n)
    b = P.csr_matvec(n, rp, ci, v, xs)
    s0, code, x0, st0 = _solve(emu_lib, n, rp, ci, v, b)
    assert code == 0 and st0["max_pivots"] > 128
    det0 = (s0.det_coefficient, s0.det_exponent)
    s0.close()
    for env in ({"HIPMF_UPD_G4": "65", "HIPMF_UPD_G8": "100", "HIPMF_UPD_G16": "150"},
                {"HIPMF_SPLIT_PIVOTS": "64", "HIPMF_UPD_G4": "65", "HIPMF_UPD_G8": "65"}):
        for k, val in env.items():
            monkeypatch.setenv(k, val)
        s, code, x, st = _solve(emu_lib, n, rp, ci, v, b)
        assert code == 0
        if "HIPMF_SPLIT_PIVOTS" in env:
            assert st["nsuper"] > st0["nsuper"] and st["max_pivots"] <= 64
        assert np.max(np.abs(x - xs)) < 1e-12 and np.max(np.abs(x - x0)) < 1e-13
        assert s.det_exponent == det0[1] and abs(s.det_coefficient - det0[0]) < 1e-10
        s.close()
        for k in env:
            monkeypatch.delenv(k)


def test_emulated_hub_vertices_are_ordered_last(emu_lib, monkeypatch):
    # a grid plus hub rows / columns tied to a third of the unknowns (supply nets of circuit matrices): the hubs are taken out of the
    # dissection and numbered last -- without that every level structure collapses to three levels and the fronts explode
    import scipy.sparse as sp

    n0, rp, ci, v = P.poisson2d(40)
    rng = np.random.default_rng(1)
    n = n0 + 2
    B = sp.lil_matrix((n, n))
    B[:n0, :n0] = sp.csr_matrix((v, ci, rp), shape=(n0, n0))
    for h in range(2):
        idx = rng.choice(n0, n0 // 3, replace=False)  # degree 533 > max(32, 10 sqrt(n)) = 400
        B[n0 + h, idx] = 0.01
        B[idx, n0 + h] = 0.02
        B[n0 + h, n0 + h] = 5.0
    B = B.tocsr()
    B.sort_indices()
    rp2, ci2, v2 = B.indptr.astype(np.int32), B.indices.astype(np.int32), B.data.astype(np.float64)
    xs = P.manufactured_solution(n)
    b = B @ xs
    s, code, x, st = _solve(emu_lib, n, rp2, ci2, v2, b)
    assert code == 0 and np.max(np.abs(x - xs)) < 1e-11
    perm = s.permutation()
    assert sorted(perm[-2:].tolist()) == [n0, n0 + 1]  # the hubs come last
    flops_deferred = st["flops"]
    s.close()
    monkeypatch.setenv("HIPMF_DENSE_ROWS", "0")
    s = Hipmf(emu_lib)
    assert s.initialize(n, rp2, ci2) == 0
    assert s.stats()["flops"] > 20 * flops_deferred  # what the deferral saves
    s.close()


def test_emulated_star_with_many_one_entry_children(emu_lib):
    # a hub row / column and 300 unknowns tied only to it: the hub front has 300 children with one contribution entry each
    # (batches of 64 descriptors, one lane per child, added in child order) -- factorisation and forward solve
    import scipy.sparse as sp

    n = 301
    rng = np.random.default_rng(4)
    A = sp.lil_matrix((n, n))
    A.setdiag(3.0 + rng.random(n))
    A[0, 1:] = 1e-2 * (1.0 + rng.random(n - 1))
    A[1:, 0] = 2e-2 * (1.0 + rng.random((n - 1, 1)))
    A = A.tocsr()
    A.sort_indices()
    xs = P.manufactured_solution(n)
    b = A @ xs
    s, code, x, st = _solve(emu_lib, n, A.indptr.astype(np.int32), A.indices.astype(np.int32), A.data.astype(np.float64), b)
    assert code == 0 and st["nlevels"] == 2 and st["nsuper"] >= 300
    want = np.linalg.solve(A.toarray(), b)
    assert np.max(np.abs(x - want)) < 1e-13
    sign, logdet = np.linalg.slogdet(A.toarray())
    assert abs(np.log10(abs(s.det_coefficient)) + s.det_exponent - logdet / np.log(10.0)) < 1e-10
    X = s.solve_many(np.tile(b[None, :], (3, 1)))
    assert np.max(np.abs(X - want[None, :])) < 1e-13
    s.close()
