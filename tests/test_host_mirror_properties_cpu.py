"""Property tests of the host mirror's COO -> CSC / CSR conversions (csc_matrix.rs:365-505, csr_matrix.rs:359-480) and
matrix-vector products against scipy on random triplet lists with duplicates, all three storage kinds."""
import numpy as np
import scipy.sparse as sp
from hypothesis import given, settings
from hypothesis import strategies as st

from russell_amd.sparse import CooMatrix, CscMatrix, CsrMatrix, Sym


@st.composite
def triplet_lists(draw):
    n = draw(st.integers(1, 12))
    m = n if draw(st.booleans()) else draw(st.integers(1, 12))
    sym = draw(st.sampled_from([Sym.No, Sym.YesFull, Sym.YesLower])) if m == n else Sym.No
    k = draw(st.integers(1, 40))
    ii = draw(st.lists(st.integers(0, n - 1), min_size=k, max_size=k))
    jj = draw(st.lists(st.integers(0, m - 1), min_size=k, max_size=k))
    vv = draw(st.lists(st.floats(-8.0, 8.0, allow_nan=False, width=32), min_size=k, max_size=k))
    if sym == Sym.YesLower:
        ii, jj = [max(a, b) for a, b in zip(ii, jj)], [min(a, b) for a, b in zip(ii, jj)]
    return n, m, sym, ii, jj, vv


@settings(max_examples=150, deadline=None)
@given(triplet_lists())
def test_conversions_sum_duplicates_and_sort(data):
    n, m, sym, ii, jj, vv = data
    coo = CooMatrix(n, m, len(ii), sym)
    for i, j, a in zip(ii, jj, vv):
        coo.put(i, j, a)
    ref = sp.coo_matrix((vv, (ii, jj)), shape=(n, m)).tocsr()  # sums duplicates
    ref.sort_indices()
    rp, ci, vx = CsrMatrix.from_coo(coo).arrays()
    # the pattern keeps entries that cancel to zero (the reference's conversion only adds, it never drops)
    pat = sp.coo_matrix((np.ones(len(ii)), (ii, jj)), shape=(n, m)).tocsr()
    pat.sort_indices()
    assert rp.tolist() == pat.indptr.tolist() and ci.tolist() == pat.indices.tolist()
    got = sp.csr_matrix((vx, ci, rp), shape=(n, m)).toarray()
    assert np.allclose(got, ref.toarray(), rtol=0, atol=1e-12)
    cp, ri, cx = CscMatrix.from_coo(coo).arrays()
    patc = pat.tocsc()
    patc.sort_indices()
    assert cp.tolist() == patc.indptr.tolist() and ri.tolist() == patc.indices.tolist()
    assert np.allclose(sp.csc_matrix((cx, ri, cp), shape=(n, m)).toarray(), ref.toarray(), rtol=0, atol=1e-12)
    # transpositions agree with the direct conversions
    assert all(np.array_equal(a, b) for a, b in zip(CscMatrix.from_csr(CsrMatrix.from_coo(coo)).arrays(), (cp, ri, cx)))
    # products: triangular storage is mirrored, every layout gives the same vector
    full = ref.toarray()
    if sym == Sym.YesLower:
        full = full + np.tril(full, -1).T
    u = np.linspace(-1.0, 1.0, m)
    want = full @ u
    assert np.allclose(coo.mat_vec_mul(u), want, atol=1e-12)
    assert np.allclose(CsrMatrix.from_coo(coo).mat_vec_mul(u, n), want, atol=1e-12)
    assert np.allclose(CscMatrix.from_coo(coo).mat_vec_mul(u, n), want, atol=1e-12)
    assert np.allclose(coo.to_dense(), full, atol=1e-12)
    assert np.allclose(coo.mat_t_vec_mul(np.linspace(1.0, 2.0, n)), full.T @ np.linspace(1.0, 2.0, n), atol=1e-12)
