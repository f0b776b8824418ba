"""Host-side maximum-product matching + scaling (russell_amd/csrc/matching.cpp) through the C-ABI helper
hipmf_max_product_matching: optimality against scipy's assignment solver on small matrices and the scaling
property |dr_i a_ij dc_j| <= 1 with equality on the matched entries.  No device is needed."""
import numpy as np
import pytest
import scipy.sparse as sp
from scipy.optimize import linear_sum_assignment

from russell_amd import _capi


def _match(A):
    A = A.tocsr()
    A.sort_indices()
    n = A.shape[0]
    lib = _capi.load()
    mrow, dr, dc = np.zeros(n, np.int32), np.zeros(n), np.zeros(n)
    code = lib.hipmf_max_product_matching(n, A.indptr.astype(np.int32), A.indices.astype(np.int32), A.data.astype(np.float64), mrow, dr, dc)
    return code, mrow, dr, dc


@pytest.mark.parametrize("seed,n,density", [(1, 40, 0.15), (2, 90, 0.08), (3, 150, 0.05), (4, 25, 0.5)])
def test_matching_is_a_maximum_product_permutation(seed, n, density):
    rng = np.random.default_rng(seed)
    A = sp.random(n, n, density=density, random_state=seed, format="csr", data_rvs=lambda k: rng.standard_normal(k) * 10.0 ** rng.uniform(-3, 3, k))
    A = (A + sp.diags(1e-6 * rng.standard_normal(n))).tocsr()  # structurally non-singular, numerically useless diagonal
    code, mrow, dr, dc = _match(A)
    assert code == 0
    assert sorted(mrow.tolist()) == list(range(n))
    D = np.abs(A.toarray())
    with np.errstate(divide="ignore"):
        cost = np.where(D > 0, -np.log(D), 1e6)
    r, c = linear_sum_assignment(cost)
    best = -cost[r, c].sum()
    mine = np.sum(np.log(D[mrow, np.arange(n)]))
    assert np.all(D[mrow, np.arange(n)] > 0)
    assert mine >= best - 1e-8 * max(1.0, abs(best))
    S = dr[:, None] * D * dc[None, :]
    assert np.max(S) <= 1.0 + 1e-10
    assert np.allclose(S[mrow, np.arange(n)], 1.0, rtol=0, atol=1e-10)


def test_identity_matching_for_dominant_diagonal():
    n = 60
    A = (sp.diags([-1.0, 4.0, -1.0], [-1, 0, 1], shape=(n, n)) + sp.random(n, n, density=0.05, random_state=5) * 0.1).tocsr()
    code, mrow, dr, dc = _match(A)
    assert code == 0 and np.array_equal(mrow, np.arange(n))


def test_structurally_singular_is_reported():
    A = sp.csr_matrix(np.array([[1.0, 2.0, 0.0], [3.0, 4.0, 0.0], [5.0, 6.0, 0.0]]))
    code, *_ = _match(A)
    assert code == 600  # ERROR_HIPMF_INVALID_MATRIX


@pytest.mark.parametrize("seed,n,density", [(11, 30, 0.2), (12, 80, 0.08)])
def test_paired_matching_of_a_real_equivalent_form_is_the_matching_of_the_moduli(seed, n, density):
    # round 4 (complex twin): rows / columns 2 k, 2 k + 1 = (Re, Im) of complex row / column k; the pairs move as a whole
    rng = np.random.default_rng(seed)
    mag = lambda k: 10.0 ** rng.uniform(-3, 3, k)
    Z = sp.random(n, n, density=density, random_state=seed, format="coo", dtype=np.complex128, data_rvs=lambda k: mag(k) * np.exp(2j * np.pi * rng.random(k)))
    Z = (Z + sp.diags(1e-6 * np.exp(2j * np.pi * rng.random(n)))).tocoo()
    r2 = np.concatenate([2 * Z.row, 2 * Z.row, 2 * Z.row + 1, 2 * Z.row + 1])
    c2 = np.concatenate([2 * Z.col, 2 * Z.col + 1, 2 * Z.col, 2 * Z.col + 1])
    v2 = np.concatenate([Z.data.real, -Z.data.imag, Z.data.imag, Z.data.real])
    K = sp.csr_matrix((v2, (r2, c2)), shape=(2 * n, 2 * n))
    K.sort_indices()
    lib = _capi.load()
    mrow, dr, dc = np.zeros(2 * n, np.int32), np.zeros(2 * n), np.zeros(2 * n)
    assert lib.hipmf_paired_matching(2 * n, K.indptr.astype(np.int32), K.indices.astype(np.int32), K.data.astype(np.float64), mrow, dr, dc) == 0
    assert np.all(mrow[0::2] % 2 == 0) and np.array_equal(mrow[1::2], mrow[0::2] + 1)
    assert np.array_equal(dr[0::2], dr[1::2]) and np.array_equal(dc[0::2], dc[1::2])
    mc = mrow[0::2] // 2
    assert sorted(mc.tolist()) == list(range(n))
    D = np.abs(sp.csr_matrix(Z).toarray())
    with np.errstate(divide="ignore"):
        cost = np.where(D > 0, -np.log(D), 1e6)
    r, c = linear_sum_assignment(cost)
    assert np.sum(np.log(D[mc, np.arange(n)])) >= -cost[r, c].sum() - 1e-8 * max(1.0, abs(cost[r, c].sum()))
    S = dr[0::2, None] * D * dc[None, 0::2]
    assert np.max(S) <= 1.0 + 1e-10 and np.allclose(S[mc, np.arange(n)], 1.0, rtol=0, atol=1e-10)
    assert lib.hipmf_paired_matching(2 * n - 1, K.indptr.astype(np.int32), K.indices.astype(np.int32), K.data.astype(np.float64), mrow, dr, dc) != 0
