"""Synthetic inputs of SURVEY.md section 8(d): the matrices the bench and the parity tests feed to the solver.

All generators return 0-based CSR (int32 row_pointers, int32 col_indices ascending, float64 values),
the layout the C-ABI takes (include/russell_hipmf.h).  Grid index m = i + j*nx (+ k*nx*ny), the
ordering of /root/reference/russell_pde/src/fdm_2d.rs:953-954; 5-point molecule
[2(kx/dx^2+ky/dy^2), -kx/dx^2, -kx/dx^2, -ky/dy^2, -ky/dy^2] of fdm_2d.rs:376-386 with kx=ky=dx=dy=1.
"""
import numpy as np


def _stencil_csr(shape, offsets_coeffs):
    """shape = (nx, ny[, nz]); offsets_coeffs = [((di, dj, dk), value), ...] with Dirichlet elimination."""
    dims = list(shape) + [1] * (3 - len(shape))
    nx, ny, nz = dims
    n = nx * ny * nz
    idx = np.arange(n, dtype=np.int64)
    i = idx % nx
    j = (idx // nx) % ny
    k = idx // (nx * ny)
    rows, cols, vals = [], [], []
    for (di, dj, dk), v in offsets_coeffs:
        ok = (i + di >= 0) & (i + di < nx) & (j + dj >= 0) & (j + dj < ny) & (k + dk >= 0) & (k + dk < nz)
        r = idx[ok]
        rows.append(r)
        cols.append(r + di + dj * nx + dk * nx * ny)
        vals.append(np.full(r.size, float(v)))
    rows = np.concatenate(rows)
    cols = np.concatenate(cols)
    vals = np.concatenate(vals)
    order = np.lexsort((cols, rows))
    rows, cols, vals = rows[order], cols[order], vals[order]
    rp = np.zeros(n + 1, np.int64)
    np.add.at(rp, rows + 1, 1)
    rp = np.cumsum(rp)
    return n, rp.astype(np.int32), cols.astype(np.int32), vals


def poisson2d(nx, ny=None):
    """A = I (x) T + T (x) I, T = tridiag(-1, 2, -1); n = nx*ny, nnz = 5n - 2(nx+ny)."""
    ny = ny or nx
    return _stencil_csr((nx, ny), [((0, 0, 0), 4.0), ((-1, 0, 0), -1.0), ((1, 0, 0), -1.0), ((0, -1, 0), -1.0), ((0, 1, 0), -1.0)])


def poisson3d(nx, ny=None, nz=None):
    ny = ny or nx
    nz = nz or nx
    return _stencil_csr((nx, ny, nz), [((0, 0, 0), 6.0), ((-1, 0, 0), -1.0), ((1, 0, 0), -1.0), ((0, -1, 0), -1.0), ((0, 1, 0), -1.0),
                                        ((0, 0, -1), -1.0), ((0, 0, 1), -1.0)])


def convection_diffusion2d(nx, ny=None, peclet=0.7, seed=20260927, scale_decades=6.0):
    """Unsymmetric stand-in for the ill-conditioned SuiteSparse inputs of config 3:
    5-point convection-diffusion with random row scaling 10^U(-d, d)."""
    ny = ny or nx
    n, rp, ci, v = _stencil_csr((nx, ny), [((0, 0, 0), 4.0), ((-1, 0, 0), -1.0 - peclet), ((1, 0, 0), -1.0 + peclet),
                                            ((0, -1, 0), -1.0 - 0.5 * peclet), ((0, 1, 0), -1.0 + 0.5 * peclet)])
    r = 10.0 ** np.random.default_rng(seed).uniform(-scale_decades, scale_decades, n)
    rows = np.repeat(np.arange(n), np.diff(rp))
    return n, rp, ci, v * r[rows]


def fe_block2d(nx, ny, dof, symmetric, seed=20260927, scale_decades=0.0, shift=None):
    """FE-like stand-in for BASELINE config 3 (bbmat / af_shell10 are not in the tree): nx x ny nodes with `dof` unknowns each,
    every node coupled to its 8 neighbours (Q4 elements) by dense dof x dof blocks -> 9 dof stored entries per interior row
    (dof = 5: 45 per row, bbmat has ~46; dof = 4: 36 per row, af_shell10 has ~35).  Values: uniform random blocks (symmetrised when
    `symmetric`), diagonal shifted to the row's absolute off-diagonal sum times `shift` (default 1.05 unsymmetric: non-singular but
    far from diagonally dominant after the scaling below; 1.0 + 1e-3 symmetric: SPD, condition ~1e3 x grid), then rows scaled by
    10^U(-d, d) (unsymmetric only).  Returns general-storage CSR; symmetric=True also keeps the matrix exactly symmetric."""
    import scipy.sparse as sp
    rng = np.random.default_rng(seed)
    ex = sp.diags([1.0, 1.0, 1.0], [-1, 0, 1], shape=(nx, nx))
    ey = sp.diags([1.0, 1.0, 1.0], [-1, 0, 1], shape=(ny, ny))
    S = sp.kron(ey, ex).tocsr()                       # 9-point node graph, node m = i + j nx
    A = sp.kron(S, np.ones((dof, dof))).tocsr()
    A.sort_indices()
    A.data = rng.uniform(-1.0, 1.0, A.nnz)
    if symmetric:
        A = ((A + A.T) * 0.5).tocsr()
        A.sort_indices()
    n = A.shape[0]
    rows = np.repeat(np.arange(n), np.diff(A.indptr))
    diag = A.indices == rows
    off = np.abs(A.data) * (~diag)
    rowsum = np.zeros(n)
    np.add.at(rowsum, rows, off)
    if shift is None:
        shift = 1.0 + 1e-3 if symmetric else 1.05
    A.data[diag] = shift * rowsum
    v = A.data.copy()
    if not symmetric and scale_decades > 0.0:
        v *= (10.0 ** rng.uniform(-scale_decades, scale_decades, n))[rows]
    return n, A.indptr.astype(np.int32), A.indices.astype(np.int32), v


def lower_triangle(n, rp, ci, v):
    """the stored lower triangle (Sym::YesLower) of a general-storage CSR"""
    rows = np.repeat(np.arange(n, dtype=np.int32), np.diff(rp))
    keep = ci <= rows
    lrp = np.concatenate([[0], np.cumsum(np.bincount(rows[keep], minlength=n))]).astype(np.int32)
    return lrp, np.ascontiguousarray(ci[keep]), np.ascontiguousarray(v[keep])


def brusselator_pattern(npoint, gamma=1.0e4 * 4.0, seed=7):
    """K = gamma*I - J with J's sparsity of the Brusselator-PDE Jacobian
    (/root/reference/russell_ode/src/samples.rs:549-571): ndim = 2*npoint^2, unknowns (u, v) interleaved
    per grid point, 2x2 diagonal blocks plus a periodic 5-point Laplacian on each species."""
    n2 = npoint * npoint
    n = 2 * n2
    rng = np.random.default_rng(seed)
    u = 22.0 * rng.random(n2) + 1.0
    vv = 27.0 * rng.random(n2) + 1.0
    alpha = 0.1 * (npoint * npoint)
    rows, cols, vals = [], [], []
    p = np.arange(n2)
    i, j = p % npoint, p // npoint
    for s, (dself, dother) in enumerate([(2.0 * u * vv - 4.4, u * u), (-u * u, 3.4 - 2.0 * u * vv)]):
        me = 2 * p + s
        other = 2 * p + (1 - s)
        rows += [me, me]
        cols += [me, other]
        vals += [gamma - (dself - 4.0 * alpha), -dother]
        for di, dj in ((1, 0), (-1, 0), (0, 1), (0, -1)):
            q = ((i + di) % npoint) + ((j + dj) % npoint) * npoint
            rows.append(me)
            cols.append(2 * q + s)
            vals.append(np.full(n2, -alpha))
    rows = np.concatenate(rows)
    cols = np.concatenate(cols)
    vals = np.concatenate(vals)
    # merge duplicates (tiny periodic grids can repeat a neighbour)
    key = rows.astype(np.int64) * n + cols
    order = np.argsort(key, kind="stable")
    key, vals = key[order], vals[order]
    uniq, start = np.unique(key, return_index=True)
    vals = np.add.reduceat(vals, start)
    rows, cols = (uniq // n).astype(np.int64), (uniq % n).astype(np.int32)
    rp = np.zeros(n + 1, np.int64)
    np.add.at(rp, rows + 1, 1)
    return n, np.cumsum(rp).astype(np.int32), cols, vals


def csr_matvec(n, rp, ci, v, x):
    """y = A x on the host with numpy (test helper, not a product path)."""
    rows = np.repeat(np.arange(n), np.diff(rp))
    y = np.zeros(n)
    np.add.at(y, rows, v * x[ci])
    return y


def manufactured_solution(n):
    """x*_i = 1 + (i mod 7)/7."""
    return 1.0 + (np.arange(n) % 7) / 7.0
