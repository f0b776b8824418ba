"""ctypes loader for oracle/liboracle.so -- the CPU checker (test infrastructure only)."""
import ctypes as C
import os
import subprocess

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
_SO = os.path.join(ROOT, "oracle", "liboracle.so")

i32p = np.ctypeslib.ndpointer(dtype=np.int32, flags="C_CONTIGUOUS")
f64p = np.ctypeslib.ndpointer(dtype=np.float64, flags="C_CONTIGUOUS")


def _load():
    src = os.path.join(ROOT, "oracle", "oracle.c")
    if not os.path.exists(_SO) or os.path.getmtime(_SO) < os.path.getmtime(src):
        subprocess.check_call(["make", "-C", os.path.join(ROOT, "oracle")], stdout=subprocess.DEVNULL)
    lib = C.CDLL(_SO)
    lib.oracle_coo_to_csc.restype = C.c_int32
    lib.oracle_coo_to_csc.argtypes = [C.c_int32, C.c_int32, C.c_int32, i32p, i32p, f64p, i32p, i32p, f64p]
    lib.oracle_coo_to_csr.restype = C.c_int32
    lib.oracle_coo_to_csr.argtypes = [C.c_int32, C.c_int32, C.c_int32, i32p, i32p, f64p, i32p, i32p, f64p]
    lib.oracle_coo_matvec.restype = None
    lib.oracle_coo_matvec.argtypes = [C.c_int32, C.c_int32, i32p, i32p, f64p, C.c_int32, C.c_double, f64p, f64p]
    lib.oracle_csr_matvec.restype = None
    lib.oracle_csr_matvec.argtypes = [C.c_int32, i32p, i32p, f64p, C.c_int32, C.c_double, f64p, f64p]
    lib.oracle_csc_matvec.restype = None
    lib.oracle_csc_matvec.argtypes = [C.c_int32, C.c_int32, i32p, i32p, f64p, C.c_int32, C.c_double, f64p, f64p]
    lib.oracle_verify.restype = C.c_int32
    lib.oracle_verify.argtypes = [C.c_int32, C.c_int32, i32p, i32p, f64p, C.c_int32, f64p, f64p, f64p]
    lib.oracle_lu_factor.restype = C.c_void_p
    lib.oracle_lu_factor.argtypes = [C.c_int32, i32p, i32p, f64p, C.c_void_p, C.c_int32, C.c_double]
    lib.oracle_lu_free.restype = None
    lib.oracle_lu_free.argtypes = [C.c_void_p]
    lib.oracle_lu_solve.restype = C.c_int32
    lib.oracle_lu_solve.argtypes = [C.c_void_p, f64p, f64p, C.c_int32]
    lib.oracle_lu_status.restype = C.c_int32
    lib.oracle_lu_status.argtypes = [C.c_void_p]
    lib.oracle_lu_nnz_l.restype = C.c_int64
    lib.oracle_lu_nnz_l.argtypes = [C.c_void_p]
    lib.oracle_lu_nnz_u.restype = C.c_int64
    lib.oracle_lu_nnz_u.argtypes = [C.c_void_p]
    lib.oracle_lu_determinant.restype = None
    lib.oracle_lu_determinant.argtypes = [C.c_void_p, C.POINTER(C.c_double), C.POINTER(C.c_double)]
    lib.oracle_lu_rcond.restype = C.c_double
    lib.oracle_lu_rcond.argtypes = [C.c_void_p]
    lib.oracle_fdm_lmm.restype = C.c_int64
    lib.oracle_fdm_lmm.argtypes = [C.c_int32] * 7 + [C.c_void_p] + [C.c_double] * 7 + [i32p, i32p, f64p, C.POINTER(C.c_int64)]
    lib.oracle_fdm_sps.restype = C.c_int64
    lib.oracle_fdm_sps.argtypes = [C.c_int32] * 7 + [C.c_void_p] + [C.c_double] * 7 + [i32p, i32p, i32p, f64p, i32p, i32p, f64p, C.POINTER(C.c_int64)]
    return lib


LIB = _load()


def _i32(a):
    return np.ascontiguousarray(a, dtype=np.int32)


def _f64(a):
    return np.ascontiguousarray(a, dtype=np.float64)


def coo_to_csc(nrow, ncol, ai, aj, ax):
    ai, aj, ax = _i32(ai), _i32(aj), _f64(ax)
    nnz = len(ax)
    bp = np.zeros(ncol + 1, np.int32)
    bi = np.zeros(max(nnz, 1), np.int32)
    bx = np.zeros(max(nnz, 1), np.float64)
    final = LIB.oracle_coo_to_csc(nrow, ncol, nnz, ai, aj, ax, bp, bi, bx)
    assert final >= 0
    return bp, bi[:final].copy(), bx[:final].copy()


def coo_to_csr(nrow, ncol, ai, aj, ax):
    ai, aj, ax = _i32(ai), _i32(aj), _f64(ax)
    nnz = len(ax)
    bp = np.zeros(nrow + 1, np.int32)
    bj = np.zeros(max(nnz, 1), np.int32)
    bx = np.zeros(max(nnz, 1), np.float64)
    final = LIB.oracle_coo_to_csr(nrow, ncol, nnz, ai, aj, ax, bp, bj, bx)
    assert final >= 0
    return bp, bj[:final].copy(), bx[:final].copy()


def coo_matvec(nrow, ai, aj, ax, u, sym_triangular=False, alpha=1.0):
    v = np.zeros(nrow)
    LIB.oracle_coo_matvec(nrow, len(ax), _i32(ai), _i32(aj), _f64(ax), int(sym_triangular), alpha, _f64(u), v)
    return v


def csr_matvec(nrow, rp, cj, ax, u, sym_triangular=False, alpha=1.0):
    v = np.zeros(nrow)
    LIB.oracle_csr_matvec(nrow, _i32(rp), _i32(cj), _f64(ax), int(sym_triangular), alpha, _f64(u), v)
    return v


def csc_matvec(nrow, ncol, cp, ri, ax, u, sym_triangular=False, alpha=1.0):
    v = np.zeros(nrow)
    LIB.oracle_csc_matvec(nrow, ncol, _i32(cp), _i32(ri), _f64(ax), int(sym_triangular), alpha, _f64(u), v)
    return v


def verify(nrow, ai, aj, ax, x, rhs, sym_triangular=False):
    out = np.zeros(4)
    st = LIB.oracle_verify(nrow, len(ax), _i32(ai), _i32(aj), _f64(ax), int(sym_triangular), _f64(x), _f64(rhs), out)
    assert st == 0
    return dict(max_abs_a=out[0], max_abs_ax=out[1], max_abs_diff=out[2], relative_error=out[3])


class OracleLU:
    """P R A Q = L U on the CPU (CSC input, full storage).  scaling: 0 none, 1 sum, 2 max."""

    def __init__(self, n, cp, ri, ax, q=None, scaling=1, pivot_tol=0.1):
        self.n = n
        self._q = None if q is None else _i32(q)
        qptr = None if q is None else self._q.ctypes.data_as(C.c_void_p)
        self._h = LIB.oracle_lu_factor(n, _i32(cp), _i32(ri), _f64(ax), qptr, scaling, pivot_tol)
        if not self._h:
            raise MemoryError("oracle_lu_factor")

    @property
    def status(self):
        return LIB.oracle_lu_status(self._h)

    @property
    def nnz_l(self):
        return LIB.oracle_lu_nnz_l(self._h)

    @property
    def nnz_u(self):
        return LIB.oracle_lu_nnz_u(self._h)

    def solve(self, b, nrefine=2):
        x = np.zeros(self.n)
        LIB.oracle_lu_solve(self._h, _f64(b), x, nrefine)
        return x

    def determinant(self):
        m, e = C.c_double(), C.c_double()
        LIB.oracle_lu_determinant(self._h, C.byref(m), C.byref(e))
        return m.value, e.value

    def rcond(self):
        return LIB.oracle_lu_rcond(self._h)

    def __del__(self):
        if getattr(self, "_h", None):
            LIB.oracle_lu_free(self._h)
            self._h = None


def solve_coo(n, trip_i, trip_j, trip_x, rhs, sym="No", q=None, nrefine=2):
    """Convenience: COO (possibly triangular storage, duplicates) -> full CSC -> LU -> x."""
    ai, aj, ax = list(trip_i), list(trip_j), list(trip_x)
    if sym in ("YesLower", "YesUpper"):
        for i, j, v in zip(list(ai), list(aj), list(ax)):
            if i != j:
                ai.append(j), aj.append(i), ax.append(v)
    cp, ri, vx = coo_to_csc(n, n, ai, aj, ax)
    lu = OracleLU(n, cp, ri, vx, q=q)
    return lu.solve(rhs, nrefine), lu


def fdm_sps(nx, ny, nz=1, periodic=(False, False, False), sym=0, prescribed=None, d=(1.0, 1.0, 1.0), k=(1.0, 1.0, 1.0), alpha=0.0):
    """K-bar / K-check triplets of Fdm2d::get_matrices_sps (and its 7-point analogue) in the reference's order."""
    ntot = nx * ny * nz
    local = np.zeros(ntot, np.int32)
    cap = 7 * ntot
    bi, bj, bv = np.zeros(cap, np.int32), np.zeros(cap, np.int32), np.zeros(cap)
    ci, cj, cv = np.zeros(cap, np.int32), np.zeros(cap, np.int32), np.zeros(cap)
    mask = None if prescribed is None else np.ascontiguousarray(prescribed, dtype=np.uint8)
    nchk = C.c_int64(0)
    nbar = LIB.oracle_fdm_sps(nx, ny, nz, int(periodic[0]), int(periodic[1]), int(periodic[2]), sym,
                              None if mask is None else mask.ctypes.data_as(C.c_void_p), d[0], d[1], d[2], k[0], k[1], k[2], alpha,
                              local, bi, bj, bv, ci, cj, cv, C.byref(nchk))
    nu = ntot if mask is None else int(ntot - np.count_nonzero(mask))
    return {"nu": nu, "np": ntot - nu, "local": local, "bar": (bi[:nbar].copy(), bj[:nbar].copy(), bv[:nbar].copy()),
            "check": (ci[:nchk.value].copy(), cj[:nchk.value].copy(), cv[:nchk.value].copy())}


def fdm_lmm(nx, ny, nz=1, periodic=(False, False, False), sym=0, prescribed=None, d=(1.0, 1.0, 1.0), k=(1.0, 1.0, 1.0), alpha=0.0):
    """Triplets of the augmented matrix M = [K C^T; C 0] of Fdm2d::get_matrices_lmm (and its 7-point analogue) in the reference's order."""
    ntot = nx * ny * nz
    mask = None if prescribed is None else np.ascontiguousarray(prescribed, dtype=np.uint8)
    cap = 7 * ntot + 2 * (0 if mask is None else int(np.count_nonzero(mask)))
    mi, mj, mv = np.zeros(cap, np.int32), np.zeros(cap, np.int32), np.zeros(cap)
    nlag = C.c_int64(0)
    nnz = LIB.oracle_fdm_lmm(nx, ny, nz, int(periodic[0]), int(periodic[1]), int(periodic[2]), sym,
                             None if mask is None else mask.ctypes.data_as(C.c_void_p), d[0], d[1], d[2], k[0], k[1], k[2], alpha, mi, mj, mv,
                             C.byref(nlag))
    return {"neq": ntot, "nlag": nlag.value, "ndim": ntot + nlag.value, "mm": (mi[:nnz].copy(), mj[:nnz].copy(), mv[:nnz].copy())}
