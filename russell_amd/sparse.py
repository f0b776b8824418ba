"""Python face of the host layer (include/russell_host.h): the russell_sparse names a user of the reference knows.

    from russell_amd.sparse import CooMatrix, Genie, LinSolver, LinSolParams, Sym

    coo = CooMatrix(5, 5, 13, Sym.No); coo.put(0, 0, 2.0); ...
    solver = LinSolver(Genie.Hipmf)
    solver.actual.factorize(coo, None)
    x = solver.actual.solve(rhs)

Everything below is a thin ctypes veneer: validation, COO->CSC/CSR conversion, error strings and the solver calls
live in C++ (russell_amd/csrc/host/) and in the HIP library.  Errors surface as `StrError` carrying the exact
string the Rust layer would return (e.g. "the matrix must be square").
"""
import ctypes as C
import enum
import json
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_HOSTLIB = os.path.join(_HERE, "lib", "librussell_host.so")
_lib = None


class StrError(Exception):
    """The analogue of Rust's `StrError = &'static str`."""


class Sym(enum.IntEnum):
    No = 0
    YesFull = 1
    YesLower = 2
    YesUpper = 3


class Genie(enum.IntEnum):
    Hipmf = 0
    Umfpack = 1
    Mumps = 2
    Cudss = 3

    def to_string(self):
        return _L().rh_enum_name(2, int(self)).decode()

    def get_sym(self, symmetric):
        return Sym(_L().rh_genie_get_sym(int(self), int(bool(symmetric))))

    @staticmethod
    def from_name(name):
        return {"umfpack": Genie.Umfpack, "mumps": Genie.Mumps, "cudss": Genie.Cudss}.get(name.lower(), Genie.Hipmf)


class Ordering(enum.IntEnum):
    Amd = 0
    Amf = 1
    Auto = 2
    Best = 3
    BtfColamd = 4
    Cholmod = 5
    Colamd = 6
    Metis = 7
    No = 8
    Pord = 9
    Qamd = 10
    Scotch = 11


class Scaling(enum.IntEnum):
    Auto = 0
    Column = 1
    Diagonal = 2
    Max = 3
    No = 4
    RowCol = 5
    RowColIter = 6
    RowColRig = 7
    Sum = 8


class MMsym(enum.IntEnum):
    LeaveAsLower = 0
    SwapToUpper = 1
    MakeItFull = 2


class _RhParams(C.Structure):
    _fields_ = [("ordering", C.c_int32), ("scaling", C.c_int32), ("has_pivot_epsilon", C.c_int32), ("pivot_epsilon", C.c_double),
                ("has_refinement_nstep", C.c_int32), ("refinement_nstep", C.c_int32), ("positive_definite", C.c_int32),
                ("compute_determinant", C.c_int32), ("verbose", C.c_int32), ("matching", C.c_int32), ("pivoting", C.c_int32),
                ("has_hybrid_memory_factor", C.c_int32), ("hybrid_memory_factor", C.c_double), ("compute_error_estimates", C.c_int32),
                ("compute_condition_numbers", C.c_int32)]


class LinSolParams:
    """lin_sol_params.rs:5-107 (the fields this backend honours)."""

    def __init__(self):
        self.ordering = Ordering.Auto
        self.scaling = Scaling.Auto
        self.pivot_epsilon = None
        self.refinement_nstep = None
        self.positive_definite = False
        self.compute_determinant = False
        self.verbose = False
        self.matching = 1                 # enums.rs Matching: 0 None, 1 Auto, 2.. the named variants (all select the maximum-product matching)
        self.pivoting = 0                 # enums.rs Pivoting: 0 Auto .. 5 LocalBlock: a request (round 6); effective: LocalBlock
        self.hybrid_memory_factor = None  # lin_sol_params.rs:39 (recorded; no out-of-core path)
        self.compute_error_estimates = False
        self.compute_condition_numbers = False

    def _c(self):
        return _RhParams(int(self.ordering), int(self.scaling), int(self.pivot_epsilon is not None), float(self.pivot_epsilon or 0.0),
                         int(self.refinement_nstep is not None), int(self.refinement_nstep or 0), int(self.positive_definite),
                         int(self.compute_determinant), int(self.verbose), int(self.matching), int(self.pivoting),
                         int(self.hybrid_memory_factor is not None), float(self.hybrid_memory_factor or 0.0),
                         int(self.compute_error_estimates), int(self.compute_condition_numbers))


def _L():
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(_HOSTLIB):
        raise RuntimeError("russell_amd: %s is missing -- run `python -c 'import __graft_entry__ as g; g.build()'`" % _HOSTLIB)
    lib = C.CDLL(_HOSTLIB)
    vp, cp, i64, i32, f64 = C.c_void_p, C.c_char_p, C.c_int64, C.c_int32, C.c_double
    pp = C.POINTER
    sig = {
        "rh_set_hipmf_library": (None, [cp]),
        "rh_coo_new": (vp, [i64, i64, i64, i32, pp(cp)]),
        "rh_coo_free": (None, [vp]),
        "rh_coo_put": (cp, [vp, i64, i64, f64]),
        "rh_coo_reset": (None, [vp]),
        "rh_coo_info": (None, [vp, pp(i64), pp(i64), pp(i64), pp(i64), pp(i32)]),
        "rh_coo_arrays": (None, [vp, pp(pp(i32)), pp(pp(i32)), pp(pp(f64))]),
        "rh_coo_mat_vec_mul": (cp, [vp, vp, i64, f64, vp, i64]),
        "rh_coo_from": (vp, [i64, i64, i64, vp, vp, vp, i32, pp(cp)]),
        "rh_coo_mat_vec_mul_update": (cp, [vp, vp, i64, f64, vp, i64]),
        "rh_coo_mat_t_vec_mul": (cp, [vp, vp, i64, f64, vp, i64]),
        "rh_coo_assign": (cp, [vp, f64, vp]),
        "rh_coo_add": (cp, [vp, f64, vp]),
        "rh_coo_put_lagrange_block": (cp, [vp, vp]),
        "rh_coo_to_dense": (cp, [vp, vp, i64]),
        "rh_coo_actual_nnz": (i64, [vp]),
        "rh_csc_new": (vp, [i64, i64, vp, i64, vp, vp, i64, i32, pp(cp)]),
        "rh_csr_new": (vp, [i64, i64, vp, i64, vp, vp, i64, i32, pp(cp)]),
        "rh_csc_to_dense": (cp, [vp, vp, i64]),
        "rh_csr_to_dense": (cp, [vp, vp, i64]),
        "rh_csc_from_csr": (vp, [vp, pp(cp)]),
        "rh_csr_from_csc": (vp, [vp, pp(cp)]),
        "rh_csc_from_coo": (vp, [vp, pp(cp)]),
        "rh_csc_update_from_coo": (cp, [vp, vp]),
        "rh_csc_arrays": (None, [vp, pp(pp(i32)), pp(pp(i32)), pp(pp(f64)), pp(i64), pp(i64)]),
        "rh_csc_mat_vec_mul": (cp, [vp, vp, i64, f64, vp, i64]),
        "rh_csc_free": (None, [vp]),
        "rh_csr_from_coo": (vp, [vp, pp(cp)]),
        "rh_csr_update_from_coo": (cp, [vp, vp]),
        "rh_csr_arrays": (None, [vp, pp(pp(i32)), pp(pp(i32)), pp(pp(f64)), pp(i64), pp(i64)]),
        "rh_csr_mat_vec_mul": (cp, [vp, vp, i64, f64, vp, i64]),
        "rh_csr_free": (None, [vp]),
        "rh_verify": (cp, [vp, vp, i64, vp, i64, vp]),
        "rh_read_matrix_market": (vp, [cp, i32, pp(cp)]),
        "rh_linsolver_new": (vp, [i32, pp(cp)]),
        "rh_linsolver_free": (None, [vp]),
        "rh_linsolver_factorize": (cp, [vp, vp, pp(_RhParams)]),
        "rh_linsolver_solve": (cp, [vp, vp, i64, vp, i64, i32]),
        "rh_linsolver_solve_many": (cp, [vp, vp, vp, i64, i64]),
        "rh_linsolver_times": (None, [vp, pp(C.c_uint64)]),
        "rh_linsolver_outputs": (None, [vp, pp(f64), pp(f64), pp(f64), pp(i32), pp(i32), pp(i32)]),
        "rh_linsolver_stats_json": (cp, [vp, vp, cp, vp, vp]),
        "rh_coo_put_many": (cp, [vp, i64, vp, vp, vp]),
        "rh_ccoo_put_many": (cp, [vp, i64, vp, vp, vp]),
        "rh_ccoo_new": (vp, [i64, i64, i64, i32, pp(cp)]),
        "rh_ccoo_free": (None, [vp]),
        "rh_ccoo_put": (cp, [vp, i64, i64, f64, f64]),
        "rh_ccoo_reset": (None, [vp]),
        "rh_ccoo_mat_vec_mul": (cp, [vp, vp, i64, f64, f64, vp, i64]),
        "rh_clinsolver_new": (vp, [pp(cp)]),
        "rh_clinsolver_free": (None, [vp]),
        "rh_clinsolver_factorize": (cp, [vp, vp, pp(_RhParams)]),
        "rh_clinsolver_solve": (cp, [vp, vp, i64, vp, i64, i32]),
        "rh_clinsolver_outputs": (None, [vp, pp(f64), pp(f64), pp(f64), pp(f64), pp(i32)]),
        "rh_error_string": (cp, [i32]),
        "rh_format_nanoseconds": (None, [C.c_uint64, C.c_char_p, i32]),
        "rh_is_memory_error": (i32, [cp]),
        "rh_ccoo_info": (None, [vp, pp(i64), pp(i64), pp(i64), pp(i64), pp(i32)]),
        "rh_ccoo_arrays": (None, [vp, pp(pp(i32)), pp(pp(i32)), pp(pp(f64))]),
        "rh_read_matrix_market_any": (cp, [cp, i32, pp(vp), pp(vp)]),
        "rh_enum_name": (cp, [i32, i32]),
        "rh_genie_get_sym": (i32, [i32, i32]),
    }
    for name, (res, args) in sig.items():
        fn = getattr(lib, name)
        fn.restype, fn.argtypes = res, args
    lib.rh_set_hipmf_library(os.path.join(_HERE, "lib", "librussell_hipmf.so").encode())
    _lib = lib
    return lib


def _check(err):
    if err:
        raise StrError(err.decode())


def _vec(a):
    return np.ascontiguousarray(a, dtype=np.float64)


def _ptr(a):
    return a.ctypes.data_as(C.c_void_p)


class CooMatrix:
    """coo_matrix.rs:21-73: triplets with duplicates allowed, nnz <= max_nnz, triangular-storage guard in put()."""

    def __init__(self, nrow, ncol, max_nnz, symmetric=Sym.No, _handle=None):
        if _handle is not None:
            self._h = _handle
            return
        err = C.c_char_p()
        self._h = _L().rh_coo_new(int(nrow), int(ncol), int(max_nnz), int(symmetric), C.byref(err))
        _check(err.value)

    def __del__(self):
        if getattr(self, "_h", None) and _lib is not None:
            _lib.rh_coo_free(self._h)
            self._h = None

    @classmethod
    def from_arrays(cls, nrow, ncol, row_indices, col_indices, values, symmetric=Sym.No):
        """coo_matrix.rs:246-291 (`CooMatrix::from`): the triplet arrays as given (nnz = max_nnz = their length)."""
        ii, jj = np.ascontiguousarray(row_indices, dtype=np.int32), np.ascontiguousarray(col_indices, dtype=np.int32)
        aa = np.ascontiguousarray(values, dtype=np.float64)
        if jj.size != ii.size:
            raise StrError("col_indices.len() must be = nnz")
        if aa.size != ii.size:
            raise StrError("values.len() must be = nnz")
        err = C.c_char_p()
        h = _L().rh_coo_from(int(nrow), int(ncol), ii.size, _ptr(ii), _ptr(jj), _ptr(aa), int(symmetric), C.byref(err))
        _check(err.value)
        return cls(0, 0, 0, _handle=h)

    def put(self, i, j, aij):
        _check(_L().rh_coo_put(self._h, int(i), int(j), float(aij)))

    def put_many(self, i, j, aij):
        ii, jj = np.ascontiguousarray(i, dtype=np.int32), np.ascontiguousarray(j, dtype=np.int32)
        aa = np.ascontiguousarray(aij, dtype=np.float64)
        _check(_L().rh_coo_put_many(self._h, ii.size, _ptr(ii), _ptr(jj), _ptr(aa)))

    def reset(self):
        _L().rh_coo_reset(self._h)

    def get_info(self):
        a, b, c, d, s = C.c_int64(), C.c_int64(), C.c_int64(), C.c_int64(), C.c_int32()
        _L().rh_coo_info(self._h, C.byref(a), C.byref(b), C.byref(c), C.byref(d), C.byref(s))
        return a.value, b.value, c.value, Sym(s.value)

    @property
    def nrow(self):
        return self.get_info()[0]

    @property
    def ncol(self):
        return self.get_info()[1]

    @property
    def nnz(self):
        return self.get_info()[2]

    @property
    def symmetric(self):
        return self.get_info()[3]

    def triplets(self):
        ai, aj, ax = C.POINTER(C.c_int32)(), C.POINTER(C.c_int32)(), C.POINTER(C.c_double)()
        _L().rh_coo_arrays(self._h, C.byref(ai), C.byref(aj), C.byref(ax))
        n = self.nnz
        return (np.ctypeslib.as_array(ai, (n,)).copy(), np.ctypeslib.as_array(aj, (n,)).copy(), np.ctypeslib.as_array(ax, (n,)).copy())

    def mat_vec_mul(self, u, alpha=1.0, nv=None):
        u = _vec(u)
        v = np.zeros(self.nrow if nv is None else nv)
        _check(_L().rh_coo_mat_vec_mul(self._h, _ptr(v), v.size, float(alpha), _ptr(u), u.size))
        return v

    def mat_vec_mul_update(self, v, u, alpha=1.0):
        """v += alpha * A * u in place (coo_matrix.rs:629)."""
        u = _vec(u)
        if not (isinstance(v, np.ndarray) and v.dtype == np.float64 and v.flags.c_contiguous):
            raise TypeError("v must be a contiguous float64 array (updated in place)")
        _check(_L().rh_coo_mat_vec_mul_update(self._h, _ptr(v), v.size, float(alpha), _ptr(u), u.size))
        return v

    def mat_t_vec_mul(self, u, alpha=1.0, nv=None):
        """v = alpha * A^T * u (coo_matrix.rs:708)."""
        u = _vec(u)
        v = np.zeros(self.ncol if nv is None else nv)
        _check(_L().rh_coo_mat_t_vec_mul(self._h, _ptr(v), v.size, float(alpha), _ptr(u), u.size))
        return v

    def assign(self, alpha, other):
        """this = alpha * other, triplet by triplet (coo_matrix.rs:738)."""
        _check(_L().rh_coo_assign(self._h, float(alpha), other._h))

    def add(self, alpha, other):
        """this += alpha * other: the triplets of other are appended (coo_matrix.rs:779)."""
        _check(_L().rh_coo_add(self._h, float(alpha), other._h))

    def put_lagrange_block(self, bb):
        """Appends B (and B^T for full storage) below / right of the leading ncol(B) block (coo_matrix.rs:823)."""
        _check(_L().rh_coo_put_lagrange_block(self._h, bb._h))

    def to_dense(self):
        a = np.zeros((self.nrow, self.ncol))
        _check(_L().rh_coo_to_dense(self._h, _ptr(a), a.size))
        return a

    as_dense = to_dense

    def get_actual_nnz(self):
        return int(_L().rh_coo_actual_nnz(self._h))


class _Compressed:
    _kind = ""

    def __init__(self, handle):
        self._h = handle

    @classmethod
    def from_coo(cls, coo):
        err = C.c_char_p()
        h = getattr(_L(), "rh_%s_from_coo" % cls._kind)(coo._h, C.byref(err))
        _check(err.value)
        return cls(h)

    @classmethod
    def _from_other(cls, fn, other):
        err = C.c_char_p()
        h = fn(other._h, C.byref(err))
        _check(err.value)
        return cls(h)

    @classmethod
    def new(cls, nrow, ncol, pointers, indices, values, symmetric=Sym.No):
        """csc_matrix.rs:197-262 / csr_matrix.rs:193-257: validated constructor from ready arrays."""
        pp_, ii = np.ascontiguousarray(pointers, dtype=np.int32), np.ascontiguousarray(indices, dtype=np.int32)
        vv = np.ascontiguousarray(values, dtype=np.float64)
        nv = min(ii.size, vv.size)
        if pp_.size > 0 and ((cls._kind == "csc" and pp_.size == ncol + 1) or (cls._kind == "csr" and pp_.size == nrow + 1)):
            if ii.size < pp_[-1]:
                raise StrError("%s_indices.len() must be ≥ nnz" % ("row" if cls._kind == "csc" else "col"))
            if vv.size < pp_[-1]:
                raise StrError("values.len() must be ≥ nnz")
        err = C.c_char_p()
        h = getattr(_L(), "rh_%s_new" % cls._kind)(int(nrow), int(ncol), _ptr(pp_), pp_.size, _ptr(ii), _ptr(vv), nv, int(symmetric), C.byref(err))
        _check(err.value)
        obj = cls(h)
        obj._shape = (int(nrow), int(ncol))
        return obj

    def to_dense(self, nrow=None, ncol=None):
        nrow, ncol = (nrow, ncol) if nrow is not None else getattr(self, "_shape")
        a = np.zeros((nrow, ncol))
        _check(getattr(_L(), "rh_%s_to_dense" % self._kind)(self._h, _ptr(a), a.size))
        return a

    def update_from_coo(self, coo):
        _check(getattr(_L(), "rh_%s_update_from_coo" % self._kind)(self._h, coo._h))

    def arrays(self):
        p, i, x = C.POINTER(C.c_int32)(), C.POINTER(C.c_int32)(), C.POINTER(C.c_double)()
        nd, nnz = C.c_int64(), C.c_int64()
        getattr(_L(), "rh_%s_arrays" % self._kind)(self._h, C.byref(p), C.byref(i), C.byref(x), C.byref(nd), C.byref(nnz))
        return (np.ctypeslib.as_array(p, (nd.value + 1,)).copy(), np.ctypeslib.as_array(i, (max(nnz.value, 1),))[:nnz.value].copy(),
                np.ctypeslib.as_array(x, (max(nnz.value, 1),))[:nnz.value].copy())

    def mat_vec_mul(self, u, nrow, alpha=1.0):
        u = _vec(u)
        v = np.zeros(nrow)
        _check(getattr(_L(), "rh_%s_mat_vec_mul" % self._kind)(self._h, _ptr(v), v.size, float(alpha), _ptr(u), u.size))
        return v

    def __del__(self):
        if getattr(self, "_h", None) and _lib is not None:
            getattr(_lib, "rh_%s_free" % self._kind)(self._h)
            self._h = None


class CscMatrix(_Compressed):
    """csc_matrix.rs:337-505 (col_pointers, row_indices, values)."""
    _kind = "csc"

    @classmethod
    def from_csr(cls, csr):
        """csc_matrix.rs:508-584."""
        return cls._from_other(_L().rh_csc_from_csr, csr)


class CsrMatrix(_Compressed):
    """csr_matrix.rs:332-480 (row_pointers, col_indices, values)."""
    _kind = "csr"

    @classmethod
    def from_csc(cls, csc):
        """csr_matrix.rs:483-558."""
        return cls._from_other(_L().rh_csr_from_csc, csc)


class VerifyLinSys:
    """verify_lin_sys.rs:60-96."""

    def __init__(self, mat, x, rhs):
        out = np.zeros(4)
        x, rhs = _vec(x), _vec(rhs)
        _check(_L().rh_verify(mat._h, _ptr(x), x.size, _ptr(rhs), rhs.size, _ptr(out)))
        self.max_abs_a, self.max_abs_ax, self.max_abs_diff, self.relative_error = (float(v) for v in out)


def read_matrix_market(full_path, symmetric_handling=MMsym.LeaveAsLower):
    """read_matrix_market.rs:346-475 (real matrices)."""
    err = C.c_char_p()
    h = _L().rh_read_matrix_market(os.fspath(full_path).encode(), int(symmetric_handling), C.byref(err))
    _check(err.value)
    return CooMatrix(0, 0, 0, _handle=h)


def read_matrix_market_any(full_path, symmetric_handling=MMsym.LeaveAsLower):
    """read_matrix_market.rs:346-475: returns (CooMatrix | None, ComplexCooMatrix | None), exactly one of them set."""
    coo, ccoo = C.c_void_p(), C.c_void_p()
    _check(_L().rh_read_matrix_market_any(os.fspath(full_path).encode(), int(symmetric_handling), C.byref(coo), C.byref(ccoo)))
    if ccoo.value:
        return None, ComplexCooMatrix(0, 0, 0, _handle=ccoo.value)
    return CooMatrix(0, 0, 0, _handle=coo.value), None


def format_nanoseconds(nanoseconds):
    """russell_lab base/formatters.rs:60-95."""
    buf = C.create_string_buffer(96)
    _L().rh_format_nanoseconds(int(nanoseconds), buf, 96)
    return buf.value.decode()


def is_memory_error(message):
    """stats_lin_sol.rs:334-340."""
    return bool(_L().rh_is_memory_error(message.encode()))


class _Actual:
    """What `solver.actual` exposes: the LinSolTrait methods (lin_solver.rs:12-64)."""

    def __init__(self, handle):
        self._h = handle  # owned here: `LinSolver(g).actual.factorize(...)` must keep the C++ object alive
        self._ndim = None

    def __del__(self):
        if getattr(self, "_h", None) and _lib is not None:
            _lib.rh_linsolver_free(self._h)
            self._h = None

    def factorize(self, mat, params=None):
        p = C.byref(params._c()) if params is not None else None
        _check(_L().rh_linsolver_factorize(self._h, mat._h, p))
        self._ndim = mat.nrow

    def solve(self, rhs, x=None, verbose=False):
        rhs = _vec(rhs)
        # (the reference's signature is solve(&mut x, &rhs): a caller's float64 vector is written in place, like the Rust caller's)
        if isinstance(x, np.ndarray) and x.dtype == np.float64 and x.ndim == 1 and x.flags.c_contiguous and x.flags.writeable:
            out = x
        else:
            out = np.zeros(self._ndim if (x is None and self._ndim is not None) else (len(x) if x is not None else rhs.size))
        _check(_L().rh_linsolver_solve(self._h, _ptr(out), out.size, _ptr(rhs), rhs.size, int(verbose)))
        if x is not None and out is not x:
            x[:] = out
        return out

    def solve_many(self, rhs_rows):
        b = _vec(rhs_rows)
        nrhs, n = b.shape
        x = np.zeros_like(b)
        _check(_L().rh_linsolver_solve_many(self._h, _ptr(x), _ptr(b), n, nrhs))
        return x

    def get_ns(self):
        ns = (C.c_uint64 * 3)()
        _L().rh_linsolver_times(self._h, ns)
        return int(ns[0]), int(ns[1]), int(ns[2])

    def get_ns_init(self):
        return self.get_ns()[0]

    def get_ns_fact(self):
        return self.get_ns()[1]

    def get_ns_solve(self):
        return self.get_ns()[2]

    def outputs(self):
        dc, de, rc = C.c_double(), C.c_double(), C.c_double()
        eo, es, npv = C.c_int32(), C.c_int32(), C.c_int32()
        _L().rh_linsolver_outputs(self._h, C.byref(dc), C.byref(de), C.byref(rc), C.byref(eo), C.byref(es), C.byref(npv))
        return dict(determinant_coefficient=dc.value, determinant_exponent=de.value, rcond_estimate=rc.value, effective_ordering=eo.value,
                    effective_scaling=es.value, perturbed_pivots=npv.value)

    def stats(self, mat=None, name="", x=None, rhs=None):
        xx = _vec(x) if x is not None else None
        rr = _vec(rhs) if rhs is not None else None
        s = _L().rh_linsolver_stats_json(self._h, mat._h if mat is not None else None, name.encode(), _ptr(xx) if xx is not None else None,
                                         _ptr(rr) if rr is not None else None)
        return json.loads(s.decode())


class LinSolver:
    """lin_solver.rs:105-142: `LinSolver::new(genie)` boxes a backend behind `actual`."""

    def __init__(self, genie=Genie.Hipmf):
        err = C.c_char_p()
        h = _L().rh_linsolver_new(int(genie), C.byref(err))
        _check(err.value)
        self.actual = _Actual(h)

    @staticmethod
    def compute(genie, mat, rhs, params=None):
        """lin_solver.rs:212-224: allocate, factorize and solve in one call; returns (solver, x)."""
        s = LinSolver(genie)
        s.actual.factorize(mat, params)
        return s, s.actual.solve(rhs)


class ComplexCooMatrix:
    """complex_coo_matrix.rs: COO triplets with Complex64 values (duplicates allowed, summed at conversion)."""

    def __init__(self, nrow, ncol, max_nnz, symmetric=Sym.No, _handle=None):
        if _handle is not None:
            self._h = _handle
            self.nrow, self.ncol = self.get_info()[:2]
            return
        err = C.c_char_p()
        self._h = _L().rh_ccoo_new(int(nrow), int(ncol), int(max_nnz), int(symmetric), C.byref(err))
        _check(err.value)
        self.nrow, self.ncol = int(nrow), int(ncol)

    def get_info(self):
        a, b, c, d, s = C.c_int64(), C.c_int64(), C.c_int64(), C.c_int64(), C.c_int32()
        _L().rh_ccoo_info(self._h, C.byref(a), C.byref(b), C.byref(c), C.byref(d), C.byref(s))
        return a.value, b.value, c.value, Sym(s.value)

    def triplets(self):
        ai, aj, ax = C.POINTER(C.c_int32)(), C.POINTER(C.c_int32)(), C.POINTER(C.c_double)()
        _L().rh_ccoo_arrays(self._h, C.byref(ai), C.byref(aj), C.byref(ax))
        n = self.get_info()[2]
        vals = np.ctypeslib.as_array(ax, (2 * n,)).copy().view(np.complex128)
        return np.ctypeslib.as_array(ai, (n,)).copy(), np.ctypeslib.as_array(aj, (n,)).copy(), vals

    def __del__(self):
        if getattr(self, "_h", None) and _lib is not None:
            _lib.rh_ccoo_free(self._h)
            self._h = None

    def put(self, i, j, aij):
        aij = complex(aij)
        _check(_L().rh_ccoo_put(self._h, int(i), int(j), aij.real, aij.imag))

    def put_many(self, i, j, aij):
        ii, jj = np.ascontiguousarray(i, dtype=np.int32), np.ascontiguousarray(j, dtype=np.int32)
        aa = np.ascontiguousarray(np.asarray(aij, dtype=np.complex128)).view(np.float64)
        _check(_L().rh_ccoo_put_many(self._h, ii.size, _ptr(ii), _ptr(jj), _ptr(aa)))

    def reset(self):
        _L().rh_ccoo_reset(self._h)

    def mat_vec_mul(self, u, alpha=1.0):
        """v = alpha * A * u (complex vectors)."""
        uu = np.ascontiguousarray(np.asarray(u, dtype=np.complex128)).view(np.float64)
        v = np.zeros(2 * self.nrow)
        alpha = complex(alpha)
        _check(_L().rh_ccoo_mat_vec_mul(self._h, _ptr(v), v.size, alpha.real, alpha.imag, _ptr(uu), uu.size))
        return v.view(np.complex128)


class _ComplexActual:
    """ComplexLinSolTrait (complex_lin_solver.rs:12-64)."""

    def __init__(self, handle):
        self._h = handle

    def __del__(self):
        if getattr(self, "_h", None) and _lib is not None:
            _lib.rh_clinsolver_free(self._h)
            self._h = None

    def factorize(self, mat, params=None):
        p = C.byref(params._c()) if params is not None else None
        _check(_L().rh_clinsolver_factorize(self._h, mat._h, p))

    def solve(self, rhs, x=None, verbose=False):
        r = np.ascontiguousarray(np.asarray(rhs, dtype=np.complex128)).view(np.float64)
        nx = r.size if x is None else 2 * len(x)
        out = np.zeros(nx)
        _check(_L().rh_clinsolver_solve(self._h, _ptr(out), out.size, _ptr(r), r.size, int(verbose)))
        z = out.view(np.complex128)
        if x is not None:
            x[:] = z
        return z

    def outputs(self):
        """determinant = determinant_coefficient x 10^determinant_exponent (complex_solver_umfpack.rs:411-414), rcond, perturbed pivots"""
        dr, di, de, rc, npv = C.c_double(), C.c_double(), C.c_double(), C.c_double(), C.c_int32()
        _L().rh_clinsolver_outputs(self._h, C.byref(dr), C.byref(di), C.byref(de), C.byref(rc), C.byref(npv))
        return dict(determinant_coefficient=complex(dr.value, di.value), determinant_exponent=de.value, rcond_estimate=rc.value, perturbed_pivots=npv.value)


class ComplexLinSolver:
    """complex_lin_solver.rs:105-142 for Genie::Hipmf: the complex system is solved through its real-equivalent form."""

    def __init__(self, genie=Genie.Hipmf):
        if genie != Genie.Hipmf:
            raise StrError("only Genie::Hipmf is available in this build")
        err = C.c_char_p()
        h = _L().rh_clinsolver_new(C.byref(err))
        _check(err.value)
        self.actual = _ComplexActual(h)


def handle_hipmf_error_code(code):
    return _L().rh_error_string(int(code)).decode()
