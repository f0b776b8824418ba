"""Finite-difference Laplacian assembled in HBM: host-side handle of the `hipmf_fdm_*` entry points (include/russell_hipmf.h).

Mirrors what a caller of russell_pde's Fdm2d::get_matrices_sps sees (/root/reference/russell_pde/src/fdm_2d.rs:603-649) --
K-bar (unknown x unknown) and K-check (unknown x prescribed) as COO triplets in the reference's order -- with the arrays living
on the device.  There is no CPU fallback: the arrays are produced by HIP kernels.
"""
import ctypes as C

import numpy as np

from . import _capi

SYM_NO, SYM_LOWER, SYM_UPPER = 0, 1, 2


class FdmDevice:
    def __init__(self, nx, ny, nz=1, periodic=(False, False, False), sym=SYM_NO, prescribed=None, lib_path=None):
        self.lib = _capi.load(lib_path)
        self.nx, self.ny, self.nz = nx, ny, nz
        mask = None
        if prescribed is not None:
            mask = np.ascontiguousarray(prescribed, dtype=np.uint8)
            if mask.size != nx * ny * nz:
                raise ValueError("prescribed must have nx*ny*nz entries")
        self.h = self.lib.hipmf_fdm_new(nx, ny, nz, int(periodic[0]), int(periodic[1]), int(periodic[2]), sym,
                                        None if mask is None else mask.ctypes.data_as(C.c_void_p))
        if not self.h:
            raise RuntimeError("hipmf_fdm_new failed (invalid grid, or no HIP device / memory: there is no CPU fallback)")
        nu, npre, nb, nc = C.c_int64(), C.c_int64(), C.c_int64(), C.c_int64()
        assert self.lib.hipmf_fdm_dims(self.h, C.byref(nu), C.byref(npre), C.byref(nb), C.byref(nc)) == 0
        self.nu, self.np, self.nnz_bar, self.nnz_check = nu.value, npre.value, nb.value, nc.value
        self._bufs = []

    def _alloc(self, nbytes):
        p = self.lib.hipmf_device_malloc(max(int(nbytes), 8))
        if not p:
            raise MemoryError("hipmf_device_malloc")
        self._bufs.append(p)
        return p

    def structure_device(self):
        """Device arrays (bar_i, bar_j, check_i, check_j) of int32 local indices."""
        bi, bj = self._alloc(4 * self.nnz_bar), self._alloc(4 * self.nnz_bar)
        ci = cj = None
        if self.nnz_check > 0:
            ci, cj = self._alloc(4 * self.nnz_check), self._alloc(4 * self.nnz_check)
        code = self.lib.hipmf_fdm_structure_device(self.h, bi, bj, ci, cj)
        if code != 0:
            raise RuntimeError("hipmf_fdm_structure_device failed with status %d" % code)
        return bi, bj, ci, cj

    def values_device(self, d=(1.0, 1.0, 1.0), k=(1.0, 1.0, 1.0), alpha=0.0, out=None):
        """Device arrays (bar_values, check_values); `out` re-uses a previous pair."""
        bv, cv = out if out is not None else (self._alloc(8 * self.nnz_bar), self._alloc(8 * self.nnz_check) if self.nnz_check > 0 else None)
        code = self.lib.hipmf_fdm_values_device(self.h, d[0], d[1], d[2], k[0], k[1], k[2], alpha, bv, cv)
        if code != 0:
            raise RuntimeError("hipmf_fdm_values_device failed with status %d" % code)
        return bv, cv

    # ---- Lagrange-multiplier form: M = [K C^T; C 0] of Fdm2d::get_matrices_lmm (fdm_2d.rs:672-748) ----
    def lmm_dims(self):
        """(neq, nlag, nnz) of the augmented matrix; its order is neq + nlag."""
        neq, nlag, nnz = C.c_int64(), C.c_int64(), C.c_int64()
        code = self.lib.hipmf_fdm_lmm_dims(self.h, C.byref(neq), C.byref(nlag), C.byref(nnz))
        if code != 0:
            raise RuntimeError("hipmf_fdm_lmm_dims failed with status %d" % code)
        return neq.value, nlag.value, nnz.value

    def lmm_structure_device(self):
        """Device arrays (i, j) of the triplets of M (global node numbers; multiplier ip is row / column neq + ip)."""
        nnz = self.lmm_dims()[2]
        di, dj = self._alloc(4 * nnz), self._alloc(4 * nnz)
        code = self.lib.hipmf_fdm_lmm_structure_device(self.h, di, dj)
        if code != 0:
            raise RuntimeError("hipmf_fdm_lmm_structure_device failed with status %d" % code)
        return di, dj

    def lmm_values_device(self, d=(1.0, 1.0, 1.0), k=(1.0, 1.0, 1.0), alpha=0.0, out=None):
        dv = out if out is not None else self._alloc(8 * self.lmm_dims()[2])
        code = self.lib.hipmf_fdm_lmm_values_device(self.h, d[0], d[1], d[2], k[0], k[1], k[2], alpha, dv)
        if code != 0:
            raise RuntimeError("hipmf_fdm_lmm_values_device failed with status %d" % code)
        return dv

    def to_host(self, dptr, count, dtype):
        a = np.zeros(max(count, 1), dtype)
        if count > 0:
            assert self.lib.hipmf_memcpy_d2h(a.ctypes.data_as(C.c_void_p), dptr, a.itemsize * count) == 0
        return a[:count]

    def close(self):
        for p in self._bufs:
            self.lib.hipmf_device_free(p)
        self._bufs = []
        if getattr(self, "h", None):
            self.lib.hipmf_fdm_drop(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
