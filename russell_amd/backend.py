"""Thin object wrapper over the C-ABI (include/russell_hipmf.h): one handle = one solver on one GPU.

This is plumbing for tests / bench / the Python mirror of the Rust host layer (russell_amd.sparse).
It adds no numerics of its own; every number comes out of the HIP library.
"""
import ctypes as C

import numpy as np

from . import _capi

ISTAT_NAMES = ["ndim", "nnz_a", "nsuper", "nlevels", "nnz_l", "nnz_u", "max_front", "max_pivots", "n_perturbed", "n_zero_pivot",
               "refinement_steps", "factor_launches", "solve_launches", "pool_bytes", "matched", "fused_fallbacks"]
DSTAT_NAMES = ["flops", "flops_gemm", "ordering_s", "symbolic_s", "assemble_ms", "factor_ms", "fwd_ms", "bwd_ms", "solve_total_ms",
               "residual_inf", "acc_assemble_ms", "acc_factor_ms", "acc_factor_count", "acc_fwd_ms", "acc_bwd_ms", "acc_tri_count"]


class HipmfError(RuntimeError):
    def __init__(self, code, where, detail=""):
        super().__init__("%s failed with status %d %s" % (where, code, detail))
        self.code = code


class Hipmf:
    def __init__(self, lib_path=None):
        self.lib = _capi.load(lib_path)
        self.h = self.lib.solver_hipmf_new()
        if not self.h:
            raise RuntimeError("solver_hipmf_new returned NULL: no HIP device visible (there is no CPU fallback)")
        self.n = 0
        self.nnz = 0

    def close(self):
        if getattr(self, "h", None):
            self.lib.solver_hipmf_drop(self.h)
            self.h = None

    def __del__(self):
        self.close()

    def _err(self, code, where):
        return HipmfError(code, where, (self.lib.solver_hipmf_last_error(self.h) or b"").decode())

    def initialize(self, n, row_pointers, col_indices, ordering=0, scaling=1, pivot_epsilon=-1.0, refinement_nstep=-1,
                   verbose=False, general_symmetric=False, positive_definite=False, values=None):
        """values (optional, as in the reference's shims, which hand the numbers to the analysis phase too): lets the
        analysis apply the maximum-product matching + scaling when the diagonal is weak."""
        rp = np.ascontiguousarray(row_pointers, dtype=np.int32)
        ci = np.ascontiguousarray(col_indices, dtype=np.int32)
        self.n, self.nnz = int(n), int(rp[n])
        vptr = None
        if values is not None:
            self._init_values = np.ascontiguousarray(values, dtype=np.float64)
            assert self._init_values.size >= self.nnz
            vptr = self._init_values.ctypes.data
        return self.lib.solver_hipmf_initialize(self.h, ordering, scaling, pivot_epsilon, refinement_nstep, int(verbose),
                                                int(general_symmetric), int(positive_definite), n, rp, ci, vptr)

    def factorize(self, values, compute_determinant=False, verbose=False):
        v = np.ascontiguousarray(values, dtype=np.float64)
        assert v.size >= self.nnz
        eo, es, npv = C.c_int32(), C.c_int32(), C.c_int32()
        rc, dc, de = C.c_double(), C.c_double(), C.c_double()
        code = self.lib.solver_hipmf_factorize(self.h, C.byref(eo), C.byref(es), C.byref(npv), C.byref(rc), C.byref(dc), C.byref(de),
                                               int(compute_determinant), int(verbose), v)
        self.effective_ordering, self.effective_scaling, self.num_perturbed = eo.value, es.value, npv.value
        self.rcond, self.det_coefficient, self.det_exponent = rc.value, dc.value, de.value
        return code

    def set_value_map(self, seg_ptr, seg_idx):
        """CSR entry j <- sum of input[seg_idx[seg_ptr[j]:seg_ptr[j+1]]] (e.g. COO triplets with duplicates)."""
        sp = np.ascontiguousarray(seg_ptr, dtype=np.int32)
        si = np.ascontiguousarray(seg_idx, dtype=np.int32)
        self.nnz_in = int(si.size)
        return self.lib.solver_hipmf_set_value_map(self.h, self.nnz_in, sp, si)

    def factorize_mapped(self, input_values, compute_determinant=False, verbose=False):
        v = np.ascontiguousarray(input_values, dtype=np.float64)
        assert v.size >= self.nnz_in
        eo, es, npv = C.c_int32(), C.c_int32(), C.c_int32()
        rc, dc, de = C.c_double(), C.c_double(), C.c_double()
        code = self.lib.solver_hipmf_factorize_mapped(self.h, C.byref(eo), C.byref(es), C.byref(npv), C.byref(rc), C.byref(dc),
                                                      C.byref(de), int(compute_determinant), int(verbose), v)
        self.effective_ordering, self.effective_scaling, self.num_perturbed = eo.value, es.value, npv.value
        self.rcond, self.det_coefficient, self.det_exponent = rc.value, dc.value, de.value
        return code

    def solve(self, rhs, verbose=False):
        b = np.ascontiguousarray(rhs, dtype=np.float64)
        x = np.zeros(self.n)
        code = self.lib.solver_hipmf_solve(self.h, x, b, int(verbose))
        if code != 0:
            raise self._err(code, "solver_hipmf_solve")
        return x

    def solve_many(self, rhs_colmajor):
        """rhs_colmajor: array of shape (nrhs, n) whose rows are the right-hand sides (= column-major n x nrhs)."""
        b = np.ascontiguousarray(rhs_colmajor, dtype=np.float64)
        if b.ndim != 2 or b.shape[1] != self.n:
            raise ValueError("solve_many expects an array of shape (nrhs, n) with n = %d, got %r" % (self.n, b.shape))
        nrhs = b.shape[0]
        x = np.zeros_like(b)
        code = self.lib.solver_hipmf_solve_many(self.h, x, b, nrhs, self.n, 0)
        if code != 0:
            raise self._err(code, "solver_hipmf_solve_many")
        return x

    def mat_vec_mul(self, u, alpha=1.0):
        v = np.zeros(self.n)
        code = self.lib.solver_hipmf_mat_vec_mul(self.h, v, alpha, np.ascontiguousarray(u, dtype=np.float64))
        if code != 0:
            raise self._err(code, "solver_hipmf_mat_vec_mul")
        return v

    def permutation(self):
        p = np.zeros(self.n, np.int32)
        code = self.lib.solver_hipmf_get_permutation(self.h, p)
        if code != 0:
            raise self._err(code, "solver_hipmf_get_permutation")
        return p

    def stats(self):
        i, d = np.zeros(16, np.int64), np.zeros(16)
        code = self.lib.solver_hipmf_get_stats(self.h, i, d)
        if code != 0:
            raise self._err(code, "solver_hipmf_get_stats")
        out = {k: int(v) for k, v in zip(ISTAT_NAMES, i)}
        out.update({k: float(v) for k, v in zip(DSTAT_NAMES, d)})
        return out

    COUNTERS = {"rematch": 0, "weak_diagonal_rows": 1, "fused_fallbacks": 2, "persistent_bytes": 3, "arena_bytes": 4, "symmetric_ldlt": 5, "sym_expanded": 6, "chain_fallbacks": 7, "mid_fronts": 8, "plan_digest": 9, "tagged_solve": 10, "gate_waits": 11, "wave_fronts": 12, "leaf_fronts": 13, "split_slabs": 14, "event_fence_free": 15, "block_groups": 16, "sym_weak_diagonal": 17, "bcast_sliced_bytes": 18, "krylov_iterations": 19}

    OPTIONS = {"matching": 0, "pivoting": 1, "hybrid_memory": 2, "error_estimates": 3, "condition_numbers": 4, "sym_recheck": 5}

    def set_option(self, name, value):
        """solver_hipmf_set_option (before initialize): the LinSolParams fields the initialize signature does not carry."""
        return int(self.lib.solver_hipmf_set_option(self.h, self.OPTIONS[name], float(value)))

    def counter(self, name):
        return int(self.lib.solver_hipmf_get_counter(self.h, self.COUNTERS[name]))

    def reset_timers(self):
        self.lib.solver_hipmf_reset_timers(self.h)

    # ---- device-resident operands (bench / multi-GPU) -------------------------------------------
    def dev_alloc(self, nbytes):
        p = self.lib.hipmf_device_malloc(nbytes)
        if not p:
            raise MemoryError("hipmf_device_malloc(%d)" % nbytes)
        return p

    def dev_free(self, p):
        self.lib.hipmf_device_free(p)

    def h2d(self, dptr, arr):
        a = np.ascontiguousarray(arr)
        code = self.lib.hipmf_memcpy_h2d(dptr, a.ctypes.data_as(C.c_void_p), a.nbytes)
        if code != 0:
            raise self._err(code, "hipmf_memcpy_h2d")

    def d2h(self, arr, dptr):
        assert arr.flags["C_CONTIGUOUS"]
        code = self.lib.hipmf_memcpy_d2h(arr.ctypes.data_as(C.c_void_p), dptr, arr.nbytes)
        if code != 0:
            raise self._err(code, "hipmf_memcpy_d2h")

    def factorize_device(self, d_values):
        return self.lib.solver_hipmf_factorize_device(self.h, d_values)

    def factorize_mapped_device(self, d_input_values):
        """Numeric factorisation from a DEVICE array of input values (e.g. COO triplets) through the installed value map."""
        return self.lib.solver_hipmf_factorize_mapped_device(self.h, d_input_values)

    def factor_buffers(self):
        """(pointer, bytes) of the device buffers that hold the numeric factor: persistent part of the front pool, local row
        interchanges, scaling, pivots.  A peer that ran `initialize` on the same structure can be handed their contents and then
        `adopt_factor`."""
        ptrs = (C.c_void_p * 4)()
        sizes = (C.c_int64 * 4)()
        k = self.lib.solver_hipmf_factor_parts(self.h, 4, ptrs, sizes)
        if k != 4:
            raise self._err(k, "solver_hipmf_factor_parts")
        return [(int(ptrs[i] or 0), int(sizes[i])) for i in range(4)]

    def broadcast_factor(self, comm, root, rank):
        """RCCL broadcast of the factor (and the matrix values) from `root`; returns (seconds, bytes)."""
        sec, nb = C.c_double(0.0), C.c_int64(0)
        code = self.lib.solver_hipmf_broadcast_factor(self.h, comm, root, rank, C.byref(sec), C.byref(nb))
        if code != 0:
            raise self._err(code, "solver_hipmf_broadcast_factor")
        return sec.value, nb.value

    def solve_many_sharded(self, d_x, d_rhs, nrhs_total, nranks, rank, ld=None):
        first, count = C.c_int32(0), C.c_int32(0)
        code = self.lib.solver_hipmf_solve_many_sharded(self.h, d_x, d_rhs, nrhs_total, ld or self.n, nranks, rank, C.byref(first), C.byref(count))
        if code != 0:
            raise self._err(code, "solver_hipmf_solve_many_sharded")
        return first.value, count.value

    def adopt_factor(self, d_values):
        """Declare the factor buffers (filled by a peer) valid; d_values: the matrix values on the device (refinement SpMV)."""
        return self.lib.solver_hipmf_adopt_factor(self.h, d_values)

    def prepare_solve_many(self, nrhs):
        """Allocate and touch the block buffers of a later many-RHS solve ahead of time (any time after initialize)."""
        code = self.lib.solver_hipmf_prepare_solve_many(self.h, int(nrhs))
        if code != 0:
            raise self._err(code, "solver_hipmf_prepare_solve_many")

    def solve_device(self, d_x, d_rhs, nrhs=1, ld=None):
        code = self.lib.solver_hipmf_solve_device(self.h, d_x, d_rhs, nrhs, ld or self.n)
        if code != 0:
            raise self._err(code, "solver_hipmf_solve_device")
