"""ctypes binding of include/russell_hipmf.h (the C-ABI of the HIP multifrontal backend).

`load()` opens russell_amd/lib/librussell_hipmf.so -- the gfx950 build.  There is no CPU fallback:
if the shared library is missing or no HIP device is visible, the product raises.
(`load(path)` with an explicit path is used by the development-only emulator tests.)
"""
import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
DEFAULT_LIB = os.path.join(_HERE, "lib", "librussell_hipmf.so")

i32p = np.ctypeslib.ndpointer(dtype=np.int32, flags="C_CONTIGUOUS")
i64p = np.ctypeslib.ndpointer(dtype=np.int64, flags="C_CONTIGUOUS")
f64p = np.ctypeslib.ndpointer(dtype=np.float64, flags="C_CONTIGUOUS")

# every symbol include/russell_hipmf.h declares
SYMBOLS = {
    "solver_hipmf_new": (C.c_void_p, []),
    "solver_hipmf_drop": (None, [C.c_void_p]),
    "solver_hipmf_initialize": (C.c_int32, [C.c_void_p, C.c_int32, C.c_int32, C.c_double, C.c_int32, C.c_int32, C.c_int32,
                                            C.c_int32, C.c_int32, i32p, i32p, C.c_void_p]),
    "solver_hipmf_factorize": (C.c_int32, [C.c_void_p, C.POINTER(C.c_int32), C.POINTER(C.c_int32), C.POINTER(C.c_int32),
                                           C.POINTER(C.c_double), C.POINTER(C.c_double), C.POINTER(C.c_double), C.c_int32,
                                           C.c_int32, f64p]),
    "solver_hipmf_solve": (C.c_int32, [C.c_void_p, f64p, f64p, C.c_int32]),
    "solver_hipmf_solve_many": (C.c_int32, [C.c_void_p, f64p, f64p, C.c_int32, C.c_int32, C.c_int32]),
    "solver_hipmf_prepare_solve_many": (C.c_int32, [C.c_void_p, C.c_int32]),
    "solver_hipmf_factorize_device": (C.c_int32, [C.c_void_p, C.c_void_p]),
    "solver_hipmf_solve_device": (C.c_int32, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32, C.c_int32]),
    "solver_hipmf_mat_vec_mul": (C.c_int32, [C.c_void_p, f64p, C.c_double, f64p]),
    "solver_hipmf_get_permutation": (C.c_int32, [C.c_void_p, i32p]),
    "solver_hipmf_set_value_map": (C.c_int32, [C.c_void_p, C.c_int32, i32p, i32p]),
    "solver_hipmf_factorize_mapped": (C.c_int32, [C.c_void_p, C.POINTER(C.c_int32), C.POINTER(C.c_int32), C.POINTER(C.c_int32),
                                                  C.POINTER(C.c_double), C.POINTER(C.c_double), C.POINTER(C.c_double), C.c_int32,
                                                  C.c_int32, f64p]),
    "solver_hipmf_factorize_mapped_device": (C.c_int32, [C.c_void_p, C.c_void_p]),
    "hipmf_max_product_matching": (C.c_int32, [C.c_int32, i32p, i32p, f64p, i32p, f64p, f64p]),
    "hipmf_paired_matching": (C.c_int32, [C.c_int32, i32p, i32p, f64p, i32p, f64p, f64p]),
    "solver_hipmf_get_stats": (C.c_int32, [C.c_void_p, i64p, f64p]),
    "solver_hipmf_reset_timers": (C.c_int32, [C.c_void_p]),
    "complex_solver_hipmf_new": (C.c_void_p, []),
    "complex_solver_hipmf_drop": (None, [C.c_void_p]),
    "complex_solver_hipmf_initialize": (C.c_int32, [C.c_void_p, C.c_int32, C.c_int32, C.c_double, C.c_int32, C.c_int32, C.c_int32, C.c_int32,
                                                     i32p, i32p, C.c_void_p]),
    "complex_solver_hipmf_factorize": (C.c_int32, [C.c_void_p, C.POINTER(C.c_int32), C.POINTER(C.c_int32), C.POINTER(C.c_int32),
                                                    C.POINTER(C.c_double), C.POINTER(C.c_double), C.POINTER(C.c_double), C.POINTER(C.c_double),
                                                    C.c_int32, C.c_int32, f64p]),
    "complex_solver_hipmf_get_determinant": (C.c_int32, [C.c_void_p, C.POINTER(C.c_double), C.POINTER(C.c_double), C.POINTER(C.c_double)]),
    "complex_solver_hipmf_solve": (C.c_int32, [C.c_void_p, f64p, f64p, C.c_int32]),
    "complex_solver_hipmf_set_value_map": (C.c_int32, [C.c_void_p, C.c_int32, i32p, i32p]),
    "complex_solver_hipmf_factorize_mapped": (C.c_int32, [C.c_void_p, C.POINTER(C.c_int32), C.POINTER(C.c_int32), C.POINTER(C.c_int32),
                                                           C.POINTER(C.c_double), C.c_int32, f64p]),
    "complex_solver_hipmf_get_stats": (C.c_int32, [C.c_void_p, i64p, f64p]),
    "complex_solver_hipmf_get_counter": (C.c_int64, [C.c_void_p, C.c_int32]),
    "complex_solver_hipmf_last_error": (C.c_char_p, [C.c_void_p]),
    "solver_hipmf_get_counter": (C.c_int64, [C.c_void_p, C.c_int32]),
    "solver_hipmf_set_option": (C.c_int32, [C.c_void_p, C.c_int32, C.c_double]),
    "solver_hipmf_get_option": (C.c_int32, [C.c_void_p, C.c_int32, C.POINTER(C.c_double)]),
    "solver_hipmf_factor_parts": (C.c_int32, [C.c_void_p, C.c_int32, C.POINTER(C.c_void_p), C.POINTER(C.c_int64)]),
    "hipmf_comm_unique_id": (C.c_int32, [C.c_void_p]),
    "hipmf_comm_init_rank": (C.c_int32, [C.POINTER(C.c_void_p), C.c_int32, C.c_void_p, C.c_int32]),
    "hipmf_comm_destroy": (None, [C.c_void_p]),
    "solver_hipmf_broadcast_factor": (C.c_int32, [C.c_void_p, C.c_void_p, C.c_int32, C.c_int32, C.POINTER(C.c_double), C.POINTER(C.c_int64)]),
    "solver_hipmf_solve_many_sharded": (C.c_int32, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32, C.c_int32, C.c_int32, C.c_int32,
                                                     C.POINTER(C.c_int32), C.POINTER(C.c_int32)]),
    "solver_hipmf_adopt_factor": (C.c_int32, [C.c_void_p, C.c_void_p]),
    "solver_hipmf_last_error": (C.c_char_p, [C.c_void_p]),
    "hipmf_device_malloc": (C.c_void_p, [C.c_size_t]),
    "hipmf_device_free": (None, [C.c_void_p]),
    "hipmf_memcpy_h2d": (C.c_int32, [C.c_void_p, C.c_void_p, C.c_size_t]),
    "hipmf_memcpy_d2h": (C.c_int32, [C.c_void_p, C.c_void_p, C.c_size_t]),
    "hipmf_device_synchronize": (C.c_int32, []),
    "hipmf_device_count": (C.c_int32, []),
    "hipmf_device_copy_bandwidth": (C.c_int32, [C.c_int64, C.c_int32, C.POINTER(C.c_double)]),
    "hipmf_device_mem_info": (C.c_int32, [C.POINTER(C.c_size_t), C.POINTER(C.c_size_t)]),
    "hipmf_device_mfma_rate": (C.c_int32, [C.c_int32, C.c_int32, C.POINTER(C.c_double)]),
    "hipmf_set_device": (C.c_int32, [C.c_int32]),
    "hipmf_fdm_new": (C.c_void_p, [C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_void_p]),
    "hipmf_fdm_drop": (None, [C.c_void_p]),
    "hipmf_fdm_dims": (C.c_int32, [C.c_void_p, C.POINTER(C.c_int64), C.POINTER(C.c_int64), C.POINTER(C.c_int64), C.POINTER(C.c_int64)]),
    "hipmf_fdm_structure_device": (C.c_int32, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    "hipmf_fdm_values_device": (C.c_int32, [C.c_void_p, C.c_double, C.c_double, C.c_double, C.c_double, C.c_double, C.c_double, C.c_double,
                                            C.c_void_p, C.c_void_p]),
    "hipmf_fdm_lmm_dims": (C.c_int32, [C.c_void_p, C.POINTER(C.c_int64), C.POINTER(C.c_int64), C.POINTER(C.c_int64)]),
    "hipmf_fdm_lmm_structure_device": (C.c_int32, [C.c_void_p, C.c_void_p, C.c_void_p]),
    "hipmf_fdm_lmm_values_device": (C.c_int32, [C.c_void_p, C.c_double, C.c_double, C.c_double, C.c_double, C.c_double, C.c_double, C.c_double, C.c_void_p]),
}

_cache = {}


def load(path=None):
    path = path or os.environ.get("HIPMF_DEV_LIB") or DEFAULT_LIB  # (HIPMF_DEV_LIB: a variant build under study, tools/gpu/*.sh)
    if path in _cache:
        return _cache[path]
    if not os.path.exists(path):
        raise RuntimeError(
            "russell_amd: %s is missing -- build the HIP extension first (python -c 'import __graft_entry__ as g; g.build()'). "
            "There is no CPU fallback." % path)
    lib = C.CDLL(path)
    for name, (res, args) in SYMBOLS.items():
        fn = getattr(lib, name)  # AttributeError if the library does not export a declared symbol
        fn.restype = res
        fn.argtypes = args
    _cache[path] = lib
    return lib
