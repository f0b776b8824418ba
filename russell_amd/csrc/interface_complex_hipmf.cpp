// interface_complex_hipmf.cpp -- complex (Complex64) twin of the C-ABI: complex_solver_hipmf_{new,drop,initialize,factorize,solve}
// with the shape of the reference's complex shim (/root/reference/russell_sparse/c_code/interface_complex_umfpack.c:82-248;
// values are interleaved (re, im) pairs, COMPLEX64 of constants.h:18; Rust side: complex_solver_umfpack.rs, complex_lin_solver.rs:12-104).
//
// The numeric path is the real one: a complex system A z = c of order n is solved as its REAL-EQUIVALENT system of order 2n with the
// unknowns (Re z_k, Im z_k) interleaved,
//     a + i b at (i, j)   ->   [ a  -b ; b  a ]   at rows 2i, 2i+1 / columns 2j, 2j+1,
// so the interleaved complex vectors ARE the real vectors (no copy), and the caller's complex values are expanded ON THE DEVICE
// (signed value map + k_gather_values): a factorisation moves nnz x 16 bytes over PCIe, nothing is expanded on the host.
// A complex SYMMETRIC matrix handed over as its lower triangle has an unsymmetric real-equivalent form: the mirrored entries are
// written out in the pattern (general LU on the device).
// Round 4: the real path runs in its PAIRED mode (NumericOptions.complex_pairs): the ordering and the matching work on the graph / the
// moduli of the complex matrix and keep rows 2 k, 2 k + 1 together, and every pivot search takes a pair of rows at a time (the row with
// the largest real or imaginary part in the column, then its partner) -- a complex LU with partial pivoting carried out in real
// arithmetic, whose complex pivots give the determinant (the real pivots alone only give |det A|^2).  HIPMF_COMPLEX_PAIRS=0: the plain
// real-equivalent factorisation of rounds 2 - 3 (no determinant).
#include <hipmf_device_rt.h>

#include <algorithm>
#include <cstdio>
#include <new>
#include <string>
#include <vector>

#include "numeric.hpp"
#include "../../include/russell_hipmf.h"

using namespace hipmf;

struct InterfaceComplexHIPMF {
    Solver solver;       // the real-equivalent system of order 2 n
    int32_t n = 0;
    int64_t nnz = 0;     // stored complex entries of the caller's CSR
    bool sym_lower = false;
    // real-equivalent CSR entry q <- (stored complex entry, code): 0 +re, 1 -im, 2 +im, 3 +re
    std::vector<int32_t> src_entry;
    std::vector<int8_t> src_code;
    int32_t effective_ordering = 0;
    bool triplet_map = false; // the installed map reads the caller's COO triplets (complex_solver_hipmf_set_value_map)
};

// No C++ exception crosses the C boundary (the same rule as interface_hipmf.cpp): a failed host allocation comes back as ERROR_MALLOC
// with a message the reference's harness recognises as a memory error (stats_lin_sol.rs:334-340), anything else as ERROR_HIPMF_SYMBOLIC.
template <typename Fn> static int32_t guarded(struct InterfaceComplexHIPMF *h, Fn fn) {
    try {
        return fn();
    } catch (const std::bad_alloc &) {
        if (h) h->solver.last_error = "Not enough memory: a host allocation failed";
        return ERROR_MALLOC;
    } catch (const std::exception &e) {
        if (h) h->solver.last_error = std::string("internal error: ") + e.what();
        return ERROR_HIPMF_SYMBOLIC;
    }
}

namespace {
// builds the real-equivalent CSR pattern (rows 2i, 2i+1) of the complex CSR (mirroring the strict lower triangle when sym_lower) and,
// per real entry, where its value comes from
int32_t build_real_equivalent(InterfaceComplexHIPMF *h, const int32_t *rp, const int32_t *ci, std::vector<int32_t> &rp2, std::vector<int32_t> &ci2) {
    const int32_t n = h->n;
    // full pattern of the complex matrix: per row, (column, stored entry)
    std::vector<int64_t> cnt((size_t)n + 1, 0);
    for (int32_t i = 0; i < n; i++)
        for (int32_t k = rp[i]; k < rp[i + 1]; k++) {
            const int32_t j = ci[k];
            if (j < 0 || j >= n) return ERROR_HIPMF_INVALID_MATRIX;
            if (h->sym_lower && j > i) return ERROR_HIPMF_INVALID_MATRIX; // lower storage promised
            cnt[(size_t)i + 1]++;
            if (h->sym_lower && j != i) cnt[(size_t)j + 1]++;
        }
    for (int32_t i = 0; i < n; i++) cnt[(size_t)i + 1] += cnt[i];
    if (4 * cnt[n] > 0x7fffffffLL) return ERROR_HIPMF_INVALID_MATRIX;
    std::vector<int32_t> fcol((size_t)cnt[n]), fsrc((size_t)cnt[n]);
    {
        std::vector<int64_t> w(cnt.begin(), cnt.end() - 1);
        // rows ascending, stored entries of a row ascending in column: the mirrored entries (j, i), i > j, arrive in ascending i after the
        // row's own entries (columns <= j), so every full row is ascending without a sort
        for (int32_t i = 0; i < n; i++)
            for (int32_t k = rp[i]; k < rp[i + 1]; k++) fcol[(size_t)w[i]] = ci[k], fsrc[(size_t)w[i]++] = k;
        if (h->sym_lower)
            for (int32_t i = 0; i < n; i++)
                for (int32_t k = rp[i]; k < rp[i + 1]; k++)
                    if (ci[k] != i) fcol[(size_t)w[ci[k]]] = i, fsrc[(size_t)w[ci[k]]++] = k;
    }
    rp2.assign((size_t)2 * n + 1, 0);
    ci2.resize((size_t)4 * cnt[n]);
    h->src_entry.resize(ci2.size());
    h->src_code.resize(ci2.size());
    int64_t q = 0;
    for (int32_t i = 0; i < n; i++)
        for (int half = 0; half < 2; half++) {
            rp2[(size_t)2 * i + half] = (int32_t)q;
            for (int64_t e = cnt[i]; e < cnt[(size_t)i + 1]; e++) {
                ci2[(size_t)q] = 2 * fcol[(size_t)e], h->src_entry[(size_t)q] = fsrc[(size_t)e], h->src_code[(size_t)q] = (int8_t)(half == 0 ? 0 : 2), q++;
                ci2[(size_t)q] = 2 * fcol[(size_t)e] + 1, h->src_entry[(size_t)q] = fsrc[(size_t)e], h->src_code[(size_t)q] = (int8_t)(half == 0 ? 1 : 3), q++;
            }
        }
    rp2[(size_t)2 * n] = (int32_t)q;
    return SUCCESSFUL_EXIT;
}

// signed value map of the real solver: real entry q = sum over the caller's inputs behind its complex entry.  tri_ptr / tri_idx
// (optional): complex CSR entry c = sum of the caller's triplets tri_idx[tri_ptr[c] .. tri_ptr[c+1]) (COO with duplicates);
// without them input k IS complex entry k.  Inputs are interleaved (re, im): value 2k / 2k+1.
int32_t install_map(InterfaceComplexHIPMF *h, int64_t nin_complex, const int32_t *tri_ptr, const int32_t *tri_idx) {
    const size_t nq = h->src_entry.size();
    std::vector<int32_t> seg_ptr(nq + 1, 0), seg_idx;
    seg_idx.reserve(tri_ptr ? 4 * (size_t)nin_complex : nq);
    for (size_t q = 0; q < nq; q++) {
        const int32_t c = h->src_entry[q];
        const int code = h->src_code[q];
        const int32_t t0 = tri_ptr ? tri_ptr[c] : c, t1 = tri_ptr ? tri_ptr[c + 1] : c + 1;
        for (int32_t t = t0; t < t1; t++) {
            const int32_t k = tri_ptr ? tri_idx[t] : t;
            const int32_t v = (code == 0 || code == 3) ? 2 * k : 2 * k + 1;
            seg_idx.push_back(code == 1 ? ~v : v);
        }
        if (seg_idx.size() > 0x7fffffffULL) return ERROR_HIPMF_INVALID_VALUE;
        seg_ptr[q + 1] = (int32_t)seg_idx.size();
    }
    return h->solver.set_value_map(2 * nin_complex, seg_ptr.data(), seg_idx.data(), true);
}
} // namespace

extern "C" {

struct InterfaceComplexHIPMF *complex_solver_hipmf_new(void) {
    try {
        int ndev = 0;
        if (hipGetDeviceCount(&ndev) != hipSuccess || ndev < 1) return nullptr;
        return new (std::nothrow) InterfaceComplexHIPMF();
    } catch (...) {
        return nullptr;
    }
}

void complex_solver_hipmf_drop(struct InterfaceComplexHIPMF *h) {
    if (!h) return;
    try {
        h->solver.release();
    } catch (...) {
    }
    delete h;
}

static int32_t c_initialize_body(struct InterfaceComplexHIPMF *h, int32_t ordering, int32_t scaling, double pivot_epsilon,
                                        int32_t refinement_nstep, C_BOOL verbose, C_BOOL general_symmetric, int32_t ndim,
                                        const int32_t *row_pointers, const int32_t *col_indices, const double *values) {
    if (!h || !row_pointers || !col_indices) return ERROR_NULL_POINTER;
    if (h->solver.initialized) return ERROR_ALREADY_INITIALIZED;
    if (ndim < 1 || ndim > 0x3fffffff) return ERROR_HIPMF_INVALID_MATRIX;
    if (validate_csr(ndim, row_pointers, col_indices) != 0) return ERROR_HIPMF_INVALID_MATRIX;
    h->n = ndim;
    h->nnz = row_pointers[ndim];
    h->sym_lower = general_symmetric == 1;
    std::vector<int32_t> rp2, ci2;
    int32_t code = build_real_equivalent(h, row_pointers, col_indices, rp2, ci2);
    if (code != SUCCESSFUL_EXIT) return code;
    SymbolicOptions so;
    so.ordering = (ordering == HIPMF_ORDERING_NONE) ? ORDERING_NATURAL
                  : (ordering == HIPMF_ORDERING_AMD ? ORDERING_MIN_DEGREE : (ordering == HIPMF_ORDERING_BEST ? ORDERING_BEST : ORDERING_NESTED_DISSECTION));
    NumericOptions no;
    no.scaling = (scaling < 0 || scaling > 2) ? HIPMF_SCALE_SUM : scaling;
    if (pivot_epsilon >= 0.0) no.pivot_epsilon = pivot_epsilon;
    if (refinement_nstep >= 0) no.refinement_nstep = refinement_nstep;
    no.verbose = verbose == 1;
    no.complex_pairs = true;
    if (const char *e = getenv("HIPMF_COMPLEX_PAIRS")) no.complex_pairs = atoi(e) != 0;
    h->effective_ordering = (ordering == HIPMF_ORDERING_NONE || ordering == HIPMF_ORDERING_AMD || ordering == HIPMF_ORDERING_BEST) ? ordering : HIPMF_ORDERING_NESTED_DISSECTION; // (BEST: resolved once the analysis has chosen)
    // the values (when given) let the analysis apply the maximum-product matching to a weak diagonal, as for real matrices
    std::vector<double> v2;
    if (values) {
        v2.resize(ci2.size());
        for (size_t q = 0; q < v2.size(); q++) {
            const double re = values[2 * (size_t)h->src_entry[q]], im = values[2 * (size_t)h->src_entry[q] + 1];
            const int c = h->src_code[q];
            v2[q] = (c == 0 || c == 3) ? re : (c == 1 ? -im : im);
        }
    }
    code = h->solver.initialize(2 * ndim, rp2.data(), ci2.data(), false, so, no, values ? v2.data() : nullptr);
    if (code != SUCCESSFUL_EXIT) return code;
    return install_map(h, h->nnz, nullptr, nullptr);
}

int32_t complex_solver_hipmf_initialize(struct InterfaceComplexHIPMF *h, int32_t ordering, int32_t scaling, double pivot_epsilon,
                                        int32_t refinement_nstep, C_BOOL verbose, C_BOOL general_symmetric, int32_t ndim,
                                        const int32_t *row_pointers, const int32_t *col_indices, const double *values) {
    return guarded(h, [&]() { return c_initialize_body(h, ordering, scaling, pivot_epsilon, refinement_nstep, verbose, general_symmetric, ndim, row_pointers, col_indices, values); });
}

static int32_t c_set_value_map_body(struct InterfaceComplexHIPMF *h, int32_t nnz_in, const int32_t *seg_ptr, const int32_t *seg_idx) {
    if (!h || !seg_ptr || !seg_idx) return ERROR_NULL_POINTER;
    if (!h->solver.initialized) return ERROR_NEED_INITIALIZATION;
    if (nnz_in < 1 || seg_ptr[0] != 0 || seg_ptr[h->nnz] != nnz_in) return ERROR_HIPMF_INVALID_VALUE;
    for (int64_t c = 0; c < h->nnz; c++)
        if (seg_ptr[c + 1] < seg_ptr[c]) return ERROR_HIPMF_INVALID_VALUE;
    for (int32_t t = 0; t < nnz_in; t++)
        if (seg_idx[t] < 0 || seg_idx[t] >= nnz_in) return ERROR_HIPMF_INVALID_VALUE;
    h->triplet_map = true;
    return install_map(h, nnz_in, seg_ptr, seg_idx);
}

int32_t complex_solver_hipmf_set_value_map(struct InterfaceComplexHIPMF *h, int32_t nnz_in, const int32_t *seg_ptr, const int32_t *seg_idx) {
    return guarded(h, [&]() { return c_set_value_map_body(h, nnz_in, seg_ptr, seg_idx); });
}

static int32_t finish(struct InterfaceComplexHIPMF *h, int32_t code, int32_t *effective_ordering, int32_t *effective_scaling, int32_t *num_perturbed,
                      double *rcond) {
    if (effective_ordering) {
        *effective_ordering = h->effective_ordering;
        if (h->effective_ordering == HIPMF_ORDERING_BEST) *effective_ordering = h->solver.S.best_chose_min_degree ? HIPMF_ORDERING_AMD : HIPMF_ORDERING_NESTED_DISSECTION;
    }
    if (effective_scaling) *effective_scaling = h->solver.opt.scaling;
    if (num_perturbed) *num_perturbed = h->solver.n_perturbed;
    if (rcond) {
        *rcond = 0.0;
        if (code == SUCCESSFUL_EXIT) (void)h->solver.rcond_estimate(rcond); // (estimate of the real-equivalent matrix)
    }
    return code;
}

// determinant_coefficient_real / _imag / _exponent: det A = (real + i imag) x 10^exponent, the triple umfpack_zi_get_determinant hands
// to interface_complex_umfpack.c:187-195 (zeros when compute_determinant = 0, as there: :196-200)
static int32_t determinant_out(struct InterfaceComplexHIPMF *h, int32_t code, C_BOOL compute_determinant, double *det_re, double *det_im, double *det_exp) {
    if (det_re) *det_re = 0.0;
    if (det_im) *det_im = 0.0;
    if (det_exp) *det_exp = 0.0;
    if (code != SUCCESSFUL_EXIT || compute_determinant != 1) return code;
    return h->solver.determinant_complex(det_re, det_im, det_exp, nullptr);
}

static int32_t c_factorize_body(struct InterfaceComplexHIPMF *h, int32_t *effective_ordering, int32_t *effective_scaling,
                                       int32_t *num_perturbed_pivots, double *rcond_estimate, double *det_re, double *det_im, double *det_exp,
                                       C_BOOL compute_determinant, C_BOOL verbose, const double *values) {
    if (!h || !values) return ERROR_NULL_POINTER;
    if (!h->solver.initialized) return ERROR_NEED_INITIALIZATION;
    if (compute_determinant == 1 && !h->solver.opt.complex_pairs) return ERROR_NOT_AVAILABLE; // det of the plain real-equivalent form is |det A|^2: the phase is lost
    if (h->triplet_map) { // a triplet map is installed: plain CSR values need the identity map back
        int32_t c = install_map(h, h->nnz, nullptr, nullptr);
        if (c != SUCCESSFUL_EXIT) return c;
        h->triplet_map = false;
    }
    h->solver.opt.verbose = verbose == 1;
    const int32_t code = finish(h, h->solver.factorize_mapped(values, false), effective_ordering, effective_scaling, num_perturbed_pivots, rcond_estimate);
    return determinant_out(h, code, compute_determinant, det_re, det_im, det_exp);
}

int32_t complex_solver_hipmf_factorize(struct InterfaceComplexHIPMF *h, int32_t *effective_ordering, int32_t *effective_scaling,
                                       int32_t *num_perturbed_pivots, double *rcond_estimate, double *determinant_coefficient_real,
                                       double *determinant_coefficient_imag, double *determinant_exponent, C_BOOL compute_determinant, C_BOOL verbose,
                                       const double *values) {
    return guarded(h, [&]() {
        return c_factorize_body(h, effective_ordering, effective_scaling, num_perturbed_pivots, rcond_estimate, determinant_coefficient_real,
                                determinant_coefficient_imag, determinant_exponent, compute_determinant, verbose, values);
    });
}

static int32_t c_factorize_mapped_body(struct InterfaceComplexHIPMF *h, int32_t *effective_ordering, int32_t *effective_scaling,
                                              int32_t *num_perturbed_pivots, double *rcond_estimate, C_BOOL verbose, const double *input_values) {
    if (!h || !input_values) return ERROR_NULL_POINTER;
    if (!h->solver.initialized) return ERROR_NEED_INITIALIZATION;
    h->solver.opt.verbose = verbose == 1;
    return finish(h, h->solver.factorize_mapped(input_values, false), effective_ordering, effective_scaling, num_perturbed_pivots, rcond_estimate);
}

int32_t complex_solver_hipmf_factorize_mapped(struct InterfaceComplexHIPMF *h, int32_t *effective_ordering, int32_t *effective_scaling,
                                              int32_t *num_perturbed_pivots, double *rcond_estimate, C_BOOL verbose, const double *input_values) {
    return guarded(h, [&]() { return c_factorize_mapped_body(h, effective_ordering, effective_scaling, num_perturbed_pivots, rcond_estimate, verbose, input_values); });
}

// the determinant of the last factorisation (complex_solver_hipmf_factorize_mapped has no determinant arguments: Radau5 never asks)
int32_t complex_solver_hipmf_get_determinant(struct InterfaceComplexHIPMF *h, double *determinant_coefficient_real, double *determinant_coefficient_imag,
                                             double *determinant_exponent) {
    if (!h) return ERROR_NULL_POINTER;
    return guarded(h, [&]() { return h->solver.determinant_complex(determinant_coefficient_real, determinant_coefficient_imag, determinant_exponent, nullptr); });
}

static int32_t c_solve_body(struct InterfaceComplexHIPMF *h, double *x, const double *rhs, C_BOOL verbose) {
    if (!h || !x || !rhs) return ERROR_NULL_POINTER;
    if (!h->solver.factorized) return ERROR_NEED_FACTORIZATION;
    h->solver.opt.verbose = verbose == 1;
    return h->solver.solve(x, rhs, 1, 2 * (int64_t)h->n, false); // interleaved complex vectors = vectors of the real-equivalent system
}

int32_t complex_solver_hipmf_solve(struct InterfaceComplexHIPMF *h, double *x, const double *rhs, C_BOOL verbose) {
    return guarded(h, [&]() { return c_solve_body(h, x, rhs, verbose); });
}

const char *complex_solver_hipmf_last_error(struct InterfaceComplexHIPMF *h) { return h ? h->solver.last_error.c_str() : "null solver"; }

static int32_t c_get_stats_body(struct InterfaceComplexHIPMF *h, int64_t *is, double *ds) {
    if (!h || !is || !ds) return ERROR_NULL_POINTER;
    if (!h->solver.initialized) return ERROR_NEED_INITIALIZATION;
    const Solver &s = h->solver;
    for (int i = 0; i < 16; i++) is[i] = 0, ds[i] = 0.0;
    is[0] = h->n, is[1] = h->nnz, is[2] = s.S.nsuper, is[3] = s.S.nlevels, is[4] = s.S.nnz_l, is[5] = s.S.nnz_u, is[6] = s.S.max_front;
    is[7] = s.S.max_pivots, is[8] = s.n_perturbed, is[9] = s.n_zero_pivot, is[10] = s.refinement_steps_done;
    is[11] = s.times.n_kernel_launches_factor, is[12] = s.times.n_kernel_launches_solve, is[13] = s.pool_doubles * 8, is[14] = s.matched ? 1 : 0;
    is[15] = s.fused_fallbacks;
    ds[0] = s.S.flops, ds[1] = s.S.flops_gemm, ds[2] = s.S.seconds_ordering, ds[3] = s.S.seconds_total, ds[4] = s.times.scale_assemble_ms;
    ds[5] = s.times.factor_ms, ds[6] = s.times.fwd_ms, ds[7] = s.times.bwd_ms, ds[8] = s.times.solve_total_ms, ds[9] = s.last_residual_inf;
    return SUCCESSFUL_EXIT;
}

int32_t complex_solver_hipmf_get_stats(struct InterfaceComplexHIPMF *h, int64_t *is, double *ds) {
    return guarded(h, [&]() { return c_get_stats_body(h, is, ds); });
}

int64_t complex_solver_hipmf_get_counter(struct InterfaceComplexHIPMF *h, int32_t which) {
    if (!h || !h->solver.initialized) return -1;
    const Solver &s = h->solver;
    switch (which) {
    case HIPMF_COUNTER_REMATCH: return s.rematch_count;
    case HIPMF_COUNTER_WEAK_DIAGONAL_ROWS: return s.n_weak_diag;
    case HIPMF_COUNTER_FUSED_FALLBACKS: return s.fused_fallbacks;
    case HIPMF_COUNTER_PERSISTENT_BYTES: return s.S.persist_doubles * 8;
    case HIPMF_COUNTER_ARENA_BYTES: return s.S.temp_doubles * 8;
    case HIPMF_COUNTER_SYMMETRIC_LDLT: return s.S.sym_mode ? 1 : 0;
    case HIPMF_COUNTER_CHAIN_FALLBACKS: return s.chain_fallbacks;
    case HIPMF_COUNTER_MID_FRONTS: return s.mid_front_count;
    case HIPMF_COUNTER_TAGGED_SOLVE: return s.tagged_solve() ? 1 : 0;
    case HIPMF_COUNTER_GATE_WAITS: return s.gate_waits;
    case HIPMF_COUNTER_WAVE_FRONTS: return s.wave_front_count;
    case HIPMF_COUNTER_LEAF_FRONTS: return s.leaf_front_count();
    case HIPMF_COUNTER_SPLIT_SLABS: return s.split_slab_count();
    default: return -1;
    }
}

} // extern "C"
