// interface_hipmf.cpp -- the extern "C" boundary declared in include/russell_hipmf.h.
// Behaviour follows the reference shims: NULL-safe drop, initialize exactly once
// (ERROR_ALREADY_INITIALIZED, interface_umfpack.c:95-97), factorize needs initialize
// (ERROR_NEED_INITIALIZATION), solve needs factorize (ERROR_NEED_FACTORIZATION), every phase is
// blocking (stream-synchronised) on return (interface_cudss.cu:383,449,536).
#include <hipmf_device_rt.h>

#include <algorithm>
#include <chrono>
#include <cstdio>
#include <cstring>
#include <mutex>
#include <new>
#ifdef HIPMF_HAVE_RCCL
#include <dlfcn.h>
#include <rccl/rccl.h>
#endif

#include "numeric.hpp"
#include "matching.hpp"
// the C header #defines the same status names as macros: it must come after numeric.hpp
#include "../../include/russell_hipmf.h"

using namespace hipmf;

struct InterfaceHIPMF {
    Solver solver;
    int32_t ordering_requested = 0;
    int32_t effective_ordering = 0;
    // solver_hipmf_set_option (before initialize)
    int32_t opt_matching = 1, opt_pivoting = 0, opt_sym_recheck = 0;
    int64_t bcast_sliced_bytes = 0; // bytes of factor that travelled as slices over all links (solver_hipmf_broadcast_factor, round 6)
    double opt_hybrid = 0.0;
    // a symmetric-lower matrix with a weak diagonal is analysed and factorised as the mirrored GENERAL matrix (with the matching):
    // the caller keeps handing over lower-triangle values, entry k of the handle's CSR is entry emap[k] of the caller's
    bool expanded = false;
    int64_t nnz_lower = 0;
    // a symmetric-lower handle that was initialised WITHOUT values looks at the first values it is asked to factorise (see factorize_body)
    bool sym_unchecked = false;
    SymbolicOptions so_keep;
    // device words of solver_hipmf_broadcast_factor's plan check (8 x int64 header, 2 x int32 status), allocated at initialize: no rank
    // can fail an allocation between two collectives
    int64_t *d_hdr = nullptr;
};

// No C++ exception crosses the C boundary: a failed host allocation (the analysis of a large matrix takes gigabytes) comes back as
// ERROR_MALLOC, the code of the reference's shims for the same event (c_code/constants.h:6), with a message the reference's harness
// recognises as a memory error (stats_lin_sol.rs:334-340); anything else as ERROR_HIPMF_SYMBOLIC.
// the handle's device for the duration of a scope (handles are Send), the caller's device afterwards
struct DeviceGuard {
    int prev = -1, dev;
    bool ok = true;
    explicit DeviceGuard(int d) : dev(d) {
        if (hipGetDevice(&prev) != hipSuccess) prev = -1;
        if (prev != d) ok = hipSetDevice(d) == hipSuccess;
    }
    ~DeviceGuard() {
        if (prev >= 0 && prev != dev) (void)hipSetDevice(prev);
    }
};

template <typename Fn> static int32_t guarded(struct InterfaceHIPMF *h, Fn fn) {
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

extern "C" {

struct InterfaceHIPMF *solver_hipmf_new(void) {
    try { // (the handle's constructor allocates: NULL on any failure, like the reference's *_new, interface_cudss.cu:62-123)
        int ndev = 0;
        if (hipGetDeviceCount(&ndev) != hipSuccess || ndev < 1) return nullptr;
        return new (std::nothrow) InterfaceHIPMF();
    } catch (...) {
        return nullptr;
    }
}

void solver_hipmf_drop(struct InterfaceHIPMF *h) {
    if (!h) return;
    try {
        if (h->d_hdr) (void)hipFree(h->d_hdr);
        h->solver.release();
    } catch (...) {
    }
    delete h;
}

// true when the matrix given as its LOWER triangle has a weak diagonal somewhere: missing, zero or < 1 % of the largest entry of the
// row / column (the criterion of the general path, matching.cpp); false as well when an entry above the diagonal is met
static bool sym_lower_diagonal_is_weak(int32_t ndim, const int32_t *row_pointers, const int32_t *col_indices, const double *values) {
    std::vector<double> rmax((size_t)ndim, 0.0), dg((size_t)ndim, 0.0);
    for (int32_t i = 0; i < ndim; i++)
        for (int32_t k = row_pointers[i]; k < row_pointers[i + 1]; k++) {
            const int32_t j = col_indices[k];
            if (j > i) return false;
            const double a = std::fabs(values[k]);
            rmax[(size_t)i] = std::max(rmax[(size_t)i], a), rmax[(size_t)j] = std::max(rmax[(size_t)j], a);
            if (j == i) dg[(size_t)i] = a;
        }
    for (int32_t i = 0; i < ndim; i++)
        if (dg[(size_t)i] < 0.01 * rmax[(size_t)i] || rmax[(size_t)i] == 0.0) return true;
    return false;
}

// the lower triangle mirrored to general storage, analysed with these values (matching + scaling), the expansion map installed: the
// caller keeps handing over lower-triangle values
static int32_t initialize_expanded(struct InterfaceHIPMF *h, int32_t ndim, const int32_t *row_pointers, const int32_t *col_indices, const double *values,
                                   const SymbolicOptions &so, const NumericOptions &no) {
    const int64_t nl = row_pointers[ndim];
    std::vector<int64_t> cnt((size_t)ndim + 1, 0);
    for (int32_t i = 0; i < ndim; i++)
        for (int32_t k = row_pointers[i]; k < row_pointers[i + 1]; k++) {
            cnt[(size_t)i + 1]++;
            if (col_indices[k] != i) cnt[(size_t)col_indices[k] + 1]++;
        }
    for (int32_t i = 0; i < ndim; i++) cnt[(size_t)i + 1] += cnt[(size_t)i];
    if (cnt[(size_t)ndim] > 0x7fffffffLL) return ERROR_HIPMF_INVALID_MATRIX;
    std::vector<int32_t> rpf((size_t)ndim + 1), cif((size_t)cnt[(size_t)ndim]), emap((size_t)cnt[(size_t)ndim]);
    std::vector<double> vf((size_t)cnt[(size_t)ndim]);
    for (int32_t i = 0; i <= ndim; i++) rpf[(size_t)i] = (int32_t)cnt[(size_t)i];
    // row i of the full matrix: its stored lower entries (columns ascending, <= i), then the mirrored ones (rows r > i ascending):
    // filling row by row in ascending order of the source row keeps every row's columns ascending
    std::vector<int64_t> w(cnt.begin(), cnt.end() - 1);
    for (int32_t i = 0; i < ndim; i++)
        for (int32_t k = row_pointers[i]; k < row_pointers[i + 1]; k++) {
            const int64_t q = w[(size_t)i]++;
            cif[(size_t)q] = col_indices[k], vf[(size_t)q] = values[k], emap[(size_t)q] = k;
        }
    for (int32_t i = 0; i < ndim; i++)
        for (int32_t k = row_pointers[i]; k < row_pointers[i + 1]; k++) {
            const int32_t j = col_indices[k];
            if (j == i) continue;
            const int64_t q = w[(size_t)j]++;
            cif[(size_t)q] = i, vf[(size_t)q] = values[k], emap[(size_t)q] = k;
        }
    int32_t code = h->solver.initialize(ndim, rpf.data(), cif.data(), false, so, no, vf.data());
    if (code == SUCCESSFUL_EXIT) code = h->solver.set_expansion(nl, emap);
    if (code == SUCCESSFUL_EXIT) h->expanded = true, h->nnz_lower = nl;
    else h->solver.release();
    return code;
}

static int32_t initialize_body(struct InterfaceHIPMF *h, int32_t ordering, int32_t scaling, double pivot_epsilon,
                                int32_t refinement_nstep, C_BOOL verbose, C_BOOL general_symmetric, C_BOOL positive_definite,
                                int32_t ndim, const int32_t *row_pointers, const int32_t *col_indices, const double *values) {
    // either flag promises the LOWER triangle of a symmetric matrix (interface_cudss.cu:324-333: SYMMETRIC / SPD + lower view); the
    // tiled fronts are then factorised as L D L^T (positive definite: D > 0, the same arithmetic)
    if (!h || !row_pointers || !col_indices) return ERROR_NULL_POINTER;
    bool sym_lower = general_symmetric == 1 || positive_definite == 1;
    if (h->solver.initialized) return ERROR_ALREADY_INITIALIZED;
    h->expanded = false, h->nnz_lower = 0; // (a handle whose earlier initialize failed half-way starts clean)
    if (sym_lower && general_symmetric != 1 && ndim >= 1 && validate_csr(ndim, row_pointers, col_indices) == 0) {
        // positive_definite alone: LinSolParams::positive_definite is independent of the storage (lin_sol_params.rs:41-42), and the
        // reference's GPU plug-in accepts it with FULL storage (Sym::No) by taking the lower view (solver_cudss.rs:260-261).  A CSR
        // with entries above the diagonal is therefore factorised as the general matrix it is (LU), not refused.
        bool upper = false;
        for (int32_t i = 0; i < ndim && !upper; i++)
            for (int32_t k = row_pointers[i]; k < row_pointers[i + 1]; k++)
                if (col_indices[k] > i) {
                    upper = true;
                    break;
                }
        if (upper) sym_lower = false;
    }
    if (h->solver.initialized) return ERROR_ALREADY_INITIALIZED;
    if (ndim < 1) return ERROR_HIPMF_INVALID_MATRIX;
    SymbolicOptions so;
    so.ordering = (ordering == HIPMF_ORDERING_NONE) ? ORDERING_NATURAL
                  : (ordering == HIPMF_ORDERING_AMD ? ORDERING_MIN_DEGREE : (ordering == HIPMF_ORDERING_BEST ? ORDERING_BEST : ORDERING_NESTED_DISSECTION));
    NumericOptions no;
    no.scaling = (scaling < 0 || scaling > 2) ? HIPMF_SCALE_SUM : scaling;
    if (pivot_epsilon >= 0.0) no.pivot_epsilon = pivot_epsilon;
    if (refinement_nstep >= 0) no.refinement_nstep = refinement_nstep;
    no.verbose = verbose == 1;
    no.matching = h->opt_matching;
    no.device_memory_factor = h->opt_hybrid;
    h->ordering_requested = ordering;
    h->effective_ordering = (ordering == HIPMF_ORDERING_NONE || ordering == HIPMF_ORDERING_AMD || ordering == HIPMF_ORDERING_BEST) ? ordering : HIPMF_ORDERING_NESTED_DISSECTION; // (BEST: resolved once the analysis has chosen)
    int32_t code;
    // Symmetric INDEFINITE input (saddle-point / KKT matrices: put_lagrange_block, coo_matrix.rs:823-857): the L D L^T fronts never
    // interchange rows and the matching needs general storage, so a weak diagonal used to rest on perturbed pivots + refinement.  When
    // the values are handed over and the diagonal of the symmetric matrix is weak somewhere (missing, zero or < 1 % of the row's
    // largest entry: the criterion of the general path), the matrix is mirrored to general storage HERE and takes the general path:
    // maximum-product matching + scaling, LU with pivoting inside the pivot blocks.  Twice the factor of L D L^T, no perturbed pivots.
    // The caller's interface does not change: it keeps passing lower-triangle values (HIPMF_COUNTER_SYM_EXPANDED tells).
    bool expand = false;
    if (sym_lower && values && ndim > 1 && no.matching > 0 && row_pointers[0] == 0 && validate_csr(ndim, row_pointers, col_indices) == 0) {
        const char *e = getenv("HIPMF_SYM_EXPAND");
        if (!e || atoi(e) != 0) expand = sym_lower_diagonal_is_weak(ndim, row_pointers, col_indices, values);
    }
    h->so_keep = so;
    // (no values yet: with HIPMF_OPTION_SYM_RECHECK the first factorize looks at the diagonal)
    h->sym_unchecked = h->opt_sym_recheck == 1 && sym_lower && !values && no.matching > 0;
    if (expand) {
        code = initialize_expanded(h, ndim, row_pointers, col_indices, values, so, no);
    } else {
        code = h->solver.initialize(ndim, row_pointers, col_indices, sym_lower, so, no, values);
    }
    if (code == SUCCESSFUL_EXIT && !h->d_hdr) {
        DeviceGuard dg(h->solver.device);
        if (hipMalloc((void **)&h->d_hdr, sizeof(int64_t) * 8 + sizeof(int32_t) * 2) != hipSuccess) {
            h->d_hdr = nullptr;
            h->solver.release();
            h->expanded = false, h->nnz_lower = 0;
            return ERROR_HIP_MALLOC;
        }
    }
    if (verbose == 1 && code == SUCCESSFUL_EXIT) {
        const Symbolic &S = h->solver.S;
        printf("solver_hipmf_initialize: n=%d nnz=%lld supernodes=%d levels=%d nnz(L)=%lld nnz(U)=%lld flops=%.3e "
               "(ordering %.3fs, total %.3fs)\n",
               S.n, (long long)S.nnz_a, S.nsuper, S.nlevels, (long long)S.nnz_l, (long long)S.nnz_u, S.flops, S.seconds_ordering,
               S.seconds_total);
    }
    return code;
}

int32_t solver_hipmf_set_option(struct InterfaceHIPMF *h, int32_t option, double value) {
    if (!h) return ERROR_NULL_POINTER;
    if (h->solver.initialized) return ERROR_ALREADY_INITIALIZED;
    switch (option) {
    case HIPMF_OPTION_MATCHING:
        if (value < 0.0 || value > 2.0) return ERROR_HIPMF_INVALID_VALUE;
        h->opt_matching = (int32_t)value;
        return SUCCESSFUL_EXIT;
    case HIPMF_OPTION_PIVOTING:
        // enums.rs Pivoting as an integer: 0 Auto, 1 None, 2 GlobalCol, 3 GlobalRow, 4 Diagonal, 5 LocalBlock -- a REQUEST (cuDSS reports the
        // effective strategy back from factorize, interface_cudss.cu:485-491): recorded; the kernels' one strategy runs whatever is asked
        if (!(value >= 0.0 && value <= 5.0) || value != (double)(int32_t)value) return ERROR_HIPMF_INVALID_VALUE;
        h->opt_pivoting = (int32_t)value;
        return SUCCESSFUL_EXIT;
    case HIPMF_OPTION_HYBRID_MEMORY:
        if (!(value > 0.0 && value < 1.0)) return ERROR_HIPMF_INVALID_VALUE;
        h->opt_hybrid = value;
        return SUCCESSFUL_EXIT;
    case HIPMF_OPTION_SYM_RECHECK:
        if (value != 0.0 && value != 1.0) return ERROR_HIPMF_INVALID_VALUE;
        h->opt_sym_recheck = (int32_t)value;
        return SUCCESSFUL_EXIT;
    case HIPMF_OPTION_ERROR_ESTIMATES:
    case HIPMF_OPTION_CONDITION_NUMBERS: return SUCCESSFUL_EXIT; // (always computed)
    default: return ERROR_HIPMF_INVALID_VALUE;
    }
}

int32_t solver_hipmf_get_option(struct InterfaceHIPMF *h, int32_t option, double *value) {
    if (!h || !value) return ERROR_NULL_POINTER;
    switch (option) {
    case HIPMF_OPTION_MATCHING: *value = h->solver.initialized ? h->solver.opt.matching : h->opt_matching; return SUCCESSFUL_EXIT;
    case HIPMF_OPTION_PIVOTING: *value = h->opt_pivoting; return SUCCESSFUL_EXIT;
    case HIPMF_OPTION_EFFECTIVE_PIVOTING: *value = 5.0; return SUCCESSFUL_EXIT; // LocalBlock
    case HIPMF_OPTION_HYBRID_MEMORY: *value = h->opt_hybrid; return SUCCESSFUL_EXIT;
    case HIPMF_OPTION_SYM_RECHECK: *value = h->opt_sym_recheck; return SUCCESSFUL_EXIT;
    case HIPMF_OPTION_ERROR_ESTIMATES: *value = h->solver.last_omega; return SUCCESSFUL_EXIT;
    case HIPMF_OPTION_CONDITION_NUMBERS: {
        if (!h->solver.factorized) return ERROR_NEED_FACTORIZATION;
        return h->solver.rcond_estimate(value);
    }
    default: return ERROR_HIPMF_INVALID_VALUE;
    }
}

static int32_t finish_factorize(struct InterfaceHIPMF *h, int32_t code, int32_t *effective_ordering, int32_t *effective_scaling,
                                int32_t *num_perturbed, double *rcond, double *det_c, double *det_e, C_BOOL compute_determinant) {
    if (effective_ordering) {
        *effective_ordering = h->effective_ordering;
        if (h->effective_ordering == HIPMF_ORDERING_BEST) *effective_ordering = h->solver.S.best_chose_min_degree ? HIPMF_ORDERING_AMD : HIPMF_ORDERING_NESTED_DISSECTION;
    }
    if (effective_scaling) *effective_scaling = h->solver.opt.scaling;
    if (num_perturbed) *num_perturbed = h->solver.n_perturbed;
    if (rcond) *rcond = 0.0;
    if (det_c) *det_c = 0.0;
    if (det_e) *det_e = 0.0;
    if (code != SUCCESSFUL_EXIT && code != HIPMF_WARNING_SINGULAR_MATRIX) return code;
    if (compute_determinant == 1) {
        int32_t c2 = h->solver.determinant(det_c, det_e, rcond);
        if (c2 != SUCCESSFUL_EXIT) return c2;
    } else if (rcond) {
        // the reference reports Info[UMFPACK_RCOND] after every numeric phase (interface_umfpack.c:179-184): a device reduction here
        int32_t c2 = h->solver.rcond_estimate(rcond);
        if (c2 != SUCCESSFUL_EXIT) return c2;
    }
    return code;
}

// A symmetric-lower handle analysed WITHOUT values whose caller asked for it (HIPMF_OPTION_SYM_RECHECK = 1, set before initialize): the
// first values it is asked to factorise -- through solver_hipmf_factorize or solver_hipmf_factorize_device alike -- are the first numbers
// it sees.  A weak diagonal (saddle-point / KKT matrices) sends it where initialize would have sent it with values: mirrored to general
// storage, maximum-product matching, LU with pivoting inside the pivot blocks, at the price of one more analysis inside this call
// (HIPMF_COUNTER_SYM_EXPANDED tells).  Once per handle; a value map installed meanwhile speaks of the lower triangle's entries and is
// not carried over: such handles (and solver_hipmf_factorize_mapped) keep the L D L^T path.
// Opt-in since round 5: the re-analysis changes the plan of THIS handle only -- peers that are to receive its factor
// (solver_hipmf_broadcast_factor / adopt_factor) hold the L D L^T plan and would be refused, a permutation fetched earlier goes stale,
// and a full analysis lands in a timed factorize call.  A caller that has the values hands them to initialize (as the reference's
// shims do, interface_cudss.cu:190-203) and every rank then takes the same decision; this option is for single-handle callers that
// cannot.
static int32_t recheck_symmetric_diagonal(struct InterfaceHIPMF *h, const double *host_values, bool verbose) {
    if (!h->sym_unchecked) return SUCCESSFUL_EXIT;
    h->sym_unchecked = false;
    const char *e = getenv("HIPMF_SYM_EXPAND");
    const Solver &sv = h->solver;
    if ((!e || atoi(e) != 0) && sv.S.sym_lower && !h->expanded && sv.nnz_in_values() == 0 && sv.S.n > 1 &&
        sym_lower_diagonal_is_weak(sv.S.n, sv.kept_row_pointers().data(), sv.kept_col_indices().data(), host_values)) {
        const std::vector<int32_t> rp = sv.kept_row_pointers(), ci = sv.kept_col_indices();
        const NumericOptions no = sv.opt;
        const int32_t ndim = sv.S.n;
        h->solver.release();
        const int32_t c = initialize_expanded(h, ndim, rp.data(), ci.data(), host_values, h->so_keep, no);
        if (c != SUCCESSFUL_EXIT) return c;
        if (verbose) printf("solver_hipmf_factorize: weak diagonal of a symmetric matrix: analysed again as a general matrix with matching\n");
    }
    return SUCCESSFUL_EXIT;
}

static int32_t factorize_body(struct InterfaceHIPMF *h, int32_t *effective_ordering, int32_t *effective_scaling,
                               int32_t *num_perturbed_pivots, double *rcond_estimate, double *determinant_coefficient,
                               double *determinant_exponent, C_BOOL compute_determinant, C_BOOL verbose, const double *values) {
    if (!h || !values) return ERROR_NULL_POINTER;
    if (!h->solver.initialized) return ERROR_NEED_INITIALIZATION;
    h->solver.opt.verbose = verbose == 1;
    {
        const int32_t c = recheck_symmetric_diagonal(h, values, verbose == 1);
        if (c != SUCCESSFUL_EXIT) return c;
    }
    int32_t code = h->solver.factorize(values, false);
    if (verbose == 1 && h->solver.n_perturbed > 0)
        printf("solver_hipmf_factorize: WARNING: %d pivot(s) perturbed (matrix may be (nearly) singular)\n", h->solver.n_perturbed);
    return finish_factorize(h, code, effective_ordering, effective_scaling, num_perturbed_pivots, rcond_estimate,
                            determinant_coefficient, determinant_exponent, compute_determinant);
}

static int32_t set_value_map_body(struct InterfaceHIPMF *h, int32_t nnz_in, const int32_t *seg_ptr, const int32_t *seg_idx) {
    if (!h) return ERROR_NULL_POINTER;
    if (!h->expanded) return h->solver.set_value_map(nnz_in, seg_ptr, seg_idx);
    // the caller's map speaks of ITS CSR (the lower triangle): entry k of the handle's general CSR takes the segment of entry emap[k]
    if (!seg_ptr || !seg_idx) return ERROR_NULL_POINTER;
    if (!h->solver.initialized) return ERROR_NEED_INITIALIZATION;
    const int64_t nl = h->nnz_lower;
    if (seg_ptr[0] != 0) return ERROR_HIPMF_INVALID_VALUE;
    for (int64_t j = 0; j < nl; j++)
        if (seg_ptr[j + 1] < seg_ptr[j]) return ERROR_HIPMF_INVALID_VALUE;
    if (seg_ptr[nl] != nnz_in) return ERROR_HIPMF_INVALID_VALUE;
    const std::vector<int32_t> &emap = h->solver.expansion_map();
    std::vector<int32_t> sp(emap.size() + 1, 0), si;
    int64_t tot = 0;
    for (size_t k = 0; k < emap.size(); k++) tot += seg_ptr[emap[k] + 1] - seg_ptr[emap[k]];
    if (tot > 0x7fffffffLL) return ERROR_HIPMF_INVALID_VALUE;
    si.reserve((size_t)tot);
    for (size_t k = 0; k < emap.size(); k++) {
        for (int32_t q = seg_ptr[emap[k]]; q < seg_ptr[emap[k] + 1]; q++) {
            if (seg_idx[q] < 0 || seg_idx[q] >= nnz_in) return ERROR_HIPMF_INVALID_VALUE;
            si.push_back(seg_idx[q]);
        }
        sp[k + 1] = (int32_t)si.size();
    }
    // (signed-map form: seg_ptr[nnz] entries, every input used once per mirrored entry)
    return h->solver.set_value_map(nnz_in, sp.data(), si.data(), true);
}

static int32_t factorize_mapped_body(struct InterfaceHIPMF *h, int32_t *effective_ordering, int32_t *effective_scaling,
                                      int32_t *num_perturbed_pivots, double *rcond_estimate, double *determinant_coefficient,
                                      double *determinant_exponent, C_BOOL compute_determinant, C_BOOL verbose, const double *input_values) {
    if (!h || !input_values) return ERROR_NULL_POINTER;
    if (!h->solver.initialized) return ERROR_NEED_INITIALIZATION;
    h->solver.opt.verbose = verbose == 1;
    int32_t code = h->solver.factorize_mapped(input_values, false);
    return finish_factorize(h, code, effective_ordering, effective_scaling, num_perturbed_pivots, rcond_estimate, determinant_coefficient,
                            determinant_exponent, compute_determinant);
}

static int32_t factorize_mapped_device_body(struct InterfaceHIPMF *h, const double *d_input_values) {
    if (!h || !d_input_values) return ERROR_NULL_POINTER;
    if (!h->solver.initialized) return ERROR_NEED_INITIALIZATION;
    return h->solver.factorize_mapped(d_input_values, true);
}

static int32_t factorize_device_body(struct InterfaceHIPMF *h, const double *d_values) {
    if (!h || !d_values) return ERROR_NULL_POINTER;
    if (!h->solver.initialized) return ERROR_NEED_INITIALIZATION;
    if (h->sym_unchecked) { // (the same decision as solver_hipmf_factorize takes: the values come to the host once)
        std::vector<double> hv((size_t)h->solver.S.nnz_a);
        if (hipMemcpy(hv.data(), d_values, sizeof(double) * hv.size(), hipMemcpyDeviceToHost) != hipSuccess) return ERROR_HIP_MEMCPY;
        const int32_t c = recheck_symmetric_diagonal(h, hv.data(), false);
        if (c != SUCCESSFUL_EXIT) return c;
    }
    return h->solver.factorize(d_values, true);
}

static int32_t solve_body(struct InterfaceHIPMF *h, double *x, const double *rhs, C_BOOL verbose) {
    if (!h || !x || !rhs) return ERROR_NULL_POINTER;
    if (!h->solver.factorized) return ERROR_NEED_FACTORIZATION;
    h->solver.opt.verbose = verbose == 1;
    int32_t code = h->solver.solve(x, rhs, 1, h->solver.S.n, false);
    if (verbose == 1 && code == SUCCESSFUL_EXIT)
        printf("solver_hipmf_solve: Solution completed (%d refinement step(s), |r|_inf = %.3e)\n", h->solver.refinement_steps_done,
               h->solver.last_residual_inf);
    return code;
}

static int32_t solve_many_body(struct InterfaceHIPMF *h, double *x, const double *rhs, int32_t nrhs, int32_t ld, C_BOOL verbose) {
    if (!h || !x || !rhs) return ERROR_NULL_POINTER;
    if (!h->solver.factorized) return ERROR_NEED_FACTORIZATION;
    h->solver.opt.verbose = verbose == 1;
    return h->solver.solve(x, rhs, nrhs, ld, false);
}

static int32_t solve_device_body(struct InterfaceHIPMF *h, double *d_x, const double *d_rhs, int32_t nrhs, int32_t ld) {
    if (!h || !d_x || !d_rhs) return ERROR_NULL_POINTER;
    if (!h->solver.factorized) return ERROR_NEED_FACTORIZATION;
    return h->solver.solve(d_x, d_rhs, nrhs, ld, true);
}


int32_t solver_hipmf_initialize(struct InterfaceHIPMF *h, int32_t ordering, int32_t scaling, double pivot_epsilon,
                                int32_t refinement_nstep, C_BOOL verbose, C_BOOL general_symmetric, C_BOOL positive_definite,
                                int32_t ndim, const int32_t *row_pointers, const int32_t *col_indices, const double *values) {
    return guarded(h, [&]() { return initialize_body(h, ordering, scaling, pivot_epsilon, refinement_nstep, verbose, general_symmetric, positive_definite, ndim, row_pointers, col_indices, values); });
}

int32_t solver_hipmf_factorize(struct InterfaceHIPMF *h, int32_t *effective_ordering, int32_t *effective_scaling,
                               int32_t *num_perturbed_pivots, double *rcond_estimate, double *determinant_coefficient,
                               double *determinant_exponent, C_BOOL compute_determinant, C_BOOL verbose, const double *values) {
    return guarded(h, [&]() { return factorize_body(h, effective_ordering, effective_scaling, num_perturbed_pivots, rcond_estimate, determinant_coefficient, determinant_exponent, compute_determinant, verbose, values); });
}

int32_t solver_hipmf_set_value_map(struct InterfaceHIPMF *h, int32_t nnz_in, const int32_t *seg_ptr, const int32_t *seg_idx) {
    return guarded(h, [&]() { return set_value_map_body(h, nnz_in, seg_ptr, seg_idx); });
}

int32_t solver_hipmf_factorize_mapped(struct InterfaceHIPMF *h, int32_t *effective_ordering, int32_t *effective_scaling,
                                      int32_t *num_perturbed_pivots, double *rcond_estimate, double *determinant_coefficient,
                                      double *determinant_exponent, C_BOOL compute_determinant, C_BOOL verbose, const double *input_values) {
    return guarded(h, [&]() { return factorize_mapped_body(h, effective_ordering, effective_scaling, num_perturbed_pivots, rcond_estimate, determinant_coefficient, determinant_exponent, compute_determinant, verbose, input_values); });
}

int32_t solver_hipmf_factorize_mapped_device(struct InterfaceHIPMF *h, const double *d_input_values) {
    return guarded(h, [&]() { return factorize_mapped_device_body(h, d_input_values); });
}

int32_t solver_hipmf_factorize_device(struct InterfaceHIPMF *h, const double *d_values) {
    return guarded(h, [&]() { return factorize_device_body(h, d_values); });
}

int32_t solver_hipmf_solve(struct InterfaceHIPMF *h, double *x, const double *rhs, C_BOOL verbose) {
    return guarded(h, [&]() { return solve_body(h, x, rhs, verbose); });
}

int32_t solver_hipmf_solve_many(struct InterfaceHIPMF *h, double *x, const double *rhs, int32_t nrhs, int32_t ld, C_BOOL verbose) {
    return guarded(h, [&]() { return solve_many_body(h, x, rhs, nrhs, ld, verbose); });
}

int32_t solver_hipmf_solve_device(struct InterfaceHIPMF *h, double *d_x, const double *d_rhs, int32_t nrhs, int32_t ld) {
    return guarded(h, [&]() { return solve_device_body(h, d_x, d_rhs, nrhs, ld); });
}

int32_t solver_hipmf_mat_vec_mul(struct InterfaceHIPMF *h, double *v, double alpha, const double *u) {
    if (!h || !v || !u) return ERROR_NULL_POINTER;
    return guarded(h, [&]() { return h->solver.spmv(v, u, alpha, false); });
}

int32_t hipmf_max_product_matching(int32_t ndim, const int32_t *row_pointers, const int32_t *col_indices, const double *values,
                                   int32_t *matched_row, double *row_scale, double *col_scale) {
    if (!row_pointers || !col_indices || !values || !matched_row || !row_scale || !col_scale) return ERROR_NULL_POINTER;
    if (ndim < 1 || validate_csr(ndim, row_pointers, col_indices) != 0) return ERROR_HIPMF_INVALID_MATRIX;
    std::vector<int32_t> mrow;
    std::vector<double> dr, dc;
    if (max_product_matching(ndim, row_pointers, col_indices, values, mrow, dr, dc) != 0) return ERROR_HIPMF_INVALID_MATRIX;
    for (int32_t i = 0; i < ndim; i++) matched_row[i] = mrow[i], row_scale[i] = dr[i], col_scale[i] = dc[i];
    return SUCCESSFUL_EXIT;
}

int32_t hipmf_paired_matching(int32_t ndim2, const int32_t *row_pointers, const int32_t *col_indices, const double *values, int32_t *matched_row,
                              double *row_scale, double *col_scale) {
    if (!row_pointers || !col_indices || !values || !matched_row || !row_scale || !col_scale) return ERROR_NULL_POINTER;
    if (ndim2 < 2 || ndim2 % 2 != 0 || validate_csr(ndim2, row_pointers, col_indices) != 0) return ERROR_HIPMF_INVALID_MATRIX;
    return guarded(nullptr, [&]() {
        std::vector<int32_t> mrow;
        std::vector<double> dr, dc;
        if (paired_matching(ndim2, row_pointers, col_indices, values, mrow, dr, dc) != 0) return (int32_t)ERROR_HIPMF_INVALID_MATRIX;
        for (int32_t i = 0; i < ndim2; i++) matched_row[i] = mrow[i], row_scale[i] = dr[i], col_scale[i] = dc[i];
        return (int32_t)SUCCESSFUL_EXIT;
    });
}

int32_t solver_hipmf_get_permutation(struct InterfaceHIPMF *h, int32_t *perm) {
    if (!h || !perm) return ERROR_NULL_POINTER;
    if (!h->solver.initialized) return ERROR_NEED_INITIALIZATION;
    for (int32_t i = 0; i < h->solver.S.n; i++) perm[i] = h->solver.S.perm[i];
    return SUCCESSFUL_EXIT;
}

int32_t solver_hipmf_get_stats(struct InterfaceHIPMF *h, int64_t *is, double *ds) {
    if (!h || !is || !ds) return ERROR_NULL_POINTER;
    if (!h->solver.initialized) return ERROR_NEED_INITIALIZATION;
    const Solver &s = h->solver;
    for (int i = 0; i < 16; i++) is[i] = 0, ds[i] = 0.0;
    is[0] = s.S.n, is[1] = s.S.nnz_a, is[2] = s.S.nsuper, is[3] = s.S.nlevels, is[4] = s.S.nnz_l, is[5] = s.S.nnz_u;
    is[6] = s.S.max_front, is[7] = s.S.max_pivots, is[8] = s.n_perturbed, is[9] = s.n_zero_pivot;
    is[10] = s.refinement_steps_done, is[11] = s.times.n_kernel_launches_factor, is[12] = s.times.n_kernel_launches_solve;
    is[13] = s.pool_doubles * 8;
    is[14] = s.matched ? 1 : 0;
    is[15] = s.fused_fallbacks;
    ds[0] = s.S.flops, ds[1] = s.S.flops_gemm, ds[2] = s.S.seconds_ordering, ds[3] = s.S.seconds_total;
    ds[4] = s.times.scale_assemble_ms, ds[5] = s.times.factor_ms, ds[6] = s.times.fwd_ms, ds[7] = s.times.bwd_ms;
    ds[8] = s.times.solve_total_ms, ds[9] = s.last_residual_inf;
    ds[10] = s.times.acc_assemble_ms, ds[11] = s.times.acc_factor_ms, ds[12] = (double)s.times.acc_factor_count;
    ds[13] = s.times.acc_fwd_ms, ds[14] = s.times.acc_bwd_ms, ds[15] = (double)s.times.acc_tri_count;
    return SUCCESSFUL_EXIT;
}

int64_t solver_hipmf_get_counter(struct InterfaceHIPMF *h, int32_t which) {
    if (!h || !h->solver.initialized) return -1;
    const Solver &s = h->solver;
    switch (which) {
    case HIPMF_COUNTER_REMATCH: return s.rematch_count;
    case HIPMF_COUNTER_WEAK_DIAGONAL_ROWS: return s.n_weak_diag;
    case HIPMF_COUNTER_FUSED_FALLBACKS: return s.fused_fallbacks;
    case HIPMF_COUNTER_PERSISTENT_BYTES: return s.S.persist_doubles * 8;
    case HIPMF_COUNTER_ARENA_BYTES: return s.S.temp_doubles * 8;
    case HIPMF_COUNTER_SYMMETRIC_LDLT: return s.S.sym_mode ? 1 : 0;
    case HIPMF_COUNTER_SYM_EXPANDED: return h->expanded ? 1 : 0;
    case HIPMF_COUNTER_CHAIN_FALLBACKS: return s.chain_fallbacks;
    case HIPMF_COUNTER_MID_FRONTS: return s.mid_front_count;
    case HIPMF_COUNTER_PLAN_DIGEST: return s.plan_digest;
    case HIPMF_COUNTER_TAGGED_SOLVE: return s.tagged_solve() ? 1 : 0;
    case HIPMF_COUNTER_GATE_WAITS: return s.gate_waits;
    case HIPMF_COUNTER_WAVE_FRONTS: return s.wave_front_count;
    case HIPMF_COUNTER_LEAF_FRONTS: return s.leaf_front_count();
    case HIPMF_COUNTER_SPLIT_SLABS: return s.split_slab_count();
    case HIPMF_COUNTER_EVENT_FENCE_FREE: return s.event_fence_free ? 1 : 0;
    case HIPMF_COUNTER_BLOCK_GROUPS: return s.block_groups_last;
    case HIPMF_COUNTER_SYM_WEAK_DIAGONAL: return s.sym_weak_diag_seen ? 1 : 0;
    case HIPMF_COUNTER_BCAST_SLICED_BYTES: return h->bcast_sliced_bytes;
    case HIPMF_COUNTER_KRYLOV_ITERATIONS: return s.krylov_iterations;
    default: return -1;
    }
}

int32_t solver_hipmf_factor_parts(struct InterfaceHIPMF *h, int32_t max_parts, void **d_ptrs, int64_t *bytes) {
    if (!h || !d_ptrs || !bytes) return ERROR_NULL_POINTER;
    if (!h->solver.initialized) return ERROR_NEED_INITIALIZATION;
    const Solver &s = h->solver;
    void *p[4] = {s.d_pool, s.d_lperm, s.d_rs, s.d_diag_ptr()};
    const int64_t nb[4] = {s.S.persist_doubles * 8, (int64_t)s.S.n * 4, (int64_t)s.S.n * 8, (int64_t)s.S.n * 8};
    if (max_parts < 4) return ERROR_HIPMF_INVALID_VALUE;
    for (int i = 0; i < 4; i++) d_ptrs[i] = p[i], bytes[i] = nb[i];
    return 4;
}

int32_t solver_hipmf_adopt_factor(struct InterfaceHIPMF *h, const double *d_values) {
    // the peer wrote the four parts of the factor (persistent part of the pool, interchanges, row scaling, pivots) through the
    // pointers of solver_hipmf_factor_parts -- with a matching in force also the column scaling; the values are still needed for
    // the refinement SpMV.  A re-matching factorize on the sending side (HIPMF_COUNTER_REMATCH) invalidates those pointers.
    if (!h || !d_values) return ERROR_NULL_POINTER;
    if (!h->solver.initialized) return ERROR_NEED_INITIALIZATION;
    return guarded(h, [&]() { return h->solver.adopt_factor(d_values); });
}

int32_t solver_hipmf_reset_timers(struct InterfaceHIPMF *h) {
    if (!h) return ERROR_NULL_POINTER;
    PhaseTimes &t = h->solver.times;
    t.acc_assemble_ms = t.acc_factor_ms = t.acc_fwd_ms = t.acc_bwd_ms = 0.0;
    t.acc_factor_count = t.acc_tri_count = 0;
    return SUCCESSFUL_EXIT;
}

// ---- many-RHS over the GPUs of a node: the factor goes from the rank that factorised to the others over RCCL (xGMI) ----
// RCCL is bound at the first use (dlopen): a single-GPU caller never loads it.
#ifdef HIPMF_HAVE_RCCL
namespace {
struct Rccl {
    void *dl = nullptr;
    decltype(&ncclGetUniqueId) get_unique_id = nullptr;
    decltype(&ncclCommInitRank) comm_init_rank = nullptr;
    decltype(&ncclCommDestroy) comm_destroy = nullptr;
    decltype(&ncclBroadcast) broadcast = nullptr;
    decltype(&ncclAllReduce) all_reduce = nullptr;
    // (round 6: the factor travels as slices over all links -- point-to-point sends in groups; optional: without them the broadcast stays)
    decltype(&ncclSend) send = nullptr;
    decltype(&ncclRecv) recv = nullptr;
    decltype(&ncclGroupStart) group_start = nullptr;
    decltype(&ncclGroupEnd) group_end = nullptr;
    decltype(&ncclCommCount) comm_count = nullptr;
    bool tried = false, ok = false;
    bool p2p() const { return send && recv && group_start && group_end && comm_count; }
};
Rccl g_rccl;
std::mutex g_rccl_mutex;
bool rccl_load() {
    std::lock_guard<std::mutex> lock(g_rccl_mutex);
    if (g_rccl.tried) return g_rccl.ok;
    g_rccl.tried = true;
    for (const char *name : {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"}) {
        g_rccl.dl = dlopen(name, RTLD_NOW | RTLD_GLOBAL);
        if (g_rccl.dl) break;
    }
    if (!g_rccl.dl) return false;
    g_rccl.get_unique_id = (decltype(g_rccl.get_unique_id))dlsym(g_rccl.dl, "ncclGetUniqueId");
    g_rccl.comm_init_rank = (decltype(g_rccl.comm_init_rank))dlsym(g_rccl.dl, "ncclCommInitRank");
    g_rccl.comm_destroy = (decltype(g_rccl.comm_destroy))dlsym(g_rccl.dl, "ncclCommDestroy");
    g_rccl.broadcast = (decltype(g_rccl.broadcast))dlsym(g_rccl.dl, "ncclBroadcast");
    g_rccl.all_reduce = (decltype(g_rccl.all_reduce))dlsym(g_rccl.dl, "ncclAllReduce");
    g_rccl.send = (decltype(g_rccl.send))dlsym(g_rccl.dl, "ncclSend");
    g_rccl.recv = (decltype(g_rccl.recv))dlsym(g_rccl.dl, "ncclRecv");
    g_rccl.group_start = (decltype(g_rccl.group_start))dlsym(g_rccl.dl, "ncclGroupStart");
    g_rccl.group_end = (decltype(g_rccl.group_end))dlsym(g_rccl.dl, "ncclGroupEnd");
    g_rccl.comm_count = (decltype(g_rccl.comm_count))dlsym(g_rccl.dl, "ncclCommCount");
    g_rccl.ok = g_rccl.get_unique_id && g_rccl.comm_init_rank && g_rccl.comm_destroy && g_rccl.broadcast && g_rccl.all_reduce;
    return g_rccl.ok;
}
} // namespace

int32_t hipmf_comm_unique_id(void *id128) {
    if (!id128) return ERROR_NULL_POINTER;
    if (!rccl_load()) return ERROR_NOT_AVAILABLE;
    static_assert(sizeof(ncclUniqueId) == HIPMF_COMM_ID_BYTES, "ncclUniqueId size");
    return g_rccl.get_unique_id((ncclUniqueId *)id128) == ncclSuccess ? SUCCESSFUL_EXIT : ERROR_HIPMF_COMM;
}

int32_t hipmf_comm_init_rank(void **comm, int32_t nranks, const void *id128, int32_t rank) {
    if (!comm || !id128) return ERROR_NULL_POINTER;
    if (nranks < 1 || rank < 0 || rank >= nranks) return ERROR_HIPMF_INVALID_VALUE;
    if (!rccl_load()) return ERROR_NOT_AVAILABLE;
    ncclUniqueId id;
    memcpy(&id, id128, sizeof id);
    ncclComm_t c = nullptr;
    if (g_rccl.comm_init_rank(&c, nranks, id, rank) != ncclSuccess) return ERROR_HIPMF_COMM;
    *comm = (void *)c;
    return SUCCESSFUL_EXIT;
}

void hipmf_comm_destroy(void *comm) {
    if (comm && rccl_load()) (void)g_rccl.comm_destroy((ncclComm_t)comm);
}

static int32_t broadcast_factor_body(struct InterfaceHIPMF *h, void *comm, int32_t root, int32_t rank, double *seconds, int64_t *bytes_sent) {
    // Errors of the CALL (every rank makes the same mistake) return at once; a condition that only ONE rank can be in -- the root without
    // a factorisation, a plan that differs from the root's -- is carried into the collectives below, so that every rank takes part in
    // every collective and all ranks return an error together instead of leaving their peers inside ncclBroadcast.
    if (!h || !comm) return ERROR_NULL_POINTER;
    if (!h->solver.initialized || !h->d_hdr) return ERROR_NEED_INITIALIZATION;
    if (!rccl_load()) return ERROR_NOT_AVAILABLE;
    Solver &s = h->solver;
    DeviceGuard dg(s.device);
    if (!dg.ok) return ERROR_HIPMF_NO_DEVICE;
    const hipStream_t st = (hipStream_t)s.stream;
    const bool root_unfactorized = rank == root && !s.factorized;
    // 1. every rank must hold the SAME plan as the root (same structure, same ordering, same matching): the root's plan signature
    //    travels first, every rank compares it with its own, and the ranks agree on the outcome (MIN over the ranks) before any
    //    factor data moves -- mismatched buffer sizes would otherwise hang the collective or, worse, be adopted into another plan.
    //    A root that re-matched inside factorize (rematch_count) has another plan than peers that did not: they must be initialised
    //    again from the root's values.
    const auto t0 = std::chrono::steady_clock::now();
    {
        int64_t mine[8] = {s.S.n, s.S.nnz_a, s.persist_bytes(), s.matched ? 1 : 0, (int64_t)s.plan_signature(), s.S.nsuper, s.S.sym_mode ? 1 : 0, 0};
        int64_t *d_hdr = h->d_hdr;
        int32_t *d_ok = (int32_t *)(d_hdr + 8);
        int64_t got[8] = {0, 0, 0, 0, 0, 0, 0, 0};
        // (a local copy that fails lowers this rank's status word; the collectives themselves are always entered)
        bool local_fail = hipMemcpyAsync(d_hdr, mine, sizeof(mine), hipMemcpyHostToDevice, st) != hipSuccess;
        const bool c1 = g_rccl.broadcast(d_hdr, d_hdr, sizeof(mine), ncclChar, root, (ncclComm_t)comm, st) == ncclSuccess;
        local_fail = local_fail || hipMemcpyAsync(got, d_hdr, sizeof(got), hipMemcpyDeviceToHost, st) != hipSuccess;
        local_fail = local_fail || hipStreamSynchronize(st) != hipSuccess;
        int32_t ok = (c1 && !local_fail && !root_unfactorized && memcmp(got, mine, sizeof(mine)) == 0) ? 1 : 0, all_ok = 0;
        const bool up = hipMemcpyAsync(d_ok, &ok, sizeof(ok), hipMemcpyHostToDevice, st) == hipSuccess;
        if (!up) (void)hipMemsetAsync(d_ok, 0, sizeof(ok), st); // (status 0 = "not ok": the MIN carries it to everybody)
        const bool c2 = g_rccl.all_reduce(d_ok, d_ok + 1, 1, ncclInt32, ncclMin, (ncclComm_t)comm, st) == ncclSuccess;
        bool fail = !c1 || !c2 || local_fail || !up;
        fail = hipMemcpyAsync(&all_ok, d_ok + 1, sizeof(all_ok), hipMemcpyDeviceToHost, st) != hipSuccess || fail;
        fail = hipStreamSynchronize(st) != hipSuccess || fail;
        if (root_unfactorized) return ERROR_NEED_FACTORIZATION;
        if (fail) return ERROR_HIPMF_COMM;
        if (all_ok != 1) {
            s.last_error = ok == 1 ? "broadcast_factor: another rank is not ready (the root has no factorisation, or a rank holds a different plan than the root)"
                                   : "broadcast_factor: this rank's plan differs from the root's (structure, ordering or matching); initialize it "
                                     "from the same structure and values as the root";
            return ERROR_HIPMF_INVALID_VALUE;
        }
    }
    // 2. the factor: persistent part of the pool, interchanges, row scaling, pivots; the column scaling of a matching; the matrix
    //    values (the refinement SpMV of every rank needs them)
    void *ptrs[6];
    int64_t nb[6];
    if (solver_hipmf_factor_parts(h, 4, ptrs, nb) != 4) return ERROR_HIPMF_INVALID_VALUE;
    ptrs[4] = s.d_vals_ptr(), nb[4] = s.S.nnz_a * 8;
    ptrs[5] = (s.matched && s.d_cs) ? (void *)s.d_cs : nullptr, nb[5] = ptrs[5] ? (int64_t)s.S.n * 8 : 0;
    // xGMI is point-to-point: every GPU has ONE link to each peer (~153 GB/s).  A broadcast that leaves the root over one link -- a ring --
    // is bound by that link: F / 153 GB/s (config 4's 84 GB: 0.55 s, more than the sharded solves it feeds).  Round 6: a part of at least
    // HIPMF_BCAST_SLICE_MIN bytes travels as N slices over ALL links, in two steps of point-to-point transfers (N ranks):
    //   A. the root sends slice r to rank r (N - 1 links of the root busy at once);
    //   B. every rank sends ITS slice to every other rank but the root (the root: its own slice to everybody) -- all links busy;
    // per link and step F / N bytes: 2 F / (N 153 GB/s), a quarter of the ring's time at N = 8.  The ranks hold the same plan (checked
    // above), so every rank computes the same slices; the bytes past the last whole slice and the small parts use ncclBroadcast.
    // HIPMF_BCAST_SLICES=0 keeps the plain broadcast.
    int nranks = 1;
    const bool want_slices = !(getenv("HIPMF_BCAST_SLICES") && atoi(getenv("HIPMF_BCAST_SLICES")) == 0);
    if (g_rccl.p2p() && g_rccl.comm_count((ncclComm_t)comm, &nranks) != ncclSuccess) nranks = 1;
    int64_t slice_min = 64ll << 20;
    if (const char *e = getenv("HIPMF_BCAST_SLICE_MIN")) slice_min = std::max<int64_t>(4096, atoll(e));
    const int64_t chunk = 256ll << 20; // (plain broadcast: large messages keep the links of a ring busy)
    int64_t total = 0;
    for (int i = 0; i < 6; i++) {
        int64_t done = 0;
        if (want_slices && g_rccl.p2p() && nranks >= 3 && nb[i] >= slice_min) {
            const int64_t sl = (nb[i] / nranks) & ~(int64_t)511; // whole 512-byte units per slice
            char *base = (char *)ptrs[i];
            if (sl > 0) {
                // A: root -> r
                if (g_rccl.group_start() != ncclSuccess) return ERROR_HIPMF_COMM;
                bool bad = false;
                // (a slice travels in pieces of at most 1 GiB: every piece one send / recv pair of the group, the same pieces on both ends)
                const int64_t piece = 1ll << 30;
                auto send_slice = [&](int64_t slice, int peer) {
                    for (int64_t o = 0; o < sl; o += piece)
                        bad = bad || g_rccl.send(base + slice * sl + o, (size_t)std::min(piece, sl - o), ncclChar, peer, (ncclComm_t)comm, st) != ncclSuccess;
                };
                auto recv_slice = [&](int64_t slice, int peer) {
                    for (int64_t o = 0; o < sl; o += piece)
                        bad = bad || g_rccl.recv(base + slice * sl + o, (size_t)std::min(piece, sl - o), ncclChar, peer, (ncclComm_t)comm, st) != ncclSuccess;
                };
                if (rank == root) {
                    for (int r = 0; r < nranks; r++)
                        if (r != root) send_slice(r, r);
                } else
                    recv_slice(rank, root);
                if (g_rccl.group_end() != ncclSuccess || bad) return ERROR_HIPMF_COMM;
                // B: r -> q for every pair r != q, q != root (slice r)
                if (g_rccl.group_start() != ncclSuccess) return ERROR_HIPMF_COMM;
                for (int q = 0; q < nranks; q++) {
                    if (q == rank) continue;
                    if (q != root) send_slice(rank, q);
                    if (rank != root) recv_slice(q, q);
                }
                if (g_rccl.group_end() != ncclSuccess || bad) return ERROR_HIPMF_COMM;
                done = sl * nranks;
                h->bcast_sliced_bytes += done;
            }
        }
        for (int64_t off = done; off < nb[i]; off += chunk) {
            const int64_t len = std::min(chunk, nb[i] - off);
            char *p = (char *)ptrs[i] + off;
            if (g_rccl.broadcast(p, p, (size_t)len, ncclChar, root, (ncclComm_t)comm, st) != ncclSuccess) return ERROR_HIPMF_COMM;
        }
        total += nb[i];
    }
    // 3. what the root's factorisation knows about itself and the solves of every rank need: how many pivots were replaced (round 6: a rank
    //    that adopts a factor with replaced pivots runs the Krylov rescue like the root, numeric.cpp)
    int64_t tail[2] = {s.n_perturbed, 0}, got_tail[2] = {0, 0};
    bool tail_fail = hipMemcpyAsync(h->d_hdr, tail, sizeof(tail), hipMemcpyHostToDevice, st) != hipSuccess;
    tail_fail = g_rccl.broadcast(h->d_hdr, h->d_hdr, sizeof(tail), ncclChar, root, (ncclComm_t)comm, st) != ncclSuccess || tail_fail;
    tail_fail = hipMemcpyAsync(got_tail, h->d_hdr, sizeof(got_tail), hipMemcpyDeviceToHost, st) != hipSuccess || tail_fail;
    if (hipStreamSynchronize(st) != hipSuccess) return ERROR_HIP_SYNCHRONIZE;
    if (tail_fail) return ERROR_HIPMF_COMM;
    if (seconds) *seconds = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    if (bytes_sent) *bytes_sent = total;
    if (rank != root) s.mark_factor_adopted((int32_t)std::min<int64_t>(got_tail[0], 0x7fffffff));
    return SUCCESSFUL_EXIT;
}

int32_t solver_hipmf_broadcast_factor(struct InterfaceHIPMF *h, void *comm, int32_t root, int32_t rank, double *seconds, int64_t *bytes_sent) {
    return guarded(h, [&]() { return broadcast_factor_body(h, comm, root, rank, seconds, bytes_sent); });
}
#else
int32_t hipmf_comm_unique_id(void *) { return ERROR_NOT_AVAILABLE; }
int32_t hipmf_comm_init_rank(void **, int32_t, const void *, int32_t) { return ERROR_NOT_AVAILABLE; }
void hipmf_comm_destroy(void *) {}
int32_t solver_hipmf_broadcast_factor(struct InterfaceHIPMF *, void *, int32_t, int32_t, double *, int64_t *) { return ERROR_NOT_AVAILABLE; }
#endif

int32_t solver_hipmf_solve_many_sharded(struct InterfaceHIPMF *h, double *d_x, const double *d_rhs, int32_t nrhs_total, int32_t ld, int32_t nranks,
                                        int32_t rank, int32_t *first_column, int32_t *num_columns) {
    // columns [first, first + count) of B belong to this rank (contiguous blocks whose sizes differ by at most one); they are solved
    // in place of the caller's n x nrhs_total arrays: d_rhs / d_x point at column 0 of the WHOLE arrays resident on this rank's GPU
    if (!h || !d_x || !d_rhs) return ERROR_NULL_POINTER;
    if (nranks < 1 || rank < 0 || rank >= nranks || nrhs_total < 0) return ERROR_HIPMF_INVALID_VALUE;
    if (!h->solver.factorized) return ERROR_NEED_FACTORIZATION;
    const int32_t base = nrhs_total / nranks, extra = nrhs_total % nranks;
    const int32_t count = base + (rank < extra ? 1 : 0), first = rank * base + std::min(rank, extra);
    if (first_column) *first_column = first;
    if (num_columns) *num_columns = count;
    if (count == 0) return SUCCESSFUL_EXIT;
    return h->solver.solve(d_x + (int64_t)first * ld, d_rhs + (int64_t)first * ld, count, ld, true);
}

int32_t solver_hipmf_prepare_solve_many(struct InterfaceHIPMF *h, int32_t nrhs) {
    if (!h) return ERROR_NULL_POINTER;
    if (!h->solver.initialized) return ERROR_NEED_INITIALIZATION;
    if (nrhs < 0) return ERROR_HIPMF_INVALID_VALUE;
    return guarded(h, [&]() { return h->solver.prepare_many(nrhs); });
}

const char *solver_hipmf_last_error(struct InterfaceHIPMF *h) { return h ? h->solver.last_error.c_str() : "null solver"; }

void *hipmf_device_malloc(size_t bytes) {
    void *p = nullptr;
    if (hipMalloc(&p, bytes ? bytes : 1) != hipSuccess) return nullptr;
    return p;
}
void hipmf_device_free(void *ptr) {
    if (ptr) (void)hipFree(ptr);
}
int32_t hipmf_memcpy_h2d(void *dst, const void *src, size_t bytes) {
    return hipMemcpy(dst, src, bytes, hipMemcpyHostToDevice) == hipSuccess ? SUCCESSFUL_EXIT : ERROR_HIP_MEMCPY;
}
int32_t hipmf_memcpy_d2h(void *dst, const void *src, size_t bytes) {
    return hipMemcpy(dst, src, bytes, hipMemcpyDeviceToHost) == hipSuccess ? SUCCESSFUL_EXIT : ERROR_HIP_MEMCPY;
}
int32_t hipmf_device_synchronize(void) { return hipDeviceSynchronize() == hipSuccess ? SUCCESSFUL_EXIT : ERROR_HIP_SYNCHRONIZE; }
int32_t hipmf_device_mem_info(size_t *free_bytes, size_t *total_bytes) {
    if (!free_bytes || !total_bytes) return ERROR_NULL_POINTER;
    return hipMemGetInfo(free_bytes, total_bytes) == hipSuccess ? SUCCESSFUL_EXIT : ERROR_HIPMF_NO_DEVICE;
}
int32_t hipmf_set_device(int32_t device) { return hipSetDevice(device) == hipSuccess ? SUCCESSFUL_EXIT : ERROR_HIPMF_NO_DEVICE; }
int32_t hipmf_device_copy_bandwidth(int64_t bytes, int32_t reps, double *gb_per_s) {
    if (!gb_per_s || bytes < 1 || reps < 1) return ERROR_NULL_POINTER;
    void *a = nullptr, *b = nullptr;
    hipEvent_t e0, e1;
    if (hipMalloc(&a, (size_t)bytes) != hipSuccess) return ERROR_HIP_MALLOC;
    if (hipMalloc(&b, (size_t)bytes) != hipSuccess) {
        (void)hipFree(a);
        return ERROR_HIP_MALLOC;
    }
    (void)hipMemset(a, 1, (size_t)bytes);
    (void)hipEventCreate(&e0);
    (void)hipEventCreate(&e1);
    double best = 0.0;
    for (int32_t r = 0; r <= reps; r++) { // first copy = warm-up
        (void)hipEventRecord(e0, nullptr);
        (void)hipMemcpyAsync(b, a, (size_t)bytes, hipMemcpyDeviceToDevice, nullptr);
        (void)hipEventRecord(e1, nullptr);
        (void)hipEventSynchronize(e1);
        float ms = 0.0f;
        (void)hipEventElapsedTime(&ms, e0, e1);
        if (r > 0 && ms > 0.0f) best = std::max(best, 2.0 * (double)bytes / (ms * 1e-3) / 1e9);
    }
    (void)hipEventDestroy(e0);
    (void)hipEventDestroy(e1);
    (void)hipFree(a);
    (void)hipFree(b);
    *gb_per_s = best;
    return SUCCESSFUL_EXIT;
}

namespace {
// back-to-back v_mfma_f64_16x16x4_f64 on four independent accumulators per wave, operands in registers
__global__ void __launch_bounds__(256) k_mfma_probe(double *out, int32_t iters, double a0, double b0) {
    f64x4 acc[4];
    for (int i = 0; i < 4; i++) acc[i] = f64x4{0.0, 0.0, 0.0, 0.0};
    const double a = a0 + threadIdx.x * 1e-9, b = b0 - threadIdx.x * 1e-9;
    for (int it = 0; it < iters; it++) {
#pragma unroll
        for (int i = 0; i < 4; i++) acc[i] = mfma_f64_16x16x4(a, b, acc[i]);
    }
    double sum = 0.0;
    for (int i = 0; i < 4; i++) sum += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
    out[blockIdx.x * 256 + threadIdx.x] = sum;
}
} // namespace

int32_t hipmf_device_mfma_rate(int32_t workgroups, int32_t iters, double *tflops) {
    if (!tflops || workgroups < 1 || iters < 1) return ERROR_NULL_POINTER;
    double *d = nullptr;
    if (hipMalloc((void **)&d, sizeof(double) * 256 * (size_t)workgroups) != hipSuccess) return ERROR_HIP_MALLOC;
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0);
    (void)hipEventCreate(&e1);
    double best = 0.0;
    for (int r = 0; r < 3; r++) { // first launch = warm-up
        (void)hipEventRecord(e0, nullptr);
        hipLaunchKernelGGL(k_mfma_probe, dim3(workgroups), dim3(256), 0, nullptr, d, iters, 1.0, 1.0);
        (void)hipEventRecord(e1, nullptr);
        (void)hipEventSynchronize(e1);
        float ms = 0.0f;
        (void)hipEventElapsedTime(&ms, e0, e1);
        // 4 waves x 4 accumulators x 2048 flops per MFMA and iteration
        if (r > 0 && ms > 0.0f) best = std::max(best, (double)workgroups * 4.0 * 4.0 * 2048.0 * (double)iters / (ms * 1e-3) / 1e12);
    }
    (void)hipEventDestroy(e0);
    (void)hipEventDestroy(e1);
    (void)hipFree(d);
    *tflops = best;
    return hipGetLastError() == hipSuccess ? SUCCESSFUL_EXIT : ERROR_HIP_LAUNCH;
}

int32_t hipmf_device_count(void) {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    return n;
}

} // extern "C"
