/*
 * russell_hipmf.h -- C-ABI of the MI355X-native sparse direct solver backend ("HIPMF": HIP
 * multifrontal LU) that plugs in behind russell_sparse's solver boundary.
 *
 * The entry points are what russell_sparse's FFI for this path binds.  They keep the shape of the
 * reference's existing GPU plug-in (c_code/interface_cudss.cu) and of the UMFPACK shim
 * (c_code/interface_umfpack.c): an opaque handle, new/drop, and the three phases
 * initialize (once) / factorize (values only, repeatable) / solve.  All sizes, indices, enums and
 * booleans are int32_t; matrices are 0-based CSR with f64 values; host pointers are borrowed for
 * the duration of a call and never retained (ownership rules of SURVEY.md section 8b).
 *
 * Reference interface each declaration replaces (paths relative to /root/reference):
 *   solver_hipmf_new         russell_sparse/c_code/interface_cudss.cu:62-123   (solver_cudss_new)
 *   solver_hipmf_drop        russell_sparse/c_code/interface_cudss.cu:126-171  (solver_cudss_drop)
 *   solver_hipmf_initialize  russell_sparse/c_code/interface_cudss.cu:190-396  (solver_cudss_initialize)
 *                            and interface_umfpack.c:82-124 (ordering, scaling arguments)
 *   solver_hipmf_factorize   russell_sparse/c_code/interface_cudss.cu:406-501  (solver_cudss_factorize)
 *                            and interface_umfpack.c:141-200 (rcond, determinant outputs)
 *   solver_hipmf_solve       russell_sparse/c_code/interface_cudss.cu:510-566  (solver_cudss_solve)
 * Rust side that calls them: russell_sparse/src/solver_cudss.rs:25-52,194-360.
 *
 * Status codes: 0 = success; the shared 100000..700000 codes are those of
 * russell_sparse/c_code/constants.h:5-12 verbatim; 1 = "Matrix is singular" is the value UMFPACK
 * returns (russell_sparse/src/solver_umfpack.rs:492); the 100..1000 block is this backend's
 * analogue of the cuDSS block (constants.h:22-36).
 */
#ifndef RUSSELL_HIPMF_H
#define RUSSELL_HIPMF_H

#include <inttypes.h>
#include <stddef.h>

#ifdef __cplusplus
extern "C" {
#endif

#define C_BOOL int32_t

#define SUCCESSFUL_EXIT 0
#define ERROR_NULL_POINTER 100000
#define ERROR_MALLOC 200000
#define ERROR_VERSION 300000
#define ERROR_NOT_AVAILABLE 400000
#define ERROR_NEED_INITIALIZATION 500000
#define ERROR_NEED_FACTORIZATION 600000
#define ERROR_ALREADY_INITIALIZED 700000

#define HIPMF_WARNING_SINGULAR_MATRIX 1
#define ERROR_HIP_MALLOC 100
#define ERROR_HIP_MEMCPY 200
#define ERROR_HIP_SYNCHRONIZE 300
#define ERROR_HIP_LAUNCH 350
#define ERROR_HIPMF_INVALID_MATRIX 600
#define ERROR_HIPMF_SYMBOLIC 700
#define ERROR_HIPMF_INVALID_VALUE 803
#define ERROR_HIPMF_COMM 900 /* an RCCL call failed */
#define ERROR_HIPMF_NO_DEVICE 1000

/* ordering argument of solver_hipmf_initialize */
#define HIPMF_ORDERING_DEFAULT 0            /* nested dissection */
#define HIPMF_ORDERING_NESTED_DISSECTION 1
#define HIPMF_ORDERING_NONE 2               /* natural order (Ordering::No) */
#define HIPMF_ORDERING_AMD 3                /* approximate minimum degree on A + A^T (Ordering::Amd / Amf / Qamd; own implementation of the
                                               published method, symbolic.cpp) -- the better choice for patterns without small separators */
#define HIPMF_ORDERING_BEST 4               /* both orderings, the one whose factorisation needs fewer flops is kept (Ordering::Best, UMFPACK's
                                               meaning of the word); the effective ordering reports which one that was */
/* scaling argument (same numbering as UMFPACK_SCALE_*) */
#define HIPMF_SCALE_NONE 0
#define HIPMF_SCALE_SUM 1
#define HIPMF_SCALE_MAX 2

struct InterfaceHIPMF;

/* Allocates a solver bound to the calling thread's current HIP device.  Returns NULL on failure. */
struct InterfaceHIPMF *solver_hipmf_new(void);

/* Frees every host and device resource.  NULL-safe. */
void solver_hipmf_drop(struct InterfaceHIPMF *solver);

/* Phase 1 (once): ordering + symbolic analysis on the host, device allocation, structure upload.
 * pivot_epsilon < 0 and refinement_nstep < 0 select the defaults (1e-13 relative, 2 steps).
 * general_symmetric: the CSR holds the LOWER triangle of a symmetric matrix (Sym::YesLower).
 * values may be NULL.  When given (the reference hands the numbers to the analysis phase as well:
 * umfpack_di_symbolic(Ap, Ai, Ax) at interface_umfpack.c:109, cudssExecute(ANALYSIS) at interface_cudss.cu:361)
 * and the diagonal is weak (missing, zero or < 1 % of the row's largest entry somewhere), general-storage matrices
 * get a maximum-product matching + row/column scaling pre-permutation, which stays in force for every later
 * factorisation with this handle.  A symmetric-lower matrix with such a diagonal (indefinite: the Lagrange rows of
 * CooMatrix::put_lagrange_block, coo_matrix.rs:823-857) is mirrored to general storage inside the handle and takes the same
 * path (HIPMF_COUNTER_SYM_EXPANDED); the caller keeps passing lower-triangle values and lower-triangle value maps.  Round 4: a
 * symmetric-lower handle initialised WITHOUT values makes that decision at its first solver_hipmf_factorize (one more analysis inside
 * that call when the diagonal is weak; handles with a value map installed keep the L D L^T path). */
int32_t solver_hipmf_initialize(struct InterfaceHIPMF *solver,
                                int32_t ordering,
                                int32_t scaling,
                                double pivot_epsilon,
                                int32_t refinement_nstep,
                                C_BOOL verbose,
                                C_BOOL general_symmetric,
                                C_BOOL positive_definite,
                                int32_t ndim,
                                const int32_t *row_pointers,
                                const int32_t *col_indices,
                                const double *values);

/* Phase 2 (repeatable): numeric multifrontal LU of the same structure with new values.
 * Returns 0, or 1 when an exactly-zero pivot was met ("Matrix is singular").
 * Pivot order: static (nested dissection + the matching of initialize).  Every factorize checks the diagonal of the system it is
 * about to factorise; when THESE values call for another maximum-product matching (general storage: initialize had no values, or
 * the values changed a lot) the matching is recomputed from them and the analysis redone inside this call (cost of an initialize;
 * HIPMF_COUNTER_REMATCH counts it).  UMFPACK pivots dynamically in every numeric phase (interface_umfpack.c:167). */
int32_t solver_hipmf_factorize(struct InterfaceHIPMF *solver,
                               int32_t *effective_ordering,
                               int32_t *effective_scaling,
                               int32_t *num_perturbed_pivots,
                               double *rcond_estimate,
                               double *determinant_coefficient,
                               double *determinant_exponent,
                               C_BOOL compute_determinant,
                               C_BOOL verbose,
                               const double *values);

/* Phase 3: x = A^{-1} rhs (forward/backward level-set solves + iterative refinement). */
int32_t solver_hipmf_solve(struct InterfaceHIPMF *solver, double *x, const double *rhs, C_BOOL verbose);

/* ---- extensions that the reference's single-RHS boundary lacks ---------- */

/* nrhs right-hand sides, column-major with leading dimension ld >= ndim (host pointers) */
int32_t solver_hipmf_solve_many(struct InterfaceHIPMF *solver, double *x, const double *rhs, int32_t nrhs, int32_t ld,
                                C_BOOL verbose);

/* Optional, any time after initialize: allocates and touches the block buffers of a later solve_many / solve_device / solve_many_sharded with
 * up to `nrhs` right-hand sides (a few GB at config 4's size; the FIRST blocked solve of a handle otherwise pays for them: 0.4 s there).
 * A rank that waits for the factor of another rank (solver_hipmf_broadcast_factor) calls it while the root factorises.  nrhs < 2: no-op. */
int32_t solver_hipmf_prepare_solve_many(struct InterfaceHIPMF *solver, int32_t nrhs);

/* the same two phases with operands already resident in HBM (device pointers) */
int32_t solver_hipmf_factorize_device(struct InterfaceHIPMF *solver, const double *d_values);
int32_t solver_hipmf_solve_device(struct InterfaceHIPMF *solver, double *d_x, const double *d_rhs, int32_t nrhs, int32_t ld);

/* v = alpha * A * u on the device with the values of the last factorize (host pointers);
 * the CSR SpMV of russell_sparse/src/csr_matrix.rs:709-729 */
int32_t solver_hipmf_mat_vec_mul(struct InterfaceHIPMF *solver, double *v, double alpha, const double *u);

/* perm[new] = old: the fill-reducing permutation applied to rows and columns (ndim entries) */
/* Value refresh through a map (extension for the repeat-factorise callers, SURVEY.md 8f-2: Radau5 / Newton iterations call
 * LinSolTrait::factorize with the same structure and new values every step, radau5.rs:264-303; the reference converts
 * COO -> CSC/CSR on the host each time, csc_matrix.rs:365-505).  After solver_hipmf_initialize:
 *   solver_hipmf_set_value_map: CSR entry j (the order given to initialize) is the sum of the caller's entries
 *     input[seg_idx[q]], seg_ptr[j] <= q < seg_ptr[j+1]  (seg_ptr has nnz+1 entries, seg_ptr[nnz] = nnz_in; e.g. the COO
 *     triplets with their duplicates);
 *   solver_hipmf_factorize_mapped / _device: numeric factorisation from nnz_in caller-ordered values (host / device pointer):
 *     one gather kernel replaces the host conversion. */
int32_t solver_hipmf_set_value_map(struct InterfaceHIPMF *solver, int32_t nnz_in, const int32_t *seg_ptr, const int32_t *seg_idx);
int32_t solver_hipmf_factorize_mapped(struct InterfaceHIPMF *solver, int32_t *effective_ordering, int32_t *effective_scaling,
                                      int32_t *num_perturbed_pivots, double *rcond_estimate, double *determinant_coefficient,
                                      double *determinant_exponent, C_BOOL compute_determinant, C_BOOL verbose,
                                      const double *input_values);
int32_t solver_hipmf_factorize_mapped_device(struct InterfaceHIPMF *solver, const double *d_input_values);

int32_t solver_hipmf_get_permutation(struct InterfaceHIPMF *solver, int32_t *perm);

/* Host-only helper (no device needed): the maximum-product matching + scaling of an n x n CSR matrix that
 * solver_hipmf_initialize applies to weak-diagonal matrices.  matched_row[j] = row matched to column j;
 * |row_scale[i] * a_ij * col_scale[j]| <= 1 with equality on the matched entries.  Returns 0, or
 * ERROR_HIPMF_INVALID_MATRIX when the matrix is structurally singular. */
int32_t hipmf_max_product_matching(int32_t ndim, const int32_t *row_pointers, const int32_t *col_indices, const double *values,
                                   int32_t *matched_row, double *row_scale, double *col_scale);
/* The same for the REAL-EQUIVALENT form (order ndim2 = 2 n: rows / columns 2 k, 2 k + 1 = real and imaginary part of complex row /
 * column k, entry a + i b -> [a -b; b a]) of a complex matrix, as the complex twin applies it (round 4): the matching runs on the
 * moduli of the complex entries, a pair of rows moves as a whole (matched_row[2 k + 1] = matched_row[2 k] + 1) and shares its scales. */
int32_t hipmf_paired_matching(int32_t ndim2, const int32_t *row_pointers, const int32_t *col_indices, const double *values, int32_t *matched_row,
                              double *row_scale, double *col_scale);

/* istats[16]: 0 ndim, 1 nnz(A), 2 nsuper, 3 nlevels, 4 nnz(L) strict, 5 nnz(U) incl. diag, 6 max front,
 *             7 max pivots, 8 perturbed pivots, 9 zero pivots, 10 refinement steps, 11 factor launches,
 *             12 solve launches, 13 pool bytes (persistent factor + arena of working blocks), 14 maximum-product matching in
 *             force (0/1), 15 solves that fell back from the dependency-driven to the level-set launches (hand-off timeout)
 * dstats[16]: 0 flops, 1 gemm flops, 2 ordering s, 3 symbolic total s, 4 assemble ms, 5 factor ms, 6 fwd ms,
 *             7 bwd ms, 8 solve total ms, 9 last residual inf-norm; accumulated since the last reset (HIP events on
 *             the solver's stream): 10 assemble ms, 11 factor ms, 12 #factorizations, 13 forward-solve ms,
 *             14 backward-solve ms, 15 #triangular passes */
int32_t solver_hipmf_get_stats(struct InterfaceHIPMF *solver, int64_t *istats, double *dstats);
int32_t solver_hipmf_reset_timers(struct InterfaceHIPMF *solver);
/* further counters (-1: unknown counter or handle not initialized) */
#define HIPMF_COUNTER_REMATCH 0            /* factorisations that recomputed the maximum-product matching + analysis for new values */
#define HIPMF_COUNTER_WEAK_DIAGONAL_ROWS 1 /* rows with a weak diagonal under the pivot order, values of the last factorize */
#define HIPMF_COUNTER_FUSED_FALLBACKS 2    /* = istats[15] */
#define HIPMF_COUNTER_PERSISTENT_BYTES 3   /* pool: the factor proper (small fronts, E / E' panels) */
#define HIPMF_COUNTER_ARENA_BYTES 4        /* pool: arena of the tiled fronts' working blocks */
#define HIPMF_COUNTER_SYMMETRIC_LDLT 5     /* 1: the tiled fronts are factorised as L D L^T */
#define HIPMF_COUNTER_SYM_EXPANDED 6       /* 1: symmetric-lower input with a weak diagonal (indefinite / saddle-point): mirrored to general
                                            * storage at initialize, factorised by LU with the maximum-product matching; the caller keeps
                                            * handing over lower-triangle values */
#define HIPMF_COUNTER_MID_FRONTS 8        /* fronts one workgroup factorises in one launch per level (k_front_lu: f > 64, at most 32 pivots and 192 off-diagonal rows, LU mode; k_front where switched on) */
#define HIPMF_COUNTER_CHAIN_FALLBACKS 7    /* factorisations repeated with one launch per tiled step after a hand-off of a chained launch timed out */
#define HIPMF_COUNTER_PLAN_DIGEST 9       /* diagnostic: digest of the row structures, pool layout and extend-add task lists of the last initialize
                                             (computed only when HIPMF_PLAN_DIGEST is set in the environment, else 0): equal for every thread count */
#define HIPMF_COUNTER_TAGGED_SOLVE 10     /* 1: the triangular solves of one right-hand side hand their vectors from front to front as data-tagged
                                             words above the wave-subtrees (no completion counters; HIPMF_TAG_SOLVE=0 or fronts of >= 2 048 rows: 0) */
#define HIPMF_COUNTER_GATE_WAITS 11       /* solves of this handle that found the device's gate held by another handle (dependency-driven
                                             launches of two handles are never resident together) */
#define HIPMF_COUNTER_WAVE_FRONTS 12      /* big fronts (f > 64) whose forward solve step is the work of one wavefront (at most 128 rows, 32 pivots) */
#define HIPMF_COUNTER_LEAF_FRONTS 13      /* leaves of the tree that the blocked (many-RHS) solves run in kernels of their own, sixteen columns per wavefront */
#define HIPMF_COUNTER_SPLIT_SLABS 14      /* backward slabs of the blocked solves whose dot products are split over several tasks (levels of few slabs near the root of a large factor) */
#define HIPMF_COUNTER_EVENT_FENCE_FREE 15  /* 1: the events that order this handle's streams are recorded without the system-scope fence (taken on gfx950 under a
                                             HIP 7 runtime only, where it was validated; HIPMF_EVENT_FENCE=1 or any other device / runtime: 0 = default flags) */
#define HIPMF_COUNTER_BLOCK_GROUPS 16      /* blocks of right-hand sides that travel through ONE dependency-driven launch together in the many-RHS solves
                                             (round 6: their latency chains overlap; 1 = one block per launch) -- value of the last blocked solve, 0 before */
#define HIPMF_COUNTER_SYM_WEAK_DIAGONAL 17 /* 1: a symmetric-lower handle that kept its L D L^T plan met a weak diagonal in the values of a factorize
                                             (HIPMF_OPTION_SYM_RECHECK off: nothing was re-analysed; see last_error / verbose) */
#define HIPMF_COUNTER_BCAST_SLICED_BYTES 18 /* bytes of factor that solver_hipmf_broadcast_factor moved as slices over all xGMI links (two point-to-point
                                             steps; three or more ranks, parts of >= 64 MB) instead of through ncclBroadcast, summed over the calls */
#define HIPMF_COUNTER_KRYLOV_ITERATIONS 19 /* steps of the Krylov rescue in the last solve: after a factorisation that PERTURBED pivots (static pivot order:
                                             what UMFPACK's dynamic pivoting, interface_umfpack.c:167, would have avoided) a column whose refined solution
                                             leaves |b - A x|_2 > 1e-13 |b|_2 is finished by flexible GMRES preconditioned with the factorisation
                                             (rank(E) + 1 steps in exact arithmetic); 0: not needed.  HIPMF_KRYLOV=0 switches it off */
int64_t solver_hipmf_get_counter(struct InterfaceHIPMF *solver, int32_t which);

/* Options of LinSolParams that the initialize signature (kept in the shape of interface_cudss.cu:190-203 minus the cuDSS-only
 * arguments) does not carry: set them BEFORE solver_hipmf_initialize (ERROR_ALREADY_INITIALIZED afterwards).
 *   HIPMF_OPTION_MATCHING            lin_sol_params.rs:13 / enums.rs Matching: 0 = None (never), 1 = Auto (default: when values are
 *                                    handed to initialize and the diagonal is weak), 2 = always; every reference variant other than
 *                                    None / Auto selects THE matching this backend has (maximum product + scaling, MC64 job 5)
 *   HIPMF_OPTION_PIVOTING            lin_sol_params.rs:16 / enums.rs Pivoting as an integer (0 Auto, 1 None, 2 GlobalCol, 3 GlobalRow, 4 Diagonal,
 *                                    5 LocalBlock): a REQUEST, as in the cuDSS shim, which reads the effective strategy back after
 *                                    factorize (interface_cudss.cu:485-491).  Every value is accepted (round 6; rounds 1 - 5 refused all but
 *                                    Auto / LocalBlock); the kernels have ONE strategy, readable as HIPMF_OPTION_EFFECTIVE_PIVOTING = 5
 *                                    (LocalBlock): partial pivoting inside the pivot block of a small front / the 32-row diagonal tile of a
 *                                    tiled one; a pivot below pivot_epsilon max|a| is replaced by +-sqrt(machine eps) max|a| and counted;
 *                                    solves after such a factorisation are finished by a Krylov rescue when refinement is not enough
 *                                    (HIPMF_COUNTER_KRYLOV_ITERATIONS), and an exactly zero pivot means status 1 ("singular") only when a
 *                                    probe solve confirms it
 *   HIPMF_OPTION_HYBRID_MEMORY       lin_sol_params.rs:39 (cuDSS hybrid memory, factor 0.01 .. 0.99; interface_cudss.cu:347-380: the factor spills
 *                                    to host memory, factor x total is the device share).  Accepted, range-checked and kept (get_option
 *                                    returns it), and WITHOUT effect on what fits: this backend has no out-of-core path, the factor +
 *                                    working arena live in HBM (288 GB), so the option can neither let a larger matrix through nor --
 *                                    since round 5 -- refuse one that fits the device (rounds 3 - 4 applied factor x total as a cap,
 *                                    which refused matrices the reference accepts).  A matrix that does not fit the free device memory
 *                                    is refused by initialize with the "Not enough memory" string the reference's harness recognises
 *                                    (stats_lin_sol.rs:334-340); the message names the option when it was set
 *   HIPMF_OPTION_SYM_RECHECK         (no reference counterpart) 1: a symmetric-lower handle initialised WITHOUT values looks at the diagonal
 *                                    of the first values it is asked to factorise (solver_hipmf_factorize or _factorize_device) and, when
 *                                    it is weak, redoes the analysis on the mirrored matrix with the matching inside that call (what
 *                                    initialize does when it is handed values).  0 (default since round 5): it keeps its L D L^T plan --
 *                                    the re-analysis changes the plan of this handle only, so peers waiting for its factor
 *                                    (solver_hipmf_broadcast_factor) would be refused and a permutation fetched earlier goes stale
 *   HIPMF_OPTION_ERROR_ESTIMATES     lin_sol_params.rs:50: the componentwise backward error omega of the last solve is always kept
 *   HIPMF_OPTION_CONDITION_NUMBERS   lin_sol_params.rs:55: min|u_ii| / max|u_ii| is always reported by factorize (rcond_estimate)
 * (both readable with solver_hipmf_get_option after solve / factorize: the value, not the flag). */
#define HIPMF_OPTION_MATCHING 0
#define HIPMF_OPTION_PIVOTING 1
#define HIPMF_OPTION_HYBRID_MEMORY 2
#define HIPMF_OPTION_ERROR_ESTIMATES 3
#define HIPMF_OPTION_CONDITION_NUMBERS 4
#define HIPMF_OPTION_SYM_RECHECK 5
#define HIPMF_OPTION_EFFECTIVE_PIVOTING 6 /* get_option only */
int32_t solver_hipmf_set_option(struct InterfaceHIPMF *solver, int32_t option, double value);
int32_t solver_hipmf_get_option(struct InterfaceHIPMF *solver, int32_t option, double *value);

/* ---- many right-hand sides over the GPUs of one node (the reference has no counterpart: lin_solver.rs:51,
 * interface_cudss.cu:275,281 create b and x with ONE column).  One process (or thread) per GPU, every rank runs
 * solver_hipmf_initialize on the same structure (the analysis is deterministic), ONE rank factorises, the factor travels over
 * RCCL / xGMI, every rank solves its block of columns.
 *
 * The numeric factor of a handle = 4 device buffers (solver_hipmf_factor_parts): [0] the persistent part of the front pool (small
 * fronts, E / E' panels), [1] row interchanges (int32 x n), [2] scaling (f64 x n), [3] pivots (f64 x n).  A peer that ran
 * initialize on the same structure may be handed their contents by any transport and then calls solver_hipmf_adopt_factor.
 * solver_hipmf_broadcast_factor does it with ncclBroadcast in 256 MB messages on the solver's stream (plus the matrix values for the
 * refinement SpMV), every rank of `comm` calls it; `comm` is an ncclComm_t -- the caller's own, or one made by
 * hipmf_comm_unique_id (rank 0; send the 128 bytes to the others by any means) + hipmf_comm_init_rank (every rank), so that a host
 * language needs no RCCL binding of its own.  RCCL is loaded on first use (dlopen). */
#define HIPMF_COMM_ID_BYTES 128
int32_t solver_hipmf_factor_parts(struct InterfaceHIPMF *solver, int32_t max_parts, void **d_ptrs, int64_t *bytes); /* returns 4 */
int32_t hipmf_comm_unique_id(void *id128);
int32_t hipmf_comm_init_rank(void **comm, int32_t nranks, const void *id128, int32_t rank);
void hipmf_comm_destroy(void *comm);
int32_t solver_hipmf_broadcast_factor(struct InterfaceHIPMF *solver, void *comm, int32_t root, int32_t rank, double *seconds,
                                      int64_t *bytes_sent);
/* This rank's block of the nrhs_total columns (contiguous blocks, sizes differ by at most one): solves columns
 * [first, first + count) of the n x nrhs_total device arrays d_rhs -> d_x (column-major, leading dimension ld) in place. */
int32_t solver_hipmf_solve_many_sharded(struct InterfaceHIPMF *solver, double *d_x, const double *d_rhs, int32_t nrhs_total, int32_t ld,
                                        int32_t nranks, int32_t rank, int32_t *first_column, int32_t *num_columns);
int32_t solver_hipmf_adopt_factor(struct InterfaceHIPMF *solver, const double *d_values);

const char *solver_hipmf_last_error(struct InterfaceHIPMF *solver);

/* ---- complex (Complex64) twin ----------------------------------------------------------------------------------------------
 * Replaces russell_sparse/c_code/interface_complex_umfpack.c:40-248 (complex_solver_umfpack_new / _drop / _initialize / _factorize /
 * _solve) behind the Rust ComplexLinSolTrait (russell_sparse/src/complex_lin_solver.rs:12-104, complex_solver_umfpack.rs); Radau5
 * holds one of these next to the real solver (russell_ode/src/radau5.rs:48-51,264-301).  `values`, `x`, `rhs` are interleaved
 * (re, im) pairs: COMPLEX64 of c_code/constants.h:18 = double[2].  0-based CSR of the n x n complex matrix; general_symmetric: the
 * LOWER triangle of a complex SYMMETRIC (not Hermitian) matrix.  The system is solved in its real-equivalent form of order 2 n on
 * the same device path as real matrices; the expansion of the values happens on the device.  Round 4: the pivot searches take the two rows of a
 * complex row together (a complex LU with partial pivoting in real arithmetic), so the determinant IS available: determinant_coefficient_real /
 * _imag / _exponent as umfpack_zi_get_determinant returns them to interface_complex_umfpack.c:143-155,187-200 (det = (re + i im) x 10^exponent;
 * zeros when compute_determinant = 0).  set_value_map / factorize_mapped: as for the real solver, with the
 * caller's complex COO triplets as input (Radau5's K_comp = (alpha + i beta) M - J keeps its structure over the whole run). */
struct InterfaceComplexHIPMF;
struct InterfaceComplexHIPMF *complex_solver_hipmf_new(void);
void complex_solver_hipmf_drop(struct InterfaceComplexHIPMF *solver);
int32_t complex_solver_hipmf_initialize(struct InterfaceComplexHIPMF *solver, int32_t ordering, int32_t scaling, double pivot_epsilon,
                                        int32_t refinement_nstep, C_BOOL verbose, C_BOOL general_symmetric, int32_t ndim,
                                        const int32_t *row_pointers, const int32_t *col_indices, const double *values);
int32_t complex_solver_hipmf_factorize(struct InterfaceComplexHIPMF *solver, int32_t *effective_ordering, int32_t *effective_scaling,
                                       int32_t *num_perturbed_pivots, double *rcond_estimate, double *determinant_coefficient_real,
                                       double *determinant_coefficient_imag, double *determinant_exponent, C_BOOL compute_determinant, C_BOOL verbose,
                                       const double *values);
int32_t complex_solver_hipmf_get_determinant(struct InterfaceComplexHIPMF *solver, double *determinant_coefficient_real,
                                             double *determinant_coefficient_imag, double *determinant_exponent);
int32_t complex_solver_hipmf_solve(struct InterfaceComplexHIPMF *solver, double *x, const double *rhs, C_BOOL verbose);
int32_t complex_solver_hipmf_set_value_map(struct InterfaceComplexHIPMF *solver, int32_t nnz_in, const int32_t *seg_ptr, const int32_t *seg_idx);
int32_t complex_solver_hipmf_factorize_mapped(struct InterfaceComplexHIPMF *solver, int32_t *effective_ordering, int32_t *effective_scaling,
                                              int32_t *num_perturbed_pivots, double *rcond_estimate, C_BOOL verbose, const double *input_values);
int32_t complex_solver_hipmf_get_stats(struct InterfaceComplexHIPMF *solver, int64_t *istats, double *dstats); /* istats[0..1]: complex n, nnz */
/* the HIPMF_COUNTER_* values of solver_hipmf_get_counter for the complex twin's handle (-1: unknown counter / not initialized) */
int64_t complex_solver_hipmf_get_counter(struct InterfaceComplexHIPMF *solver, int32_t which);
const char *complex_solver_hipmf_last_error(struct InterfaceComplexHIPMF *solver);

/* plain device-memory helpers so that callers need no HIP binding of their own */
void *hipmf_device_malloc(size_t bytes);
void hipmf_device_free(void *ptr);
int32_t hipmf_memcpy_h2d(void *dst, const void *src, size_t bytes);
int32_t hipmf_memcpy_d2h(void *dst, const void *src, size_t bytes);
int32_t hipmf_device_synchronize(void);
int32_t hipmf_device_count(void);
int32_t hipmf_device_mem_info(size_t *free_bytes, size_t *total_bytes); /* hipMemGetInfo of the calling thread's device */
/* Measured device-to-device copy rate in GB/s (read + written bytes over HIP-event time, `bytes` per copy, best of `reps`):
 * the achievable-HBM denominator bench.py reports beside the 8 TB/s spec. */
int32_t hipmf_device_copy_bandwidth(int64_t bytes, int32_t reps, double *gb_per_s);
/* Measured rate of v_mfma_f64_16x16x4_f64 in TFLOP/s (back-to-back MFMAs on four independent accumulators per wave, operands in
 * registers, `workgroups` x 256 threads, best of two timed launches): the achievable-MFMA denominator bench.py reports beside the
 * 78.6 TFLOP/s spec. */
int32_t hipmf_device_mfma_rate(int32_t workgroups, int32_t iters, double *tflops);
int32_t hipmf_set_device(int32_t device); /* selects the device later solver_hipmf_new() calls of this thread bind to */

/* ---- finite-difference Laplacian assembled on the device -------------------------------------------------
 * Replaces, for callers that keep the matrix in HBM, Fdm2d::get_matrices_sps of russell_pde
 * (/root/reference/russell_pde/src/fdm_2d.rs:603-649; molecule :376-386; mirrored / wrapped neighbours :944-979; local numbering
 * russell_pde/src/equation_handler.rs:153-190): the COO triplets of K-bar (unknown x unknown) and K-check (unknown x prescribed)
 * in the reference's order -- unknown nodes ascending, per node CUR, LEF, RIG, BOT, TOP -- duplicates of mirrored ghost nodes kept.
 *   nz = 1: the reference's 2D operator; nz > 1: the 7-point analogue (two more molecule entries: the z neighbours)
 *   sym: 0 every entry (Sym::No / YesFull), 1 lower triangle (Sym::YesLower), 2 upper triangle (Sym::YesUpper)
 *   prescribed: HOST array of nx*ny*nz bytes, non-zero = node with an essential boundary condition (NULL: none)
 * hipmf_fdm_new returns NULL on invalid sizes or allocation failure.  The *_device calls write DEVICE arrays of the sizes reported by
 * hipmf_fdm_dims (the K-check pointers may be NULL when np = 0); indices are the local numbers iu / ip.  Values can be regenerated
 * for new coefficients without touching the structure and handed to solver_hipmf_factorize_mapped_device. */
void *hipmf_fdm_new(int32_t nx, int32_t ny, int32_t nz, int32_t periodic_x, int32_t periodic_y, int32_t periodic_z, int32_t sym,
                    const uint8_t *prescribed);
void hipmf_fdm_drop(void *fdm);
int32_t hipmf_fdm_dims(const void *fdm, int64_t *nu, int64_t *np, int64_t *nnz_bar, int64_t *nnz_check);
int32_t hipmf_fdm_structure_device(const void *fdm, int32_t *d_bar_i, int32_t *d_bar_j, int32_t *d_check_i, int32_t *d_check_j);
int32_t hipmf_fdm_values_device(const void *fdm, double dx, double dy, double dz, double kx, double ky, double kz, double alpha,
                                double *d_bar_values, double *d_check_values);
/* The Lagrange-multiplier form of the same operator: replaces Fdm2d::get_matrices_lmm (fdm_2d.rs:672-748) -- the augmented matrix
 *   M = [K C^T; C 0]  of order neq + nlag  (neq = nx*ny*nz: every node keeps its equation; nlag = prescribed nodes)
 * as COO triplets in the reference's order: the molecule of every node ascending (entries above / below the diagonal skipped for
 * lower / upper storage), then per prescribed node, ascending, the entry of C (row neq + ip, column m) and / or of C^T (lower storage
 * keeps C, upper C^T, general storage both, C first); global node numbers as indices.  The matrix is a saddle-point system: handed to
 * solver_hipmf_initialize WITH values (or to a symmetric-lower handle) it takes the matched path, no perturbed pivots.  The constraint
 * matrix C of the reference's second return value is the (row - neq, column) view of the C entries.  The first call builds the offsets. */
int32_t hipmf_fdm_lmm_dims(void *fdm, int64_t *neq, int64_t *nlag, int64_t *nnz);
int32_t hipmf_fdm_lmm_structure_device(void *fdm, int32_t *d_i, int32_t *d_j);
int32_t hipmf_fdm_lmm_values_device(void *fdm, double dx, double dy, double dz, double kx, double ky, double kz, double alpha, double *d_values);

#ifdef __cplusplus
}
#endif
#endif
