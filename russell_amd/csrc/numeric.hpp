// numeric.hpp -- device-side state and drivers of the multifrontal LU backend (factorize / solve).
#pragma once
#include <cstdint>
#include <functional>
#include <mutex>
#include <string>
#include <vector>

#include "symbolic.hpp"

namespace hipmf {

struct FrontDesc;
struct SmallDesc;
struct EaTask;
struct EaRange;
struct SolveTask;
struct SfTask;
struct WtHdr;
struct WtWave;
struct LeafRec;
struct ZeroTask;
struct FactorInfo;

// status codes shared with the reference's C shims (/root/reference/russell_sparse/c_code/constants.h:5-12)
enum : int32_t {
    SUCCESSFUL_EXIT = 0,
    ERROR_NULL_POINTER = 100000,
    ERROR_MALLOC = 200000,
    ERROR_VERSION = 300000,
    ERROR_NOT_AVAILABLE = 400000,
    ERROR_NEED_INITIALIZATION = 500000,
    ERROR_NEED_FACTORIZATION = 600000,
    ERROR_ALREADY_INITIALIZED = 700000,
    // HIP block, analogous to the cuDSS block of constants.h:22-36
    ERROR_HIP_MALLOC = 100,
    ERROR_HIP_MEMCPY = 200,
    ERROR_HIP_SYNCHRONIZE = 300,
    ERROR_HIP_LAUNCH = 350,
    ERROR_HIPMF_INVALID_MATRIX = 600,
    ERROR_HIPMF_SYMBOLIC = 700,
    ERROR_HIPMF_INVALID_VALUE = 803,
    ERROR_HIPMF_COMM = 900,
    ERROR_HIPMF_NO_DEVICE = 1000,
    // numerical status, same value UMFPACK uses for a singular matrix (solver_umfpack.rs:492)
    WARNING_SINGULAR_MATRIX = 1,
};

struct NumericOptions {
    int32_t scaling = 1;            // 0 none, 1 sum (UMFPACK_SCALE_SUM), 2 max
    double pivot_epsilon = 1e-13;   // relative to max|scaled a_ij| (cuDSS documents 1e-13 as its f64 default)
    int32_t refinement_nstep = 2;   // UMFPACK's default UMFPACK_IRSTEP is 2
    int32_t matching = 1;           // maximum-product matching + scaling at initialize: 0 never, 1 when the diagonal is weak, 2 always
    double device_memory_factor = 0.0; // > 0: the factor + arena may take at most this share of the device's TOTAL memory (the device limit the
                                    // reference derives from hybrid_memory_factor, interface_cudss.cu:364-372); there is no host spill here: a pool
                                    // beyond the limit is refused by initialize with the out-of-memory status
                                    // (needs the values at initialize; general storage only)
    bool complex_pairs = false;     // the system is the real-equivalent form of a complex matrix (rows / columns 2 k, 2 k + 1 = Re, Im of complex
                                    // row / column k; interface_complex_hipmf.cpp): ordering, matching and pivot searches keep the pairs together,
                                    // and the factorisation leaves the complex pivots for determinant_complex()
    bool verbose = false;
};

struct PhaseTimes {
    double scale_assemble_ms = 0, factor_ms = 0, fwd_ms = 0, bwd_ms = 0, solve_total_ms = 0;
    int64_t n_kernel_launches_factor = 0, n_kernel_launches_solve = 0;
    // accumulated over calls since the last reset (HIP events on the solver's own stream)
    double acc_factor_ms = 0, acc_assemble_ms = 0, acc_fwd_ms = 0, acc_bwd_ms = 0;
    int64_t acc_factor_count = 0, acc_tri_count = 0;
};

struct StepPlan {
    int32_t nactive = 0;
    int64_t pfx_panel = 0, pfx_update = 0; // offsets into d_tasks
    int32_t n_panel = 0, n_update = 0;
    int32_t ppfx[3] = {0x7fffffff, 0x7fffffff, 0x7fffffff}, upfx[3] = {0x7fffffff, 0x7fffffff, 0x7fffffff}; // prefix words of slots 1 .. 3 (kernel arguments of launches with at most four fronts)
    // a full step whose trailing update runs in two launches: the critical strips (first block column / row + look-ahead) on the main
    // stream, the other tiles on a side stream beside the next group's panel steps
    bool split = false;
    bool all_narrow = false; // every active front is inside a group of panels: the update only touches the next panel's strips
    int64_t pfx_crit = 0, pfx_rest = 0;
    int32_t n_crit = 0, n_rest = 0;
};
struct LevelPlan {
    int32_t small_off = 0, small_cnt = 0, small_ld = 0, small_pmax = 1; // fronts with f <= SMALL_F
    int32_t small_cnt_a = 0, small_ld_a = 1;                              // ... the first small_cnt_a of them have f <= small_split
    int32_t big_off = 0, big_cnt = 0;                   // tiled path, sorted by p descending
    int32_t bigfd_off = 0;                              // their descriptors in d_bigfd (slot order)
    int32_t ea_off = 0, ea_cnt = 0;
    int32_t mirror_off = 0, mirror_cnt = 0; // symmetric mode: tiled fronts of the level with a small parent (k_mirror_cb)
    int32_t zero_off = 0, zero_cnt = 0; // zero-fill tasks of the level's working blocks
    int32_t sc_off = 0, sc_cnt = 0;     // entries of A scattered into the level's working blocks
    int32_t fwd_off = 0, fwd_cnt = 0, bwd_off = 0, bwd_cnt = 0; // SolveTask ranges of the big fronts
    int32_t big_pmax = 0, big_fmax = 0;
    bool wide = false; // solve with 32-row slabs x 32 column groups (few large fronts)
    std::vector<StepPlan> steps;
    int64_t pfx_flush = 0;  // k_eflush: prefix of tasks per tiled front (into d_tasks), their total
    int32_t n_flush = 0;
    int32_t upd_ts = 64;    // edge of the trailing-update tiles on this level (32: k_update32, one wave per tile)
    // fronts of the middle of the tree that ONE workgroup carries through their whole partial factorisation (k_front,
    // kernels_factor_front.hpp): their descriptors follow the tiled ones in d_bigfd, grouped by size class
    int32_t mid_off = 0;                 // first of them in d_bigfd
    static constexpr int MID_CLASSES = 6;
    int32_t mid_cnt[MID_CLASSES] = {0, 0, 0, 0, 0, 0}; // fronts per class: 0 .. 2 k_front with 10 / 16 / 24 columns of F12 per wavefront; 3 .. 5 k_front_lu (at most 32 pivots) by the LDS a front needs: up to 40 KB (four workgroups per CU), 80 KB (two), more (one)
    int32_t mid_lds[MID_CLASSES] = {0, 0, 0, 0, 0, 0}; // dynamic LDS of the class's launch, in doubles (its largest front)
    int32_t mid_total() const {
        int32_t t = 0;
        for (int c = 0; c < MID_CLASSES; c++) t += mid_cnt[c];
        return t;
    }
    int64_t chain_off = 0;  // the level's tiled steps as ONE launch (k_chain): its tasks in d_chain, chain_cnt of them (0: one launch per step)
    int32_t chain_cnt = 0;
};

class Solver {
  public:
    Solver();
    ~Solver();
    int32_t initialize(int32_t n, const int32_t *rp, const int32_t *ci, bool sym_lower, const SymbolicOptions &sopt,
                       const NumericOptions &nopt, const double *values = nullptr);
    // values: nnz doubles in the CSR order given to initialize; on_device tells where they live
    int32_t factorize(const double *values, bool on_device);
    // value refresh through a map: CSR entry j = sum of input[seg_idx[seg_ptr[j] .. seg_ptr[j + 1])]
    // signed_map: seg_ptr[nnz] map entries (any number), an index ~k subtracts input[k]
    int32_t set_value_map(int64_t nnz_in, const int32_t *seg_ptr, const int32_t *seg_idx, bool signed_map = false);
    int32_t factorize_mapped(const double *input, bool on_device);
    int32_t solve(double *x, const double *rhs, int32_t nrhs, int64_t ldx, bool on_device);
    int32_t solve_core(double *x, const double *rhs, int32_t nrhs, int64_t ldx, bool on_device); // the driver of solve(): triangular passes + refinement
    int32_t krylov_rescue(double *x, const double *rhs, bool on_device);                          // see numeric.cpp
    int32_t singular_verdict();      // exactly zero pivots were met: singular (status 1), or only an unlucky static order (0)?  One probe solve decides
    int64_t zero_pivots_absorbed = 0; // exactly zero pivots of factorisations that the probe solve found NOT singular (summed)
    bool krylov_enabled = true;      // HIPMF_KRYLOV=0: no rescue (the refined solution is returned as it is)
    bool in_rescue = false;
    int32_t krylov_restart = 40;     // directions per cycle (HIPMF_KRYLOV_RESTART)
    int32_t krylov_cycles = 4;       // restarts at most
    double krylov_omega_ok = 1e-13;  // a column whose refined solution has a componentwise backward error up to this is not looked at by the rescue (HIPMF_KRYLOV_OMEGA)
    std::vector<double> col_omega;   // per column of the last solve: omega of its final solution as the refinement measured it
    double krylov_tol = 1e-13;       // |b - A x|_2 <= tol |b|_2 ends the rescue (and is what triggers it)
    int64_t krylov_iterations = 0;   // FGMRES steps of the last solve (0: no rescue ran or was needed)
    double krylov_last_relres = 0.0; // |b - A x|_2 / |b|_2 after the last rescue
    int32_t prepare_many(int32_t nrhs); // the block buffers of a later many-RHS solve, ahead of time
    int32_t spmv(double *y, const double *x, double alpha, bool on_device); // y = alpha A x with the factorize()d values
    int32_t determinant(double *mantissa, double *exponent, double *rcond);
    int32_t rcond_estimate(double *rcond); // min |u_ii| / max |u_ii| by a device reduction
    int32_t determinant_complex(double *mantissa_re, double *mantissa_im, double *exponent, double *rcond); // opt.complex_pairs: det of the COMPLEX matrix = (re + i im) x 10^exponent
    int32_t adopt_factor(const double *d_values); // factor buffers were filled by a peer (many-RHS multi-GPU path)
    // The caller's value arrays hold nnz_lower entries (the lower triangle it handed to the C-ABI) while the handle was analysed with
    // the mirrored general matrix: entry k of the handle's CSR is entry emap[k] of the caller's.  Every entry point that takes CSR
    // values expands them on the device.
    int32_t set_expansion(int64_t nnz_lower, const std::vector<int32_t> &emap);
    int64_t expansion_inputs() const { return nnz_low; }
    const std::vector<int32_t> &expansion_map() const { return h_emap; }
    void mark_factor_adopted(int32_t root_perturbed = 0) { n_perturbed = root_perturbed, n_zero_pivot = 0, factorized = true; } // ... including the matrix values; the root's count of replaced pivots decides about the Krylov rescue here too
    void *d_diag_ptr() const { return d_diag; }
    // FNV-1a over what two handles must share to exchange a factor: the fill-reducing permutation, the matching's row permutation, the
    // layout of the pool (computed at initialize / after a re-matching factorize)
    uint64_t plan_signature() const { return plan_sig; }
    uint64_t plan_sig = 0;
    int64_t plan_digest = 0; // HIPMF_PLAN_DIGEST set at initialize: digest of the row structures, pool layout and extend-add task lists (else 0)
    int64_t nnz_in_values() const { return nnz_in; } // inputs the installed value map reads (0: none)
    const std::vector<int32_t> &kept_row_pointers() const { return h_rp_keep; } // the caller's CSR structure as handed to initialize
    const std::vector<int32_t> &kept_col_indices() const { return h_ci_keep; }
    void *d_vals_ptr() const { return d_vals; }
    void release();

    Symbolic S;
    NumericOptions opt;
    PhaseTimes times;
    bool initialized = false, factorized = false;
    int32_t n_perturbed = 0, n_zero_pivot = 0;
    int32_t n_weak_diag = 0;   // rows with a weak diagonal under the current pivot order, for the values of the last factorize
    int64_t rematch_count = 0; // factorisations that recomputed the maximum-product matching (and the analysis) for new values
    int32_t refinement_steps_done = 0;
    int64_t fused_fallbacks = 0; // solves that fell back to the level-set launches after a hand-off timeout (never expected)
    int64_t mid_front_count = 0; // fronts of this plan that one workgroup factorises in one launch (k_front)
    int64_t wave_front_count = 0; // big fronts that are wave-front tasks in the forward pass of this plan (sf_fwd_wave)
    int32_t block_groups_last = 0;   // blocks of right-hand sides per dependency-driven launch in the last blocked solve (round 6)
    bool sym_diag_looked = false;    // the first values of a symmetric-lower handle were checked for a weak diagonal (once per initialize)
    bool sym_weak_diag_seen = false; // L D L^T plan kept although a factorize met a weak diagonal (HIPMF_OPTION_SYM_RECHECK off)
    bool event_fence_free = false; // the events between this handle's streams are recorded without the system-scope fence (gfx950 + HIP 7 only)
    int64_t gate_waits = 0;      // solves that waited for another handle's solve on the same device (device_gate, numeric.cpp)
    bool tagged_solve() const { return tag_active && use_fused; }
    int64_t leaf_front_count() const { return use_fused ? leaf_cnt : 0; }
    int64_t split_slab_count() const { return use_fused ? split_slabs : 0; }
    int64_t chain_fallbacks = 0; // factorisations repeated with one launch per tiled step after a hand-off timeout of a chained launch (never expected)
    int64_t persist_bytes() const { return S.persist_doubles * 8; }
    double last_residual_inf = 0.0, last_omega = 0.0;
    int device = 0;
    void *stream = nullptr;
    void *stream2 = nullptr;          // the small fronts of a level are factorised beside its tiled steps
    void *ev_fork = nullptr, *ev_join = nullptr;
    // the launches of a factorisation's levels, captured once and replayed (HIPMF_FACTOR_GRAPH=0: eager launches)
    bool use_graph = false; // (measured: no gain at 1000 x 1000 -- the gaps at the level boundaries are the cross-stream edges themselves, not host latency)
    void *factor_graph = nullptr;
    int64_t graph_launches = 0;
    bool use_binv = false;            // HIPMF_BLOCK_INV=1: LU fronts of the tiled path take ONE launch per step (kernels_factor_binv.hpp) instead of k_panel + k_update; measured slower (profiles/r04_rejected_experiments.txt)
    void *stream4 = nullptr;          // ... and the bulk of a split trailing update (HIPMF_UPD_SPLIT)
    void *ev_pb = nullptr, *ev_rest = nullptr;
    void *ev_pre0 = nullptr, *ev_pre1 = nullptr; // the work before the first level that only the big fronts need (zero-fill of E / E', identity blocks, diagonal check) runs on stream4 beside the levels of small fronts
    int32_t upd_split_min = 0;        // HIPMF_UPD_SPLIT=n: full steps (two-launch form) with at least n update workgroups are split (0: never; measured: no gain)
    void *stream3 = nullptr;          // ... and so are the fronts one workgroup factorises (k_front)
    void *ev_fork3 = nullptr, *ev_join3 = nullptr;
    std::string last_error;
    std::mutex err_mutex; // (the planning thread of initialize and the calling thread both report through last_error)

    // exported for the many-RHS / multi-GPU paths: the factor lives in [d_pool, d_pool + pool_doubles)
    double *d_pool = nullptr;
    int64_t pool_doubles = 0;
    int32_t *d_lperm = nullptr;
    double *d_rs = nullptr;
    double *d_cs = nullptr;       // column scaling of the matching (nullptr: none)
    int32_t *d_rperm = nullptr;   // row of A that is row i of the permuted system (== d_perm without matching)
    bool matched = false;         // a maximum-product matching pre-permutation is in force
    int32_t match_parity = 0;     // parity of its row permutation (for the determinant)

  private:
    int32_t initialize_impl(int32_t n, const int32_t *rp, const int32_t *ci, bool sym_lower, const SymbolicOptions &sopt,
                            const NumericOptions &nopt, const double *values);
    int32_t upload_plan(const std::function<int32_t()> &tail); // tail: what initialize runs after the descriptor uploads, beside the solve task lists
    int32_t rematch_and_factorize(); // the values in d_vals invalidate the pivot order: new matching + analysis, then factorize
    // what a re-analysis needs: the caller's structure and options, the value map
    std::vector<int32_t> h_rp_keep, h_ci_keep, h_seg_ptr, h_seg_idx;
    SymbolicOptions sopt_keep;
    bool sym_lower_keep = false, rematching = false;
    bool rematch_futile = false; // the last re-matching left a weak diagonal: no further attempts for this handle
    int32_t *d_dcol = nullptr; // column of the (matched) diagonal entry of every row of A (nullptr: identity)
    int32_t run_factor();
    // forward + backward on nk permuted, scaled vectors (column c at xp + c * xstr, its workspace at wrk + c * wstr)
    struct SolveLane;
    int32_t run_triangular(double *xp, int32_t nk, double *wrk, int64_t xstr, int64_t wstr, void *lane_stream, int32_t *lane_sync, bool timed, int32_t lane_id, uint32_t gmask = 0xffffffffu);
    void harvest_tri();
    int32_t build_level_tasks();
    // optional task list of the blocked (many-RHS) instances with wider slabs (HIPMF_BLOCKED_SLABS=1).  Measured and NOT the default:
    // 64 / 128-row slabs re-read the vector block of a front less often, but lose more in parallelism -- 144^3, 64 right-hand sides:
    // 4.92 -> 7.06 ms per right-hand side; 200^3, 256 right-hand sides: 4.58 -> 6.86 s (profiles/r03_rejected_experiments.txt)
    bool blocked_slabs = false;
    bool blocked_slabs_env = false; // HIPMF_BLOCKED_SLABS was given (else: on for factors whose largest front has >= 16 384 rows when several groups share a launch)
    // round 5: the leaves of the tree leave the task lists of the blocked (many-RHS) solves: one wavefront carries sixteen columns through
    // LEAF_PER_WAVE leaves (kernels_solve_leaf.hpp).  HIPMF_LEAF_KERNELS=0: every small front stays a task.
    bool leaf_kernels = true;
    LeafRec *d_leaf = nullptr;   // records of the leaves: forward part, then backward part (other panel offsets)
    int32_t leaf_cnt = 0;
    // blocked instances, backward pass, levels of few slabs with long dot products (the top of a 3D factor): a slab's dot products are
    // split over Q consecutive tasks (k_bwd_fused); forward, the largest fronts of such levels get narrower slabs.  A level qualifies
    // below split_tasks slabs (HIPMF_SPLIT_TASKS, 0: never), a front from split_minlen rows on (HIPMF_SPLIT_MINLEN).
    int32_t split_tasks = 512, split_minlen = 2048; // (split_minlen: 2 048 x the planned block groups unless HIPMF_SPLIT_MINLEN says otherwise)
    bool split_minlen_env = false;
    int64_t split_units = 0;       // 256-double units of the scratch of partial sums
    int64_t split_slabs = 0;
    double *d_split_scr = nullptr;
    int32_t *d_split_cnt = nullptr; // one arrival counter per unit (the first unit of a slab's group is used; reset by the last arriver)
    SfTask *d_sfk = nullptr;
    int32_t *d_needk = nullptr;
    int32_t sfk_fwd_cnt = 0, sfk_bwd_cnt = 0, sfk_fwd_band = 0, sfk_bwd_top = 0;
    bool rearm_tags = false;                   // HIPMF_REARM_TAGS=1: k_wt_bwd re-arms the tagged words for the next pass pair instead of a memset before every pass pair
                                               // (built and measured in round 6: same bits, NOT faster -- pass pair 0.4037 - 0.4076 against 0.4012 - 0.4058 ms, solve
                                               // 0.94 - 0.95 against 0.91 ms, profiles/r06_rejected_experiments.txt; default off)
    bool tags_armed = false;                   // the tagged words of d_work hold the tag pattern (left by the last pass pair's k_wt_bwd)
    bool plain_band = true;                    // HIPMF_PLAIN_BAND=0: the all-small band of the blocked solves stays ONE dependency-driven launch per direction
    std::vector<int32_t> sfk_band_f, sfk_band_b; // task offsets of the band's levels in d_sfk (forward: levels ascending from task 0; backward: relative to
                                               // the first backward task, the band's levels descending) -- one PLAIN launch per level (round 6)
    SfTask *d_sf3 = nullptr;   // tasks of the level-by-level launches of the dependency-driven kernels (fallback of L D L^T / very large fronts)
    int32_t *d_need3 = nullptr;
    std::vector<int32_t> sf3_lvl, sf3_lvl_b; // task offsets per level: forward (leaves first), backward (root first)
    bool tri_pending = false;
    std::vector<LevelPlan> levels;
    int64_t work_doubles = 0;
    int32_t allbig_off = 0, allbig_cnt = 0, dws_stride = 1;
    // device buffers
    FrontDesc *d_fd = nullptr;
    FrontDesc *d_bigfd = nullptr; // copies of the tiled fronts' descriptors, level by level in slot order
    EaTask *d_ea = nullptr;
    EaRange *d_ear = nullptr;
    double *d_dws = nullptr; // factorised diagonal tiles of the current tiled step, one per active big front
    SolveTask *d_st = nullptr;
    // dependency-driven solve (kernels_solve_fused.hpp): one launch per direction
    SfTask *d_sf = nullptr;
    int32_t sf_fwd_cnt = 0, sf_bwd_cnt = 0; // forward tasks first, then the backward tasks
    int32_t sf_fwd_band = 0, sf_bwd_top = 0; // tasks of the all-small bottom band (forward: first; backward: after sf_bwd_top)
    int32_t sf_fwd_launch = 0;              // == sf_fwd_cnt unless the profiling knob HIPMF_SF_FWD_LEVELS cuts the pass short
    // bottom of the tree: one wavefront per subtree of small fronts (kernels_solve_tree.hpp); the tasks of everything above it
    // (single right-hand side; the blocked many-RHS instances keep the task lists above)
    bool use_tree = true;                   // HIPMF_TREE_SOLVE=0: the round-2 schedule (all-small band + upper band)
    int32_t wt_max_fronts = 24;             // HIPMF_WT_FRONTS: fronts per wave-subtree at most
    int32_t wt_max_kb = 64;                 // HIPMF_WT_KB: panel kilobytes per wave-subtree at most
    int32_t up_stage = 32;                  // HIPMF_UP_STAGE: entries of E per thread of a TOP-level forward slab parked in LDS before the wait (multiple of 8; 0: no top launch)
    int32_t up_stage_bwd = 48;              // HIPMF_UP_STAGE_BWD: the same for E' in the backward pass (the dot products run over f, not p)
    int32_t up_stage_mid = 8;               // HIPMF_UP_STAGE_MID: the same for the backward slabs BELOW the top levels (0: none)
    int32_t up_top_fronts = 40;             // HIPMF_UP_TOP_FRONTS: the top levels are those above which no level has more tiled fronts
    WtHdr *d_wt_hdr = nullptr;              // batch headers: forward part, then backward part
    int32_t *d_wt_meta = nullptr;           // meta blocks of the batches (records + index lists): forward part, then backward part
    WtWave *d_wt_wave = nullptr;            // batch range and first pivot column of every wave-subtree: forward part, then backward part
    int32_t wt_waves = 0, wt_recs = 0;      // wave-subtrees, fronts they hold
    int32_t wt_hdr_fwd = 0, wt_hdr_bwd = 0; // batches of the forward / backward part
    int64_t wt_meta_fwd = 0;                // words of the forward part of d_wt_meta
    SfTask *d_sf2 = nullptr;                // tasks of the fronts above the wave-subtrees: forward, then backward
    int32_t sf2_fwd_cnt = 0, sf2_bwd_cnt = 0;
    int32_t sf2_fwd_mid = 0, sf2_bwd_top = 0; // forward: tasks below the top levels come first; backward: the top levels' tasks come first
    int32_t *d_rep_idx = nullptr;           // per front: index of its "complete" replicas (tiled fronts of the top levels), else -1
    int32_t *d_rep = nullptr;               // the replicas: forward part, then backward part (zeroed before every pass)
    int64_t rep_words = 0;
    bool up_pair_xcd = true;                // HIPMF_UP_PAIR_XCD=0: the 8-row slabs of a top-level front in row order (1: neighbours eight tasks apart = same XCD)
    int32_t up_max_groups = 32;             // HIPMF_UP_MAX_GROUPS: column groups of a top-level slab at most (32: 8-row slabs; 16: 16-row slabs = whole 128-byte lines)
    bool use_rep = true;                    // HIPMF_UP_REPLICAS=0: every waiter polls the front's counter
    int32_t *d_need2 = nullptr;             // completed-task counts of that list (the slabs are cut differently)
    bool tree_active = false;               // the plan above exists for this matrix
    bool use_tag = true;                    // HIPMF_TAG_SOLVE=0: completion counters instead of data-tagged hand-offs above the wave-subtrees
    bool wave_fronts = true;                // HIPMF_WAVE_FRONTS=0: the big fronts of few rows / pivots right above the wave-subtrees stay 256-thread slab tasks in the forward pass
    int32_t mid_bwd_len4 = 256, mid_bwd_len5 = 64; // backward slabs below the top levels, tagged hand-offs: 16 / 32 rows from these dot lengths on (HIPMF_MID_BWD_LEN4 / _LEN5)
    // host-pointer solves: buffers seen in the previous call are copied without the pinned staging buffer (HIPMF_HOST_DIRECT=0: always staged)
    bool host_direct = true;
    const double *last_host_rhs = nullptr;
    double *last_host_x = nullptr;
    bool wave_fronts_bwd = true; // ... and of the backward pass (HIPMF_WAVE_FRONTS_BWD=0: slab tasks there)
    bool tag_active = false;                // the launches above the wave-subtrees run their TAG instances (kernels_solve_fused.hpp, sf_tag_wait)
    int64_t work_arm0 = 0;                  // ... of which the first work_arm0 doubles (the roots of the wave-subtrees) are not armed: k_wt_fwd writes them in a launch of its own
    int64_t work_up = 0;                    // doubles at the head of a solve workspace: the vectors of the fronts outside the wave-subtrees' interiors
                                            // (tag_active: followed by n doubles, the tagged shadow of x)
    int32_t *d_need = nullptr;              // completed-task counts that mark a front as done: [0, ns) forward, [ns, 2 ns) backward
    int32_t *d_sync = nullptr;              // 2 x (SF_SYNC_HEADER + ns) ints: ticket, error word, counters; zeroed before every pass
    bool overlap_small = true;              // HIPMF_OVERLAP_SMALL=0: everything on one stream
    bool small_pair = false;                // HIPMF_SMALL_PAIR=1: the two k_small_factor launches of an all-small level side by side (third stream)
    int32_t sf_big_rows = 6, sf_big_front = 2048; // forward solve: fronts with at least sf_big_front rows use slabs of 2^sf_big_rows rows
    int32_t sf_asm_front = 2048;                  // forward solve: fronts with at least this many rows assemble their vector once, in tasks of their own (0: never; then sf_big_rows applies)
    // levels with few tiled steps (the middle of the tree): all steps of a level in one launch with in-launch hand-offs (kernels_factor_chain.hpp)
    bool use_chain = false;                 // HIPMF_FACTOR_CHAIN=1 switches it on.  Off by default: measured at the end of round 3 it is neutral for LU at
                                            // 1000 x 1000 (7.35 -> 7.34 ms), -1.6 % for L D L^T there and +1 ... +3 % on smaller problems (profiles/r03_chain_check.txt)
    bool chain_fine = false;                // HIPMF_CHAIN_FINE=1: a panel waits only for the critical pieces of the update before (measured: no gain)
    int32_t chain_max_steps = 8;            // a level is chained when it has at most this many steps (HIPMF_CHAIN_MAX_STEPS) ...
    int32_t chain_max_update = 16384;       // ... no step of it has more update workgroups than this (HIPMF_CHAIN_MAX_WGS: wide steps are bound by throughput, and
                                            //     the 8-byte agent-scope accesses of the chained form cost more per byte) ...
    int32_t chain_min_update = 0;           // ... and its widest step has at least this many (HIPMF_CHAIN_MIN_WGS)
    void *d_chain = nullptr;                // ChainTask records of all chained levels
    int32_t *d_chain_cnt = nullptr;         // their counters (3 per front and step) + the error word (last); zeroed before every factorisation
    int64_t chain_words = 0;
    int32_t upd32_max_front = 256;          // LU: levels whose largest tiled front has at most this many rows update with 32 x 32 tiles, one wave per tile
                                            // (HIPMF_UPD32_MAXF; 0: never).  Bit-identical to the 64 x 64 instance; 1000 x 1000: 7.30 -> 7.23 ms.  The L D L^T
                                            // fronts keep the 64 x 64 tiles (measured: 6.32 -> 6.35 ms with the small ones)
    // LU mode: fronts with f > 64, at most 64 pivots and at most mid_mmax off-diagonal rows whose pivot rows + column panel fit the
    // LDS budget of k_front are factorised by one workgroup each, one launch per level and size class (HIPMF_MID_FRONT=0: off,
    // HIPMF_MID_MMAX: rows at most, <= 192)
    bool use_mid = true;
    int32_t mid_mmax = 0;  // HIPMF_MID_MMAX: k_front (eight wavefronts, up to 64 pivots) takes the fronts with at most this many off-diagonal rows that
                           // k_front_lu does not take; off by default -- measured 7.13 ms with it (80 rows) against 7.10 without, 7.28 on tiled launches alone
    bool use_mid_lu = true; // HIPMF_MID_LU=0: fronts with at most 32 pivots take k_front / the tiled path like the others
    int32_t mid_lu_mmax = 192; // HIPMF_MID_LU_MMAX: off-diagonal rows of a k_front_lu front at most
    bool is_mid_lu(int32_t s) const;
    bool is_mid(int32_t s) const;
    int32_t diag0_min_panels = 512;         // step 0 of a level: from this many panel workgroups the first diagonal tiles get their own launch (k_diag0)
    bool level_path_ok = true;              // false: some front is too large for the level-set solves' LDS staging
    bool use_fused = true;                  // false: level-set launches (HIPMF_FUSED_SOLVE=0, or after a hand-off timeout)
    // Tiled path: fronts with at least upd_g4 (upd_g8, upd_g16) rows apply 4 (8, 16) panels per pass over the trailing matrix instead of 2:
    // the read-modify-write of the trailing matrix bounds the large fronts (HIPMF_UPD_G4 / HIPMF_UPD_G8 / HIPMF_UPD_G16)
    int32_t upd_g4 = 2048, upd_g8 = 4096, upd_g16 = 1 << 30;
    int32_t update_group(int32_t f) const { return f >= upd_g16 ? 16 : (f >= upd_g8 ? 8 : (f >= upd_g4 ? 4 : 2)); }
    int32_t small_wide_max = 7000; // a k_small_factor launch with at most this many fronts uses four wavefronts per front (HIPMF_SMALL_WIDE, 0: never)
    int32_t small_split = 28; // small fronts up to this size get their own launch per level (HIPMF_SMALL_SPLIT, 0: one launch)
    bool slab64 = false;                    // HIPMF_SOLVE_SLAB64=1: same slab shape in both solve paths (bitwise comparable)
    int32_t sf_err[2] = {0, 0};
    // fused assembly of the small fronts (k_small_factor) and zero-fill of the big ones only
    int32_t *d_sa_ptr = nullptr, *d_sa_k = nullptr;
    uint16_t *d_sa_pos = nullptr;
    ZeroTask *d_zero = nullptr;
    double *d_vs = nullptr, *d_vs2 = nullptr; // scaled values (and the mirrored ones of symmetric-lower storage)
    int32_t zero_cnt = 0;
    int32_t *d_seg_ptr = nullptr, *d_seg_idx = nullptr; // value map (set_value_map)
    int32_t *d_emap = nullptr;      // expansion of symmetric-lower values to the analysed general storage (set_expansion)
    double *d_vlow = nullptr;       // staging of the caller's lower-triangle values
    int64_t nnz_low = 0;
    std::vector<int32_t> h_emap;
    int32_t load_values(const double *values, bool on_device); // values (CSR order of the caller) -> d_vals
    double *d_vin = nullptr;
    int64_t nnz_in = 0;
    int64_t n_lists = 0;            // entries of d_lists
    SmallDesc *d_sd = nullptr;      // per position of d_lists: descriptor + entry range of a small front
    double *d_blk = nullptr, *d_work_blk = nullptr; // many-RHS blocks (allocated at the first multi-column solve)
    // further lanes of the solve driver (allocated at the first solve with more than one block): stream, block buffers, hand-off
    // words, norms; lane 0 is the solver's own stream and buffers
    static constexpr int32_t MAX_SOLVE_LANES = 4;
    struct LaneBuffers {
        void *stream = nullptr;
        double *blk = nullptr, *work = nullptr;
        int32_t *sync = nullptr;
        unsigned long long *norms = nullptr;
    };
    std::vector<LaneBuffers> extra_lanes;
    double *h_nrm = nullptr;   // pinned: norms of every lane
    double *h_stage = nullptr; // pinned: rhs | x of a single host-pointer solve
    int32_t solve_lanes = 1;   // HIPMF_SOLVE_LANES (1..4).  One since late round 4: two launches full of workgroups that wait for each other's
                               // lower-numbered tasks no longer pay (0.304 against 0.307 - 0.313 ms per right-hand side at 1000^2 with 256
                               // right-hand sides, 0.307 against 0.365 with 64) and, concurrently resident, every few runs one of them ran
                               // into a hand-off time-out (3.7 s; profiles/r04_solve_lanes.txt)
    bool solve_lanes_auto = true; // no HIPMF_SOLVE_LANES given: one lane when the factor exceeds 64 GB (solve())
    int32_t block_cols = 0;    // columns per block of the many-RHS driver once its buffers exist (8 or 16; HIPMF_BLOCK_COLS forces one)
    int32_t block_groups = 0;  // blocks ("groups", kernels_solve_fused.hpp SfGroups) a dependency-driven launch of the many-RHS driver carries once its
                               // buffers exist (1 .. SF_GMAX); block_cols * block_groups columns travel together
    bool prepare_only = false;     // solve() stops after its buffers exist (prepare_many)
    int32_t block_groups_plan = 1; // ... what initialize planned for (HIPMF_BLOCK_GROUPS, else by the size of the factor): the split-dot-product scratch is sized by it
    double block_groups_max_bytes = 1e18; // factors up to this many bytes carry SF_GMAX blocks per launch, larger ones one (HIPMF_BLOCK_GROUPS_BYTES; no limit by default: measured to pay up to config 4's 84 GB, profiles/r06_block_groups.txt)
    unsigned long long *d_norms_blk = nullptr; // norm slots of lane 0's blocked solves (block_cols * block_groups columns)
    int64_t work_blk_doubles = 0;  // stride between the columns of a blocked solve workspace: work_doubles without the tagged shadow xt
    unsigned long long *d_trace = nullptr;  // HIPMF_SF_TRACE=<file>: device-clock stamps of the upper tasks (profiling aid)
    std::vector<int32_t> sf_host, sfk_host; // (kind, front) per task (single-column lists / blocked list), kept only when tracing
    FactorInfo *d_info = nullptr;
    unsigned long long *d_scalar = nullptr; // [0] anorm bits, [4 ...] the norm slots of k_residual (lane 0)
    double *d_work = nullptr, *d_vals = nullptr, *d_xp = nullptr, *d_r = nullptr, *d_den = nullptr, *d_b = nullptr, *d_x = nullptr,
           *d_du = nullptr;
    int32_t *d_rows = nullptr, *d_rel = nullptr, *d_child = nullptr, *d_lists = nullptr, *d_tasks = nullptr;
    int32_t *d_rp = nullptr, *d_ci = nullptr, *d_arow = nullptr, *d_tptr = nullptr, *d_tidx = nullptr, *d_perm = nullptr;
    bool mid_lu_split = false;    // HIPMF_MID_LU_SPLIT=1: the k_front_lu fronts of a level in three launches by LDS class (two or four fronts per CU for the smaller ones); measured slower: 6.58 -> 6.95 ms
    bool upd_xcd = true;          // HIPMF_UPD_XCD=0: the tiles of a full trailing update in plain order (1: whole tile columns per XCD)
    bool use_ea_lds = true;       // HIPMF_EA_LDS=0: LU working blocks go back to k_zero + k_scatter + k_extend_add (read-modify-write per child) instead of k_extend_add_lds
    bool ea_lds_active() const { return use_ea_lds; }
    bool use_ea_lu = true;        // HIPMF_EA_LU=0: the first diagonal tiles go back to k_diag0 / the first panel launch
    bool ea_lu_active() const { return ea_lds_active() && use_ea_lu && !use_binv; }
    int32_t *d_ea_sc = nullptr;   // k_extend_add_lds: per task, the range of its entries of A in d_sc_k / d_sc_pos (cumulative, all levels)
    uint16_t *d_sc_pos = nullptr; // ... position inside the task's tile
    int32_t *d_sc_k = nullptr;   // scatter lists of the tiled fronts, by level: input entry k (or ~k: mirrored copy) ...
    int64_t *d_sc_at = nullptr;  // ... and the pool offset it goes to
    double *d_diag = nullptr;    // pivots in pivot order (determinant, rcond, D of the symmetric fronts)
    int32_t *d_row_blk = nullptr; // row blocks of the stream SpMV (k_spmv_stream)
    int32_t spmv_blocks = 0;
    void *ev[8] = {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};
};

} // namespace hipmf
