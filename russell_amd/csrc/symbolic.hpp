// symbolic.hpp -- host-side symbolic analysis for the MI355X multifrontal LU backend.
//
// This is the "initialize" phase of the solver boundary (the role umfpack_di_symbolic plays at
// /root/reference/russell_sparse/c_code/interface_umfpack.c:109 and cudssExecute(ANALYSIS) plays
// at interface_cudss.cu:361): fill-reducing ordering, elimination tree, supernode partition,
// frontal-matrix index sets, assembly maps and the level schedule the HIP kernels run.
// Everything here is integer graph work on the host; it is deterministic (no hashing, no
// threads, ties broken by vertex number) so the permutation is reproducible bit for bit.
#pragma once
#include <atomic>
#include <cstdint>
#include <memory>
#include <utility>
#include <vector>

namespace hipmf {

// vector whose resize() leaves trivially constructible elements uninitialised: arrays of hundreds of megabytes (row structures, task
// lists of a 3D problem) are sized once and filled by host threads, which then also take the first-touch page faults
template <class T> struct NoInitAlloc : std::allocator<T> {
    template <class U> struct rebind {
        using other = NoInitAlloc<U>;
    };
    template <class U, class... A> void construct(U *p, A &&...a) {
        if constexpr (sizeof...(A) == 0) ::new ((void *)p) U;
        else ::new ((void *)p) U(std::forward<A>(a)...);
    }
};
template <class T> using noinit_vector = std::vector<T, NoInitAlloc<T>>;

enum OrderingKind : int32_t {
    ORDERING_NESTED_DISSECTION = 0, // level-structure nested dissection + minimum degree on the leaves
    ORDERING_NATURAL = 1,           // identity (Ordering::No in the reference's enum)
    ORDERING_MIN_DEGREE = 2,        // approximate minimum degree on A + A^T (Ordering::Amd / Amf / Qamd)
    ORDERING_BEST = 3,              // both of the above, the one with fewer factorisation flops (Ordering::Best)
};

struct SymbolicOptions {
    int32_t ordering = ORDERING_NESTED_DISSECTION;
    int32_t nd_leaf = 64;         // leaf regions are ordered by bitset minimum degree (<= 64 vertices)
    bool dense_leaves = false;    // every nested-dissection leaf region becomes one dense supernode
    int32_t nd_threads = 0;       // host threads of the nested dissection (0: min(16, hardware threads)); the result does not depend on it
    int32_t parallel_min_n = 200000; // below this order the row structures are built serially (thread start-up); tests set 0 (HIPMF_PAR_MIN)
    int32_t parallel_chunk_min = 4096; // smallest size bound of the subtrees handed to host threads (tests: small, so that small matrices split)
    double dense_row_factor = 10.0; // vertices of degree > max(32, min(factor sqrt(n), 4 factor x average degree)) are ordered last (0: never)
    int32_t dense_n = 32;         // n <= dense_n: one dense front, i.e. LU with full partial pivoting
    int32_t split_pivots = 4096;  // supernodes with more pivots are split into a chain of supernodes (0: never); see symbolic.cpp
    double pool_limit_bytes = 0.0; // analyse gives up (-40) when the fronts would need more than this (0: no limit)
    const std::atomic<double> *pool_limit_live = nullptr; // the same figure when it is produced by another thread while analyse runs (read at the tests; 0: not known yet)
    int32_t augment_above = 64;   // fronts with f > this take the tiled, augmented path (must equal kernels_common.hpp SMALL_F)
    bool symmetric_ldlt = false;  // the big fronts are factorised as L D L^T (symmetric-lower input): no E' panels
    bool pair_blocks = false;     // n even, rows / columns 2 k and 2 k + 1 belong together (real and imaginary part of complex unknown k,
                                  // interface_complex_hipmf.cpp): the ordering runs on the graph of the pairs, a pair stays adjacent (2 k' , 2 k' + 1
                                  // in the permuted numbering) and inside one supernode, every front has an even number of pivots and of rows
    // relaxed amalgamation (a supernode absorbs the child that ends right before it): merged supernodes of up to relax_ncol[0] columns always,
    // up to [1] / [2] columns when the share of explicit zeros stays below relax_zeros[0] / [1]; larger ones below relax_zeros[2] AND only
    // into fronts of at least relax_big_front rows (0: any).  Late round 4 (profiles/r04_relax_sweep.txt): every pivot merged into a
    // front near the root is one more step of a sequential chain, which is what the 2D factorisation is bound by -- merging large
    // supernodes pays only where the front is big enough to be bound by arithmetic (3D).  48 / any -> 64 / 2 048: 1000^2 LU 6.55 -> 6.19 ms
    // (292 -> 263 launches), L D L^T 5.65 -> 5.29 ms, 100^3 136 -> 140 ms.
    int32_t relax_ncol[3] = {4, 16, 64};
    double relax_zeros[3] = {0.8, 0.1, 0.05};
    int32_t relax_big_front = 2048;
};

struct Symbolic {
    int32_t n = 0;
    int64_t nnz_a = 0;          // entries of the input CSR
    bool sym_lower = false;     // input holds the lower triangle of a symmetric matrix
    bool best_chose_min_degree = false; // ORDERING_BEST: the minimum degree won

    std::vector<int32_t> perm;  // perm[new] = old   (applied to rows and columns)
    std::vector<int32_t> pinv;  // pinv[old] = new

    int32_t nsuper = 0;
    std::vector<int32_t> sn_first;   // nsuper+1: first permuted column of each supernode
    std::vector<int32_t> sn_of;      // n: supernode of a permuted column
    std::vector<int64_t> sn_rowptr;  // nsuper+1
    noinit_vector<int32_t> sn_rows;  // off-diagonal row structure (permuted indices, ascending)
    std::vector<int32_t> sn_parent;  // parent supernode or -1
    std::vector<int32_t> sn_level;   // 0 = leaves
    int32_t nlevels = 0;
    std::vector<int32_t> level_ptr;  // nlevels+1
    std::vector<int32_t> level_sn;   // supernodes grouped by level
    std::vector<int32_t> child_ptr;  // nsuper+1
    std::vector<int32_t> child_idx;  // children of each supernode, ascending
    noinit_vector<int32_t> rel;      // aligned with sn_rows: position of the row in the PARENT's front
    // Pool layout (offsets in doubles into ONE device allocation): [0, persist_doubles) holds what the solves need (the small
    // fronts' f x f blocks, the big fronts' E / E' panels); [persist_doubles, persist_doubles + temp_doubles) is the arena of the
    // big fronts' f x f working blocks, whose storage is re-used once the parent has consumed the contribution block.
    std::vector<int64_t> front_off;  // nsuper: offset of the f x f block (small: persistent; big: arena)
    std::vector<int32_t> front_ld;   // nsuper: its leading dimension = f
    std::vector<int64_t> e_off;      // nsuper: big fronts: E (f x p, ld f), else -1
    std::vector<int64_t> ep_off;     // nsuper: big fronts in LU mode: E' (p x f, stride front_ldp), else -1
    std::vector<int32_t> front_ldp;  // nsuper: column stride of E' (big fronts: p, rounded up to 128-byte lines above 64 pivots; small fronts: p = stride of the packed rows of U)
    int64_t persist_doubles = 0, temp_doubles = 0;
    bool sym_mode = false;           // big fronts are factorised as L D L^T (lower triangle only)
    std::vector<int64_t> amap;       // nnz_a: pool offset every input entry is added to
    std::vector<int64_t> amap2;      // nnz_a when sym_lower: mirrored position (-1 on the diagonal)
    std::vector<int32_t> amap_sn;    // nnz_a: the supernode whose front receives the entry

    // statistics
    int64_t nnz_l = 0;   // strictly lower entries of L (stored, incl. amalgamation padding)
    int64_t nnz_u = 0;   // upper entries of U incl. diagonal
    double flops = 0.0;  // sum over fronts of 2/3 p^3 + 2 p^2 m + 2 p m^2
    double flops_gemm = 0.0;
    int32_t max_front = 0;
    int32_t max_pivots = 0;
    double seconds_ordering = 0.0, seconds_total = 0.0;
    double pool_estimate_bytes = 0.0; // from the column counts, before the row structures exist
    // host phases of analyse(): graph, ordering, etree + postorder, column counts + supernodes, row structures + relative indices,
    // levels + layout, assembly map
    double seconds_phase[7] = {0, 0, 0, 0, 0, 0, 0};

    inline int32_t npiv(int32_t s) const { return sn_first[s + 1] - sn_first[s]; }
    inline int32_t nrow(int32_t s) const { return (int32_t)(sn_rowptr[s + 1] - sn_rowptr[s]); }
    inline int32_t fsize(int32_t s) const { return npiv(s) + nrow(s); }
};

// Structure check of a 0-based CSR every entry point runs BEFORE anything reads through the indices: 0 = valid,
// -1 row pointers (rp[0] != 0 or decreasing), -2 column index out of range, -3 column indices of a row not strictly
// increasing (unsorted or duplicate entries: the assembly map sends every entry to its own slot of a front).
int validate_csr(int32_t n, const int32_t *row_ptr, const int32_t *col_idx);

// Analyse the n x n matrix given as 0-based CSR (the layout solver_cudss_initialize receives,
// interface_cudss.cu:190-203).  Returns 0 on success, a negative number on invalid input.
int analyse(int32_t n, const int32_t *row_ptr, const int32_t *col_idx, bool sym_lower,
            const SymbolicOptions &opt, Symbolic &out);

} // namespace hipmf
