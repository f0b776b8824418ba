// kernels_common.hpp -- constants, descriptors and small device helpers shared by the HIP kernels.
//
// Data layout in HBM
//   pool   ONE device allocation in two parts:  [ persistent factor | temporary arena ].
//          f = p + m (p pivot columns, m off-diagonal rows) for every supernode s.
//          * small fronts (f <= SMALL_F): a column-major f x f block at pool + off in the PERSISTENT part, ld = f.  After
//            factorisation the first p columns hold L11\U11 and L21, rows 0..p of the other columns hold U12, the trailing
//            m x m block is the contribution block the parent consumes (extend-add).  The rows of U are stored once more, packed:
//            [U11 | U12] as a p x f block with stride p at pool + epoff (PERSISTENT; fronts with m = 0 have none): the backward
//            solve reads that block only.
//          * big fronts (f > SMALL_F) are factorised AUGMENTED: the partial LU runs on the (f+p) x (f+p) matrix
//            [F  Ic; Ir  0] (Ic = [I; 0], Ir = [I, 0]) and leaves
//                E  = [inv(L11) P ; -L21 inv(L11) P]      f x p, column-major, stride ld, at pool + eoff   (PERSISTENT)
//                E' = [inv(U11) , -inv(U11) U12]           p x f, column-major ld = p, at pool + epoff  (PERSISTENT)
//            which turn the triangular solves of a big supernode into dependency-free GEMVs.  The front itself, F (f x f, ld = f,
//            at pool + off), lives in the TEMPORARY arena: its storage is handed to other fronts once the parent has consumed the
//            contribution block (static lifetime plan made at initialize, symbolic.cpp).  Index (r, c) of the augmented matrix:
//            r < f, c < f -> F;  r < f, c >= f -> E(r, c - f);  r >= f, c < f -> E'(r - f, c)   (AugView below).
//          * SYMMETRIC mode (FD_SYM; general_symmetric / positive_definite input): a big front is factorised as L D L^T without
//            interchanges, only the lower triangle of F is assembled, updated and read; E' does not exist: the backward solve
//            applies E^T:  x1 = E^T [D^{-1} y1; x2].
//   diag   n doubles: the pivots (diagonal of U, resp. D) in pivot order, written by whoever factorises a diagonal block.
//   lperm  n int32: for pivot row r of front s, the front-local row that partial pivoting moved there
//          (search restricted to the pivot block for small fronts, to the 32-row diagonal tile for big ones).
//   work   one f-vector per front for the multifrontal forward/backward substitutions:
//          work[0..p) = y1 (big fronts), work[p..f) = update vector u handed to the parent.
#pragma once
#include <hipmf_device_rt.h>

#include <cstdint>

namespace hipmf {

constexpr int NB = 32;         // pivot-block width of the tiled path
constexpr int PANEL_T = 128;   // rows (L) / columns (U) handled by one panel workgroup
constexpr int UPD_T = 64;      // trailing-update tile edge (one 256-thread workgroup, 4 waves of 32x32)
constexpr int SMALL_F = 64;    // fronts with f <= SMALL_F are factorised by one wavefront in LDS
constexpr int US_LD = 34;      // LDS leading dimension of the U slice in the update kernel (bank-conflict free, see k_update)
constexpr int SOLVE_SLAB_WIDE = 32; // rows per 1024-thread workgroup on the levels near the root (32 rows x 32 column groups)
constexpr int SOLVE_SLAB = 64; // rows of a solve panel per 256-thread workgroup (64 rows x 4 column groups)

// Optional device-clock stamps for kernel tuning (builds with -DHIPMF_STAMPS only; tools/stamps.py reads them through
// hipmf_debug_read_stamps): HIPMF_STAMP(row, i) stores the 100 MHz clock into row `row`, slot i (16 slots per row, 1024 rows).
#ifdef HIPMF_STAMPS
__device__ unsigned long long hipmf_stamps[16 * 1024];
#define HIPMF_STAMP(row, i)                                                                  \
    do {                                                                                     \
        if (threadIdx.x == 0 && (row) >= 0 && (row) < 1024) hipmf_stamps[(row) * 16 + (i)] = dev_clock(); \
    } while (0)
#define HIPMF_STAMP_VAL(row, i, v)                                                           \
    do {                                                                                     \
        if (threadIdx.x == 0 && (row) >= 0 && (row) < 1024) hipmf_stamps[(row) * 16 + (i)] = (unsigned long long)(v); \
    } while (0)
#else
#define HIPMF_STAMP(row, i) ((void)0)
#define HIPMF_STAMP_VAL(row, i, v) ((void)0)
#endif

struct FrontDesc {
    int64_t off;    // offset of the front in the pool (doubles)
    int64_t rowptr; // offset of the row structure / relative indices
    int64_t woff;   // offset of the f-vector in the solve workspace
    int32_t p, m;   // pivots, off-diagonal rows
    int32_t first;  // first permuted column
    int32_t child_begin, child_end;
    int32_t parent;
    int32_t ld;     // column stride of the f x f block at `off` and of E (>= f; small fronts: f)
    int32_t ugroup; // tiled path: 32-pivot panels per read-modify-write pass over the trailing matrix (2, 4, 8 or 16)
    int64_t eoff;   // big fronts: offset of E (f x p, stride ld); -1 for small fronts
    int64_t epoff;  // big fronts, LU mode: offset of E' (p x f, stride ldp); small fronts with m > 0: packed rows of U (p x f, stride p); else -1
    int32_t flags;  // FD_BIG | FD_SYM
    int32_t ldp;    // column stride of E' (big fronts: >= p, a multiple of 16 above 64 pivots) / of the packed rows of U (small fronts: p)
};
constexpr int32_t FD_BIG = 1; // tiled path (f > SMALL_F)
constexpr int32_t FD_SYM = 2; // big front factorised as L D L^T: only the lower triangle of F (and of its contribution block) is valid
constexpr int32_t FD_DENSE_TOP = 4; // big front factorised by k_front (kernels_factor_front.hpp): E = [inv(F11); -F21 inv(F11)], E' = [I | -inv(F11) F12] --
                                    // the pivot rows of E are a full p x p block (the tiled kernels leave inv(L11) P, block lower triangular)

// Every field of a (workgroup-uniform) descriptor is requested in ONE scalar-memory round trip: without this the compiler fetches
// the fields an early-exit test needs first and the rest after the branch -- one more dependent round trip (~1.5 us) at the head
// of every tiled kernel, which are latency chains.
__device__ __forceinline__ void fd_resident(const FrontDesc &fd) {
    HIPMF_KEEP_SCALAR(fd.off);
    HIPMF_KEEP_SCALAR(fd.eoff);
    HIPMF_KEEP_SCALAR(fd.epoff);
    HIPMF_KEEP_SCALAR(fd.p);
    HIPMF_KEEP_SCALAR(fd.m);
    HIPMF_KEEP_SCALAR(fd.first);
    HIPMF_KEEP_SCALAR(fd.ld);
    HIPMF_KEEP_SCALAR(fd.ugroup);
}

// The augmented index space of a big front (see the layout note at the top of this file).
struct AugView {
    double *F;    // (r, c), r < f, c < f          at F[r + c ld]
    double *Esh;  // (r, c), r < f, c >= f         at Esh[r + c ld]    (Esh = E - f ld)
    double *Epsh; // (r, c), r >= f, c < f         at Epsh[r + c ps]   (Epsh = E' - f)
    int64_t ld;   // column stride of F and of E (>= f)
    int32_t f, p;
    int32_t ps;   // column stride of E' (>= p)
    __device__ __forceinline__ double *at(int r, int c) const {
        if (r >= f) return Epsh + r + (int64_t)c * ps;
        return (c >= f ? Esh : F) + r + (int64_t)c * ld;
    }
};
__device__ __forceinline__ AugView aug_view(const FrontDesc &fd, double *pool) {
    AugView v;
    v.f = fd.p + fd.m, v.p = fd.p;
    v.ps = fd.ldp;
    v.ld = fd.ld;
    v.F = pool + fd.off;
    v.Esh = pool + fd.eoff - (int64_t)v.f * v.ld;
    v.Epsh = pool + fd.epoff - v.f;
    return v;
}

// A small front's descriptor and the range of its entry list, in LAUNCH order (index = position in the level's list): the
// workgroup reads one record instead of list -> FD / sa_ptr, one dependent memory round trip less at the head of its chain.
struct SmallDesc {
    FrontDesc fd;
    int32_t e0, e1; // range in sa_k / sa_pos
};

struct EaTask {
    int64_t f_off;                    // pool offset of the parent front
    int32_t ld;                       // its leading dimension
    int32_t piece_begin, piece_end;   // the EaRange pieces (children in ascending order) that hit this tile of the parent
    int32_t sym;                      // parent factorised as L D L^T: only entries on or below its diagonal are added
    int32_t c0, r0, nc, nr;           // the tile: columns [c0, c0 + nc), rows [r0, r0 + nr) of the parent (k_extend_add_lds writes all of it)
    int32_t lu_slot, lu_first, lu_nb; // k_extend_add_lds, first tile of a tiled LU front: its slot in the level (< 0: none), first pivot, size of the first diagonal tile
    int32_t pad;
};

// One child's contribution block restricted to one tile of the parent: everything the kernel needs in one load.
struct EaRange {
    int64_t cb_off;             // pool offset of the child's contribution block (entry p, p of its front)
    int64_t rel_off;            // offset of the child's relative indices
    int32_t ldc;                // leading dimension of the child front
    int32_t jlo, jhi, ilo, ihi; // child entries whose relative index falls into the tile's column resp. row range
    int32_t pad;
};

struct SolveTask {
    int32_t s, r0, r1; // big front and the slab of rows its workgroup computes in the forward / backward GEMV
};

// device-side counters written by the factorisation kernels
struct FactorInfo {
    int32_t n_perturbed;  // pivots replaced by +-eps (cf. CUDSS_DATA_NPIVOTS, interface_cudss.cu:466-475)
    int32_t n_zero_pivot; // exactly-zero pivots met (singular in the UMFPACK sense, solver_umfpack.rs:492)
    int32_t n_nonfinite;  // NaN / Inf among the scaled input values (the factorisation is refused)
    int32_t n_weak_diag;  // rows whose (matched, scaled) diagonal is below 1 % of the row's largest entry (k_diag_check)
};
// Static pivoting (round 6): a pivot is REPLACED when it is below the threshold pivot_eps max|a| (cuDSS's pivot_epsilon, 1e-13 by
// default: what is smaller than that is rounding noise) -- but by +-sqrt(machine eps) max|a|, not by the threshold itself (unless the
// caller's threshold is larger).  A replacement of 1e-13 max|a| makes the trailing matrix receive updates of 1e13 times its own size: every
// digit of it is lost, the factors are no longer the LU of a matrix NEAR A and neither refinement nor the Krylov rescue (numeric.cpp)
// recovers the solution (tests/test_matrix_zoo_gpu.py, the +-1 family: forward error 1e-4 ... 1e2).  With sqrt(eps) half of the digits
// survive: the factors are the LU of A + E, |E| ~ 1e-8 |A|, rank(E) = number of replaced pivots -- the choice of SuperLU_DIST's static
// pivoting (Li & Demmel) and of PARDISO for symmetric indefinite matrices.  Pivots at or above the threshold are never touched.
__device__ __forceinline__ double pivot_replacement(double pivot_eps, double anorm) {
    const double r = 1.4901161193847656e-08; // sqrt(2^-52)
    return (pivot_eps > r ? pivot_eps : (pivot_eps > 0.0 ? r : 0.0)) * anorm; // (pivot_eps == 0: the caller asked for no replacement value)
}

// What the device allocation behind a FactorInfo pointer really holds.  zdiag (PAIRED instances of the factorisation kernels only: the
// real-equivalent form of a complex matrix, interface_complex_hipmf.cpp): the COMPLEX pivots, zdiag[2 k], zdiag[2 k + 1] = Re, Im of the
// pivot of the complex elimination step k (permuted numbering: k = first / 2 + step / 2 of the front) -- what the determinant of the
// complex matrix is the product of (the real pivots multiply to |det|^2: the phase is not in them).
struct FactorInfoExt {
    FactorInfo i;
    double *zdiag;
};
template <bool PAIRED> __device__ __forceinline__ void store_zpivot(FactorInfo *info, int at, int step, double zr, double zi) {
    if constexpr (PAIRED) {
        if ((step & 1) == 0) { // the lane whose row was the pivot row of an even step holds the pair's complex pivot
            double *zd = reinterpret_cast<const FactorInfoExt *>(info)->zdiag;
            zd[at + step] = zr, zd[at + step + 1] = zi;
        }
    }
}

__device__ __forceinline__ int lower_bound_i32(const int32_t *a, int n, int v) {
    int lo = 0, hi = n;
    while (lo < hi) {
        int mid = (lo + hi) >> 1;
        if (a[mid] < v) lo = mid + 1;
        else hi = mid;
    }
    return lo;
}

// largest a in [0, n) with pfx[a] <= g  (pfx is an exclusive prefix sum with pfx[n] = total).
// Called by whole wavefronts (g is workgroup-uniform): up to 64 slots are resolved with ONE load and a ballot
// instead of log2(n) dependent loads, which sit at the head of every tiled kernel's critical path.
__device__ __forceinline__ int find_slot(const int32_t *pfx, int n, int g) {
    if (n <= 64) {
        const int lane = threadIdx.x & 63;
        const int v = lane < n ? pfx[lane] : 0x7fffffff;
        return __popcll(__ballot(v <= g)) - 1;
    }
    int lo = 0, hi = n;
    while (hi - lo > 1) {
        int mid = (lo + hi) >> 1;
        if (pfx[mid] <= g) lo = mid;
        else hi = mid;
    }
    return lo;
}

// find_slot plus pfx[slot]: with up to 64 slots both come out of the one vector load (no second memory access); the tiled kernels
// then fetch their front's descriptor from a per-level table indexed by the slot -- two dependent round trips from the start of
// the kernel to the first load of matrix data (task prefix, descriptor), where pfx -> pfx[slot] -> list[slot] -> FD[front] took four.
__device__ __forceinline__ int find_slot_pfx(const int32_t *pfx, int n, int g, int &pfx_slot) {
    if (n <= 64) {
        const int lane = threadIdx.x & 63;
        const int v = lane < n ? pfx[lane] : 0x7fffffff;
        const int slot = wave_uniform(__popcll(__ballot(v <= g)) - 1);
        pfx_slot = wave_uniform(__shfl(v, slot));
        return slot;
    }
    int lo = 0, hi = n;
    while (hi - lo > 1) {
        int mid = (lo + hi) >> 1;
        if (pfx[mid] <= g) lo = mid;
        else hi = mid;
    }
    pfx_slot = pfx[lo];
    return lo;
}

// Slot + descriptor of the workgroup's front in the tiled kernels.  Launches with at most four fronts (the levels near the root, whose
// kernels are links of a latency chain) get the prefix words of slots 1 .. 3 as kernel ARGUMENTS (q.x, q.y, q.z; INT_MAX where there is no
// such slot): the slot comes out of scalar compares and the descriptor is requested at once -- prefix -> slot -> descriptor were two
// dependent round trips (~1.5 us each) at the head of every such kernel (root alone: 6.59 -> 6.51 ms).  (Requesting the prefix words and
// all four descriptors at once and picking afterwards was measured too: the picked descriptor no longer lives in scalar registers,
// 6.56 -> 7.45 ms.)
struct Pfx4 {
    int32_t x, y, z;
};
__device__ __forceinline__ FrontDesc load_front_pfx(const int32_t *__restrict__ pfx, const int n, const FrontDesc *__restrict__ LFD, const int g, const Pfx4 q,
                                                    int &slot, int &pfx_slot) {
    if (n <= 4 && q.x >= 0) { // (q.x < 0: no prefix arguments for this launch; fronts without tasks have the prefix of their successor: the LAST slot whose prefix is <= g, as find_slot_pfx)
        slot = g >= q.z ? 3 : (g >= q.y ? 2 : (g >= q.x ? 1 : 0));
        pfx_slot = slot == 3 ? q.z : (slot == 2 ? q.y : (slot == 1 ? q.x : 0));
        return LFD[slot];
    }
    slot = find_slot_pfx(pfx, n, g, pfx_slot);
    return LFD[slot];
}

// wave-wide arg-max of (value, index); ties resolved towards the smaller index (deterministic)
__device__ __forceinline__ void wave_argmax(double &v, int &i) {
    for (int off = 32; off > 0; off >>= 1) {
        double ov = __shfl_xor(v, off);
        int oi = __shfl_xor(i, off);
        if (ov > v || (ov == v && oi < i)) {
            v = ov;
            i = oi;
        }
    }
}

} // namespace hipmf
