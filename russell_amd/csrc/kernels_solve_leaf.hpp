// kernels_solve_leaf.hpp -- the LEAVES of the assembly tree in the blocked (many-RHS) triangular solves: one wavefront, sixteen columns.
//
// The blocked instances of kernels_solve_fused.hpp run every small front as a task of its own, one wavefront per front with lane = row and
// the K columns of the block in a loop: per front and column p substitution steps of two v_readlane and one multiply-add, six dependent
// memory round trips per task.  Two thirds of the fronts of a 2D mesh are LEAVES (no children: 67 000 of the 113 000 fronts of the 1M-DOF
// Poisson matrix, ~7 pivots x ~30 rows each), and the band they sit in is half of a blocked pass pair (profiles/r03_many_rhs_kernel_stats.txt).
// A leaf depends on nothing (forward) / on ancestors that a launch of its own finds complete (backward), so here the leaves leave the
// task lists: a wavefront owns LEAF_PER_WAVE consecutive leaves and walks them with
//     lane = (column c of the block, row slice rs):  c = lane & 15, rs = lane >> 4,
// i.e. ALL sixteen columns advance with every instruction.  The front's panel is fetched flat (64 lanes x 8 bytes per piece, whatever
// its shape) into LDS; the four row slices share the rows of the front (row i belongs to slice i mod 4), a solved pivot entry travels to
// the other slices through a 16 x 16 block of LDS (one write, one read per pivot and lane), and every panel entry is an LDS broadcast
// to the sixteen lanes of a slice.  (A first version solved the p x p triangle redundantly in every slice, all p values of a column in
// registers: 256 + 60 registers, one workgroup per compute unit.)  Per column the sums run in the order of sf_fwd_small (forward: bit-identical) / in plain column
// order (backward: equal to rounding), so a column's result does not depend on what else is in its block.
// The forward kernel publishes a leaf like a task would (update vector in `work`, completion counter); the backward kernel runs after the
// dependency-driven launches of the pass.
#pragma once
#include "kernels_common.hpp"

namespace hipmf {

constexpr int LEAF_KC = 16;         // columns a wavefront carries (blocks of 8 use the first nk)
constexpr int LEAF_PMAX = 16;       // pivots of a leaf at most
constexpr int LEAF_MMAX = 48;       // off-diagonal rows at most
constexpr int LEAF_PANEL = 768;     // doubles of panel at most (12 flat pieces of 64)
#ifndef HIPMF_LEAF_PER_WAVE
#define HIPMF_LEAF_PER_WAVE 8
#endif
#ifndef HIPMF_LEAF_WAVES
#define HIPMF_LEAF_WAVES 4
#endif
constexpr int LEAF_PER_WAVE = HIPMF_LEAF_PER_WAVE; // consecutive leaves per wavefront
constexpr int LEAF_WAVES = HIPMF_LEAF_WAVES;       // wavefronts per workgroup

struct LeafRec { // 48 bytes: six 8-byte words, fetched by six lanes
    int64_t off;    // forward: pool offset of the f x f block (its first p columns = [L11; L21], stride f); backward: of the p x f rows of U (stride p)
    int64_t woff;   // offset of the front's vector in a solve workspace
    int64_t rowptr; // offset of its row structure (global row numbers of the m off-diagonal rows)
    int32_t first, p;
    int32_t m, s;   // s: front number (completion counter)
    int64_t pad;
};
static_assert(sizeof(LeafRec) == 48, "six words");

// LDS of one wave (doubles): [ panel | pivot block 16 x 17 | zero | reciprocals 16 | ints | t 16 x 16 (backward) | update block 48 x 16 (backward) ]
constexpr int LEAF_OFF_XB = LEAF_PANEL, LEAF_XB_LD = 17, LEAF_OFF_Z = LEAF_OFF_XB + 16 * LEAF_XB_LD, LEAF_OFF_INV = LEAF_OFF_Z + 2;
constexpr int LEAF_OFF_I = LEAF_OFF_INV + 16, LEAF_OFF_T = LEAF_OFF_I + 32, LEAF_OFF_X2 = LEAF_OFF_T + 16 * 16, LEAF_LDS = LEAF_OFF_X2 + LEAF_MMAX * 16;
constexpr int LEAF_LDS_FWD = LEAF_OFF_X2; // the forward kernel has no update block: 10.8 KB per wave (backward: 16.9 KB)

// entry (row r, column c) of the block of vectors (column-major, column stride xstr).  A row-major block -- the sixteen columns of a row in
// one 128-byte line, for the gathers of the backward pass -- was measured and lost: with lane = row elsewhere in the blocked kernels every
// load / store instruction then touches 64 lines instead of 4 (0.304 against 0.272 ms per right-hand side, profiles/r05_rejected_experiments.txt)
__device__ __forceinline__ int64_t leaf_x(int64_t xstr, int64_t r, int c) { return (int64_t)c * xstr + r; }

__device__ __forceinline__ void leaf_rec(const LeafRec *__restrict__ recs, int i, int lane, int64_t &off, int64_t &woff, int64_t &rowptr, int &first, int &p,
                                         int &m, int &s) {
    const long long w = reinterpret_cast<const long long *>(recs + i)[lane < 6 ? lane : 0];
    off = wave_bcast_i64(w, 0), woff = wave_bcast_i64(w, 1), rowptr = wave_bcast_i64(w, 2);
    const long long fp = wave_bcast_i64(w, 3), ms = wave_bcast_i64(w, 4);
    first = (int)(unsigned)(unsigned long long)fp, p = (int)(unsigned)((unsigned long long)fp >> 32);
    m = (int)(unsigned)(unsigned long long)ms, s = (int)(unsigned)((unsigned long long)ms >> 32);
}

// forward: y = L11^{-1} (P b1), u = -L21 y.  Lane (c, rs) holds rows rs, rs + 4, rs + 8, rs + 12 of the pivot block for column c.
__device__ __forceinline__ void leaf_fwd_body(double *L, int lane, int f, int p, int first, int64_t woff, double *xp, int64_t xstr, double *work,
                                              int64_t wstr, int nk) {
    const int c = lane & 15, rs = lane >> 4;
    const double *Xb = L + LEAF_OFF_XB;
    double *Yb = L + LEAF_OFF_T;
    const int32_t *LPi = reinterpret_cast<const int32_t *>(L + LEAF_OFF_I);
    double y[4];
#pragma unroll
    for (int q = 0; q < 4; q++) {
        const int i = rs + 4 * q;
        y[q] = (i < p) ? Xb[LPi[i] * LEAF_XB_LD + c] : 0.0; // (row interchanges inside the pivot block)
    }
#pragma unroll
    for (int jq = 0; jq < 4; jq++) {
        if (4 * jq < p) { // (wave-uniform)
#pragma unroll
            for (int js = 0; js < 4; js++) {
                const int j = 4 * jq + js;
                if (rs == js) Yb[j * 16 + c] = y[jq]; // y_j is final: steps 0 .. j - 1 have been applied to it
                wave_sync();
                const double yj = Yb[j * 16 + c];
#pragma unroll
                for (int q = 0; q < 4; q++) {
                    const int i = rs + 4 * q;
                    const double a = L[(i > j && i < p && j < p) ? i + j * f : LEAF_OFF_Z]; // (zero: row i is not below pivot j of this front)
                    y[q] -= a * yj;
                }
            }
        }
    }
    const bool live = c < nk;
    // (plain stores: whoever reads these entries is a task of a LATER launch)
#pragma unroll
    for (int q = 0; q < 4; q++)
        if (live && rs + 4 * q < p) xp[leaf_x(xstr, first + rs + 4 * q, c)] = y[q];
    double *wc = work + (int64_t)(live ? c : 0) * wstr + woff;
#pragma unroll 1
    for (int r = p + rs; r < f; r += 4) { // the lane's off-diagonal rows: u_r = 0 - sum_j l_rj y_j, j ascending (sf_fwd_small's order)
        double u = 0.0;
#pragma unroll 4
        for (int j = 0; j < p; j++) u -= L[r + j * f] * Yb[j * 16 + c];
        if (live) wc[r] = u;
    }
}

// What a wavefront fetches for one leaf (everything requested at once; clamped addresses: unconditional loads)
struct LeafLoads {
    double pc[LEAF_PANEL / 64]; // the panel as flat pieces of 64 doubles
    double xb[4];               // rows rs, rs + 4, ... of the pivot block of column c
    int32_t lp;                 // forward: the interchange of pivot row `lane`
};
struct LeafInfo {
    int64_t off, woff, rowptr;
    int first, p, m, s;
};
__device__ __forceinline__ void leaf_issue(LeafLoads &D, const LeafInfo &R, int lane, const double *__restrict__ pool, const int32_t *__restrict__ lperm,
                                           const double *xp, int64_t xstr, int nk) {
    const int c = lane & 15, rs = lane >> 4;
    const int np = ((R.p + R.m) * R.p + 63) >> 6;
    const double *src = pool + R.off;
#pragma unroll
    for (int q = 0; q < LEAF_PANEL / 64; q++) D.pc[q] = src[(q < np ? 64 * q : 0) + lane];
    const int cc = c < nk ? c : 0;
#pragma unroll
    for (int q = 0; q < 4; q++) D.xb[q] = xp[leaf_x(xstr, R.first + (rs + 4 * q < R.p ? rs + 4 * q : 0), cc)];
    D.lp = lperm ? lperm[R.first + (lane < R.p ? lane : 0)] : 0;
}
__device__ __forceinline__ void leaf_park(double *L, const LeafLoads &D, const LeafInfo &R, int lane) {
    const int c = lane & 15, rs = lane >> 4;
    const int np = ((R.p + R.m) * R.p + 63) >> 6;
#pragma unroll
    for (int q = 0; q < LEAF_PANEL / 64; q++)
        if (q < np) L[64 * q + lane] = D.pc[q];
#pragma unroll
    for (int q = 0; q < 4; q++)
        if (rs + 4 * q < R.p) L[LEAF_OFF_XB + (rs + 4 * q) * LEAF_XB_LD + c] = D.xb[q];
    if (lane < R.p) reinterpret_cast<int32_t *>(L + LEAF_OFF_I)[lane] = D.lp;
}

// The leaves of a wavefront as a software pipeline: while leaf i is computed out of LDS, the record of leaf i + 1 has arrived and its
// panel, pivot block and interchanges are in flight in registers (a leaf is a chain record -> loads -> LDS -> substitution -> stores of
// ~9 us end to end, of which the substitution is a third: unpipelined, the kernel was bound by that chain, profiles/r05_rejected_experiments.txt).
__global__ void __launch_bounds__(64 * LEAF_WAVES) k_leaf_fwd(const LeafRec *__restrict__ recs, int nleaf, const double *__restrict__ pool,
                                                              const int32_t *__restrict__ lperm, double *xp, int64_t xstr, double *work, int64_t wstr,
                                                              int *sync, int nk, uint32_t gmask) {
    __shared__ __attribute__((aligned(16))) double lds[LEAF_WAVES][LEAF_LDS_FWD];
    const int lane = threadIdx.x & 63, wave = wave_uniform(threadIdx.x >> 6);
    double *L = lds[wave];
    (void)sync;
    {
        // block groups (kernels_solve_fused.hpp, SfGroups): blockIdx.y = group of sixteen columns; nothing in this kernel waits, any order is fine
        const int grp = blockIdx.y;
        if (!((gmask >> grp) & 1u)) return;
        xp += (int64_t)grp * 16 * xstr, work += (int64_t)grp * 16 * wstr;
        nk = nk - 16 * grp < 16 ? nk - 16 * grp : 16;
        if (nk <= 0) return;
    }
    const int i0 = (blockIdx.x * LEAF_WAVES + wave) * LEAF_PER_WAVE;
    if (i0 >= nleaf) return;
    const int i1 = i0 + LEAF_PER_WAVE < nleaf ? i0 + LEAF_PER_WAVE : nleaf;
    L[LEAF_OFF_Z + (lane & 1)] = 0.0;
    LeafInfo R, Rn;
    LeafLoads D, Dn;
    leaf_rec(recs, i0, lane, R.off, R.woff, R.rowptr, R.first, R.p, R.m, R.s);
    leaf_issue(D, R, lane, pool, lperm, xp, xstr, nk);
#pragma unroll 1
    for (int i = i0; i < i1; i++) {
        const int in = i + 1 < i1 ? i + 1 : i; // (the last leaf fetches itself once more: unconditional loads, results dropped)
        leaf_rec(recs, in, lane, Rn.off, Rn.woff, Rn.rowptr, Rn.first, Rn.p, Rn.m, Rn.s);
        wave_sync(); // (the previous leaf's LDS reads are over)
        leaf_park(L, D, R, lane);
        leaf_issue(Dn, Rn, lane, pool, lperm, xp, xstr, nk); // in flight while this leaf is computed
        wave_sync();
        leaf_fwd_body(L, lane, R.p + R.m, R.p, R.first, R.woff, xp, xstr, work, wstr, nk);
        // (the leaf's parent is a task of a LATER launch on the same stream: it finds the update vector complete, and the task list of
        //  the blocked instances expects no completion count from a leaf -- no drain, no counter)
        R = Rn;
        D = Dn;
    }
}

// backward: x1 = U11^{-1} (y1 - U12 x2); rows of U as a p x f block with stride p.  Lane (c, rs) holds pivot rows rs, rs + 4, ... of column c.
__device__ __forceinline__ void leaf_bwd_body(double *L, int lane, int p, int m, int first, double *xp, int64_t xstr, int nk) {
    const int c = lane & 15, rs = lane >> 4;
    const double *Xb = L + LEAF_OFF_XB, *X2 = L + LEAF_OFF_X2, *INV = L + LEAF_OFF_INV;
    double *Yb = L + LEAF_OFF_T;
    double v[4];
    // t_i = y1_i - sum_j u_i(p + j) x2_j for the lane's pivot rows
#pragma unroll
    for (int q = 0; q < 4; q++) {
        const int i = rs + 4 * q;
        double acc = 0.0;
        if (i < p) { // (uniform over the sixteen lanes of a row slice)
            const double *Ui = L + i + p * p;
#pragma unroll 4
            for (int j = 0; j < m; j++) acc += Ui[j * p] * X2[j * 16 + c];
        }
        v[q] = (i < p) ? Xb[i * LEAF_XB_LD + c] - acc : 0.0;
    }
    // columns from right to left: x_j = t_j / u_jj, t_i -= u_ij x_j for i < j
#pragma unroll
    for (int jq = 3; jq >= 0; jq--) {
        if (4 * jq < p) { // (wave-uniform)
#pragma unroll
            for (int js = 3; js >= 0; js--) {
                const int j = 4 * jq + js;
                if (rs == js) {
                    v[jq] *= (j < p) ? INV[j] : 0.0;
                    Yb[j * 16 + c] = v[jq];
                }
                wave_sync();
                const double xj = Yb[j * 16 + c];
#pragma unroll
                for (int q = 0; q < 4; q++) {
                    const int i = rs + 4 * q;
                    const double a = L[(i < j && j < p) ? i + j * p : LEAF_OFF_Z];
                    v[q] -= a * xj;
                }
            }
        }
    }
    const bool live = c < nk;
#pragma unroll
    for (int q = 0; q < 4; q++)
        if (live && rs + 4 * q < p) xp[leaf_x(xstr, first + rs + 4 * q, c)] = v[q];
}

// Backward: the same pipeline; the solved entries of the ancestors (x2) are a gather through the row numbers, i.e. one more dependent
// round trip: the row numbers of leaf i + 1 travel with its panel, the gather of leaf i is requested at the head of its iteration.
__global__ void __launch_bounds__(64 * LEAF_WAVES) k_leaf_bwd(const LeafRec *__restrict__ recs, int nleaf, const double *__restrict__ pool,
                                                              const int32_t *__restrict__ rows, double *xp, int64_t xstr, int nk, uint32_t gmask) {
    __shared__ __attribute__((aligned(16))) double lds[LEAF_WAVES][LEAF_LDS];
    const int lane = threadIdx.x & 63, wave = wave_uniform(threadIdx.x >> 6);
    double *L = lds[wave];
    {
        const int grp = blockIdx.y; // (block groups: see k_leaf_fwd)
        if (!((gmask >> grp) & 1u)) return;
        xp += (int64_t)grp * 16 * xstr;
        nk = nk - 16 * grp < 16 ? nk - 16 * grp : 16;
        if (nk <= 0) return;
    }
    const int i0 = (blockIdx.x * LEAF_WAVES + wave) * LEAF_PER_WAVE;
    if (i0 >= nleaf) return;
    const int i1 = i0 + LEAF_PER_WAVE < nleaf ? i0 + LEAF_PER_WAVE : nleaf;
    const int c = lane & 15, rs = lane >> 4;
    L[LEAF_OFF_Z + (lane & 1)] = 0.0;
    const int cc = c < nk ? c : 0;
    LeafInfo R, Rn;
    LeafLoads D, Dn;
    int32_t ridx[LEAF_MMAX / 4], ridxn[LEAF_MMAX / 4]; // global row numbers of the lane's off-diagonal rows rs, rs + 4, ...
    leaf_rec(recs, i0, lane, R.off, R.woff, R.rowptr, R.first, R.p, R.m, R.s);
    leaf_issue(D, R, lane, pool, nullptr, xp, xstr, nk);
#pragma unroll
    for (int q = 0; q < LEAF_MMAX / 4; q++) ridx[q] = rs + 4 * q < R.m ? rows[R.rowptr + rs + 4 * q] : 0; // (m = 0: nothing to read at rowptr -- it may be the END of `rows`; row 0 of x is always there)
#pragma unroll 1
    for (int i = i0; i < i1; i++) {
        const int in = i + 1 < i1 ? i + 1 : i;
        leaf_rec(recs, in, lane, Rn.off, Rn.woff, Rn.rowptr, Rn.first, Rn.p, Rn.m, Rn.s);
        // x2 of this leaf: every ancestor is complete (earlier launches of the pass)
        double x2[LEAF_MMAX / 4];
#pragma unroll
        for (int q = 0; q < LEAF_MMAX / 4; q++) x2[q] = xp[leaf_x(xstr, ridx[q], cc)];
        wave_sync();
        leaf_park(L, D, R, lane);
        leaf_issue(Dn, Rn, lane, pool, nullptr, xp, xstr, nk);
#pragma unroll
        for (int q = 0; q < LEAF_MMAX / 4; q++) ridxn[q] = rs + 4 * q < Rn.m ? rows[Rn.rowptr + rs + 4 * q] : 0;
#pragma unroll
        for (int q = 0; q < LEAF_MMAX / 4; q++)
            if (rs + 4 * q < R.m) L[LEAF_OFF_X2 + (rs + 4 * q) * 16 + c] = x2[q];
        wave_sync();
        if (lane < R.p) L[LEAF_OFF_INV + lane] = 1.0 / L[lane + lane * R.p];
        wave_sync();
        leaf_bwd_body(L, lane, R.p, R.m, R.first, xp, xstr, nk);
        R = Rn;
        D = Dn;
#pragma unroll
        for (int q = 0; q < LEAF_MMAX / 4; q++) ridx[q] = ridxn[q];
    }
}

} // namespace hipmf
