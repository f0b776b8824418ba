// kernels_solve_tree.hpp -- the bottom of the assembly tree in the triangular solves: one WAVEFRONT per subtree, fed as a stream.
//
// The dependency-driven kernels (kernels_solve_fused.hpp) run every small front as a task of its own: task record -> descriptor ->
// children's descriptors -> completion flags -> relative indices + update vectors -> panel, five to six dependent memory round trips
// per front, 100 000 fronts for the 1M-DOF Poisson factor: the leaf band is bound by (resident waves) / (chain length), not by HBM.
// Here a maximal subtree of small fronts (f <= 64, bounded size; planned at initialize, numeric.cpp) belongs to ONE wavefront, which
// walks it in postorder (forward) / reverse postorder (backward):
//   * dependencies inside the subtree are satisfied by program order: no flags, no waits, no agent-scope traffic;
//   * the vectors that travel between the fronts of the subtree live in the wave's own LDS (forward: a parent's accumulator
//     b1 + sum of the children's updates, children in ascending order -- the same sums in the same order as k_fwd / sf_fwd_small;
//     backward: a front's solved vector [x1; x2], from which its children gather their x2); the subtree's part of the right-hand
//     side (its pivot columns are one contiguous range) and of the interchanges is fetched once, at the start;
//   * the factor comes in BATCHES of up to 8 fronts / 6 KB: a first version fetched one front per memory round trip (16 row loads in
//     registers, the next front requested while the current one is computed) and measured 3 us per front -- a wave never had more than
//     ~2 KB in flight.  Now the panels of a batch are read as FLAT 512-byte pieces (64 lanes x 8 bytes, whatever the
//     shape of the fronts), parked in LDS, and the fronts of the batch are computed out of LDS while the pieces of the next batch are
//     in flight in the registers that staged this one.  Each batch comes with a header (where its pieces are) and a meta block (one
//     16-word record per front + the index lists), both built at initialize: no descriptor chase at all.
// Only the root of a subtree talks to the rest of the tree: forward, it publishes its update vector in `work` and bumps its
// completion counter (the upper fronts run in a later launch on the same stream); backward, its ancestors are complete before the
// launch starts.  Arithmetic and summation order are those of sf_fwd_small / sf_bwd_small: bit-identical results.
#pragma once
#include "kernels_common.hpp"

namespace hipmf {

#ifndef HIPMF_WT_NCH
#define HIPMF_WT_NCH 12
#endif
#ifndef HIPMF_WT_X
#define HIPMF_WT_X 256
#endif
#ifndef HIPMF_WT_STACK
#define HIPMF_WT_STACK 320
#endif
#ifndef HIPMF_WT_WAVES
#define HIPMF_WT_WAVES 1 // (round 6: one wave-subtree per workgroup and a 320-word meta block: 14 080 B of LDS per wave, eleven waves per CU
                         //  instead of ten -- pass pair at C2 400.0 - 400.7 -> 389.5 - 390.1 us, 2000^2 1 456 -> 1 413, 100^3 3 118 -> 3 086;
                         //  two waves per workgroup with the same footprint: no gain; four: 423 - 426.  profiles/r06_wt_variants.txt)
#endif
constexpr int WT_NCH = HIPMF_WT_NCH;    // 512-byte pieces of factor per batch
constexpr int WT_CHUNK = 64;            // doubles per piece (64 lanes x 8 bytes; with 1 KB pieces the pass fetched 1.25x the algorithmic bytes:
                                        // a piece is read whole, and what follows a front's panel in the pool is its contribution block)
constexpr int WT_NREC = 8;              // fronts per batch at most
#ifndef HIPMF_WT_MI
#define HIPMF_WT_MI 320
#endif
constexpr int WT_MI = HIPMF_WT_MI;      // 32-bit words of a batch's meta block: WT_NREC records of 16 words, then the index lists
constexpr int WT_X = HIPMF_WT_X;        // pivots of a wave-subtree at most (multiple of 256)
constexpr int WT_STACK = HIPMF_WT_STACK; // doubles of LDS per wave for the stack of front vectors (5 levels x 64 rows)
// (measured on the 1M-DOF Poisson factor, forward + backward pass pair: 8 pieces / 512 pivots / 384 stack doubles = 20 KB of LDS per
//  wave, 8 waves per CU: 470 us; 6 / 256 / 320 = 14.8 KB, 10 waves per CU: 460 us; 4 / 256 / 320: 506 us -- fronts of more than 512
//  panel entries then stay outside the wave-subtrees; those runs used 1 KB pieces, the counts are now in 512-byte pieces)
// LDS of one wave, in doubles: [ panels | meta | x of the subtree | interchanges of the subtree | stack | gather scratch ]
constexpr int WT_OFF_M = WT_NCH * WT_CHUNK, WT_OFF_X = WT_OFF_M + WT_MI / 2, WT_OFF_LP = WT_OFF_X + WT_X, WT_OFF_ST = WT_OFF_LP + WT_X / 2;
constexpr int WT_OFF_XG = WT_OFF_ST + WT_STACK, WT_OFF_Z = WT_OFF_XG + 64, WT_LDS = WT_OFF_Z + 64; // (Z: 64 zeros)
constexpr int WT_WAVES = HIPMF_WT_WAVES; // wavefronts (= subtrees) per workgroup

struct WtRec {       // one front of a batch: 16 words at the head of the batch's meta block
    int32_t pslot;   // offset (doubles) of the front's piece(s) in the panel area: forward [L11; L21] (f x p, stride f), backward the rows of U
    int32_t pm;      // p | m << 16
    int32_t xoff;    // first pivot column - first pivot column of the subtree
    int32_t pxoff;   // parent's xoff (FIRST child: it initialises the parent's accumulator with the parent's b1)
    int32_t ppf;     // parent's p | parent's f << 16
    int32_t lds;     // (stack offset of the front's own vector + 1) | (stack offset of the parent's vector + 1) << 16; 0 = none
    int32_t flags;   // bit 0: first child of its parent; bit 1 (backward): rows of U packed with stride p (else inside the f x f block);
                     // bit 2: the front has the packed copy, i.e. zeros where the substitutions would otherwise test the lane (see wt_fwd_steps)
    int32_t relo;    // word offset, inside the meta block, of the front's index list (rel; backward root: global row numbers)
    int32_t first;   // first pivot column
    int32_t s;       // front number (completion counter of the root)
    int64_t woff;    // offset of the front's vector in the global solve workspace (root: its update vector goes there)
    int32_t pad[4];
};
static_assert(sizeof(WtRec) == 64, "16 words");

struct WtHdr {                 // one batch: 16 8-byte words, fetched by 16 lanes
    int64_t src[WT_NCH];       // pool offsets of the batch's pieces (unused ones repeat src[0])
    int64_t meta;              // word offset of the batch's meta block
    int32_t nrec, words;       // fronts of the batch; words of its meta block that hold something (records + index lists)
    int64_t pad2[16 - WT_NCH - 2];
};
static_assert(sizeof(WtHdr) == 128, "16 words");

struct WtWave {       // 256 bytes (two 128-byte lines), fetched in one round trip
    WtHdr h0;         // the header of the wave's FIRST batch (a copy): the pieces of that batch are requested as soon as the record is
                      // there -- record -> header -> pieces was one dependent round trip more at the head of every wave, and a wave-subtree
                      // is only ~8 fronts long (5 - 6 waves take turns on every slot of the device: the round trip was paid 5 - 6 times)
    int32_t b0, b1;   // batches of the wave
    int32_t xfirst;   // first pivot column of the subtree
    int32_t npiv;     // its pivot columns (<= WT_X)
    int64_t pad[14];
};
static_assert(sizeof(WtWave) == 256, "two lines");

// the loads of one batch: 8 pieces of factor, 2 of meta (all unconditional: the batch after this one is requested before this one
// is computed, and the compiler can only wait for "all but the last N loads" when N does not depend on predicates)
__device__ __forceinline__ void wt_issue(long long hw, int lane, const double *__restrict__ pool, const int32_t *__restrict__ meta,
                                         double (&pc)[WT_NCH], i32x4 (&mc)[2]) {
#pragma unroll
    for (int c = 0; c < WT_NCH; c++) pc[c] = pool[wave_bcast_i64(hw, c) + lane];
    const int64_t mo = wave_bcast_i64(hw, WT_NCH);
    // (a meta block that ends inside its first 256 words -- most do -- is not fetched beyond them: the second load re-reads the first
    //  chunk, a cache hit; still unconditional, see above)
    const int words = (int)(unsigned)((unsigned long long)wave_bcast_i64(hw, WT_NCH + 1) >> 32);
    const int second = words > 256 ? 256 : 0;
    mc[0] = ld_i32x4(meta + mo + 4 * lane);
    mc[1] = ld_i32x4(meta + mo + second + 4 * lane);
}
__device__ __forceinline__ void wt_park(double *L, int lane, const double (&pc)[WT_NCH], const i32x4 (&mc)[2]) {
#pragma unroll
    for (int c = 0; c < WT_NCH; c++) L[WT_CHUNK * c + lane] = pc[c];
    int32_t *Mi = reinterpret_cast<int32_t *>(L + WT_OFF_M);
    // (the meta area holds WT_MI words: the second chunk of 256 is stored as far as it fits -- with WT_MI < 512 an unguarded store ran into
    //  the x area behind it; found in round 6 when the footprint was cut, tools/gpu/r06_z*.sh)
    static_assert(WT_MI >= 256 && WT_MI % 4 == 0 && WT_MI <= 512, "meta block: one full chunk of 256 words, at most two");
    st_lds_i32x4(Mi + 4 * lane, mc[0]);
    if (WT_MI == 512 || 256 + 4 * lane < WT_MI) st_lds_i32x4(Mi + 256 + 4 * lane, mc[1]);
}

// ------------------------------------------------------------------ forward: one front out of LDS
// The wave-subtree kernels are bound by instruction issue (measured: ~1 500 cycles per front and SIMD, for f x p / 64 ~ 4 multiply-adds
// per lane), so the substitution is straight-line code for P = 4, 8, 12 or 16 pivots (the class of the front; columns past p come from
// a block of zeros) without per-step tests: k_small_factor leaves ZEROS on and above the diagonal of the pivot block of a front that
// has a packed copy of its rows of U, so lane i <= j multiplies by zero where sf_fwd_small tests lane > j (same bits; the sign of
// an exact zero may differ).  Fronts without the packed copy (no off-diagonal rows) take the tested form.
template <int P>
__device__ __forceinline__ double wt_fwd_steps(const double *L, int pslot, int lr, int p, int f, double v) {
    double a[P];
#pragma unroll
    for (int c = 0; c < P; c++) a[c] = L[c < p ? pslot + lr + c * f : WT_OFF_Z];
#pragma unroll
    for (int c = 0; c < P; c++) {
        const double vj = wave_bcast(v, c);
        v -= a[c] * vj;
    }
    return v;
}

__device__ __forceinline__ void wt_fwd_front(double *L, int q, int lane, double *x, double *work, int *done, bool publish) {
    const int32_t *Mi = reinterpret_cast<const int32_t *>(L + WT_OFF_M);
    const int32_t *LPi = reinterpret_cast<const int32_t *>(L + WT_OFF_LP);
    const double *X = L + WT_OFF_X;
    double *ST = L + WT_OFF_ST;
    const int ri = Mi[16 * q + (lane & 15)];
    const int pslot = wave_bcast_i32(ri, 0), pm = wave_bcast_i32(ri, 1), xoff = wave_bcast_i32(ri, 2), ldsv = wave_bcast_i32(ri, 5);
    const int flags = wave_bcast_i32(ri, 6), relo = wave_bcast_i32(ri, 7), first = wave_bcast_i32(ri, 8);
    const int p = pm & 0xffff, m = pm >> 16, f = p + m;
    const int lds_self = (ldsv & 0xffff) - 1, lds_par = (ldsv >> 16) - 1;
    const int lr = lane < f ? lane : f - 1;
    const int lp = LPi[xoff + (lane < p ? lane : 0)];
    // w = b1 (pivot rows) + the children's updates: complete in the front's own stack slot when it has children
    double v;
    if (lds_self >= 0) v = ST[lds_self + (lane < p ? lp : lr)];
    else v = (lane < p) ? X[xoff + lp] : 0.0;
    // y1 = L11^{-1} (P w1) column by column, u = w2 - L21 y1 (sf_fwd_small's arithmetic)
    if ((flags & 4) && p <= 16) { // (wave-uniform) zeros on and above the diagonal of the pivot block
        if (p <= 4) v = wt_fwd_steps<4>(L, pslot, lr, p, f, v);
        else if (p <= 8) v = wt_fwd_steps<8>(L, pslot, lr, p, f, v);
        else if (p <= 12) v = wt_fwd_steps<12>(L, pslot, lr, p, f, v);
        else v = wt_fwd_steps<16>(L, pslot, lr, p, f, v);
    } else {
        const double *Pn = L + pslot + lr;
        for (int j0 = 0; j0 < p; j0 += 8) { // eight columns at a time
            double a8[8];
#pragma unroll
            for (int c = 0; c < 8; c++) a8[c] = Pn[(j0 + c < p ? j0 + c : p - 1) * f];
#pragma unroll
            for (int c = 0; c < 8; c++) {
                const int j = j0 + c;
                if (j < p) {
                    const double vj = wave_bcast(v, j & 63);
                    if (lane > j && lane < f) v -= a8[c] * vj;
                }
            }
        }
    }
    if (lane < p) x[first + lane] = v;
    if (lds_par >= 0) {
        double *ap = ST + lds_par;
        if (flags & 1) {
            const int pxoff = wave_bcast_i32(ri, 3), ppf = wave_bcast_i32(ri, 4);
            if (lane < (ppf >> 16)) ap[lane] = (lane < (ppf & 0xffff)) ? X[pxoff + lane] : 0.0;
            wave_sync();
        }
        if (lane >= p && lane < f) ap[Mi[relo + lane - p]] += v;
        wave_sync();
    } else {
        const int s = wave_bcast_i32(ri, 9);
        const int64_t woff = (int64_t)(((unsigned long long)(unsigned)wave_bcast_i32(ri, 11) << 32) | (unsigned)wave_bcast_i32(ri, 10));
        // (publish = false: the fronts above poll the words themselves -- tagged hand-offs, kernels_solve_fused.hpp -- and run in a LATER
        //  launch: plain stores, no drain, no counter)
        if (publish) {
            if (lane >= p && lane < f) st_agent(work + woff + lane, v);
            drain_stores();
            if (lane == 0) flag_add(done + s, 1);
        } else if (lane >= p && lane < f)
            work[woff + lane] = v;
    }
}

__global__ void __launch_bounds__(64 * WT_WAVES) k_wt_fwd(const WtWave *__restrict__ waves, const WtHdr *__restrict__ hdrs,
                                                        const int32_t *__restrict__ meta, const double *__restrict__ pool,
                                                        const int32_t *__restrict__ lperm, int *sync, double *work, double *x, int publish) {
    __shared__ __attribute__((aligned(16))) double lds[WT_WAVES][WT_LDS];
    const int lane = threadIdx.x & 63, wave = wave_uniform(threadIdx.x >> 6);
    // the wave's record: lanes 0 .. 15 hold the 16 words of its first batch's header, lanes 16 / 17 its four integers
    const long long *WV = reinterpret_cast<const long long *>(waves + (blockIdx.x * WT_WAVES + wave));
    const long long wrec = WV[lane & 31];
    const long long w16 = wave_bcast_i64(wrec, 16), w17 = wave_bcast_i64(wrec, 17);
    const int b0 = wave_uniform((int)(unsigned)(unsigned long long)w16), b1 = wave_uniform((int)(unsigned)((unsigned long long)w16 >> 32));
    const int xfirst = wave_uniform((int)(unsigned)(unsigned long long)w17), npiv = wave_uniform((int)(unsigned)((unsigned long long)w17 >> 32));
    if (b0 >= b1) return;
    HIPMF_STAMP((int)blockIdx.x / 6, 0);
    HIPMF_STAMP_VAL((int)blockIdx.x / 6, 4, b1 - b0);
    double *L = lds[wave];
    L[WT_OFF_Z + lane] = 0.0;
    int *done = sync + 16; // (SF_SYNC_HEADER of kernels_solve_fused.hpp)
    const long long *H = reinterpret_cast<const long long *>(hdrs);
    long long h0 = __shfl(wrec, lane & 15); // (lanes l and l + 16, ... hold word l & 15, as a header load would leave them)
    long long h1 = H[16 * (int64_t)(b0 + 1 < b1 ? b0 + 1 : b0) + (lane & 15)];
    // the subtree's part of the right-hand side and of the interchanges (the vectors are allocated with room for the over-read)
    // (chunks of 128 entries of x / 256 of the interchanges; a chunk past the subtree's pivots re-reads chunk 0 -- a cache hit -- instead
    //  of dragging the neighbouring subtrees' entries in: the average subtree has ~60 pivots)
    f64x2 xc[WT_X / 128];
    i32x4 lc[WT_X / 256];
#pragma unroll
    for (int c = 0; c < WT_X / 128; c++) xc[c] = ld_f64x2(x + xfirst + (128 * c < npiv ? 128 * c : 0) + 2 * lane);
#pragma unroll
    for (int c = 0; c < WT_X / 256; c++) lc[c] = ld_i32x4(lperm + xfirst + (256 * c < npiv ? 256 * c : 0) + 4 * lane);
    double pc[WT_NCH];
    i32x4 mc[2];
    wt_issue(h0, lane, pool, meta, pc, mc);
#pragma unroll
    for (int c = 0; c < WT_X / 128; c++) st_lds_f64x2(L + WT_OFF_X + 128 * c + 2 * lane, xc[c]);
#pragma unroll
    for (int c = 0; c < WT_X / 256; c++) st_lds_i32x4(reinterpret_cast<int32_t *>(L + WT_OFF_LP) + 256 * c + 4 * lane, lc[c]);
    HIPMF_STAMP((int)blockIdx.x / 6, 1);
    for (int j = b0; j < b1; j++) {
        if (j == b0 + 1) HIPMF_STAMP((int)blockIdx.x / 6, 5);
        wt_park(L, lane, pc, mc);
        if (j == b0 + 1) HIPMF_STAMP((int)blockIdx.x / 6, 6);
        const int nrec = wave_uniform((int)(unsigned)(unsigned long long)wave_bcast_i64(h0, WT_NCH + 1));
        if (j + 1 < b1) wt_issue(h1, lane, pool, meta, pc, mc); // (wave-uniform)
        const long long h2 = H[16 * (int64_t)(j + 2 < b1 ? j + 2 : b1 - 1) + (lane & 15)];
        wave_sync();
        if (j == b0 + 1) HIPMF_STAMP((int)blockIdx.x / 6, 7);
        if (j == b0 + 1) HIPMF_STAMP_VAL((int)blockIdx.x / 6, 9, nrec);
        for (int q = 0; q < nrec; q++) wt_fwd_front(L, q, lane, x, work, done, publish != 0);
        if (j == b0 + 1) HIPMF_STAMP((int)blockIdx.x / 6, 8);
        h0 = h1;
        h1 = h2;
        if (j == b0) HIPMF_STAMP((int)blockIdx.x / 6, 2);
    }
    HIPMF_STAMP((int)blockIdx.x / 6, 3);
}

// ------------------------------------------------------------------ backward: one front out of LDS
// Straight-line substitution for P = 4, 8, 12, 16 pivots (see wt_fwd_steps): the packed rows of U come with ZEROS below the diagonal
// of U11 (k_small_factor) and the diagonal entries are zeroed in LDS once the lane has its pivot, so step j is
// v_i -= u_ij (v_j / u_jj) for every lane; lane j gets its own v_j / u_jj at the end (the same product as in its step: later steps
// touch lanes < j only).
template <int P>
__device__ __forceinline__ double wt_bwd_steps(const double *L, int pslot, int lr, int p, int us, double v, double inv_d) {
    double a[P];
#pragma unroll
    for (int c = 0; c < P; c++) a[c] = L[p - 1 - c >= 0 ? pslot + lr + (p - 1 - c) * us : WT_OFF_Z];
#pragma unroll
    for (int c = 0; c < P; c++) {
        const int j = p - 1 - c >= 0 ? p - 1 - c : 0; // (columns past the first: a[c] = 0)
        const double vm = v * inv_d;
        const double vj = wave_bcast(vm, j);
        v -= a[c] * vj;
    }
    return v * inv_d;
}

__device__ __forceinline__ void wt_bwd_front(double *L, int q, int lane, double *x) {
    const int32_t *Mi = reinterpret_cast<const int32_t *>(L + WT_OFF_M);
    const double *X = L + WT_OFF_X; // y1 of the subtree (forward launch)
    double *ST = L + WT_OFF_ST, *xg = L + WT_OFF_XG;
    const int ri = Mi[16 * q + (lane & 15)];
    const int pslot = wave_bcast_i32(ri, 0), pm = wave_bcast_i32(ri, 1), xoff = wave_bcast_i32(ri, 2), ldsv = wave_bcast_i32(ri, 5);
    const int flags = wave_bcast_i32(ri, 6), relo = wave_bcast_i32(ri, 7), first = wave_bcast_i32(ri, 8);
    const int p = pm & 0xffff, m = pm >> 16, f = p + m;
    const int lds_self = (ldsv & 0xffff) - 1, lds_par = (ldsv >> 16) - 1;
    const int us = (flags & 2) ? p : f; // column stride of the rows of U
    double *Ub = L + pslot;
    const int sh = p <= 16 ? 4 : (p <= 32 ? 5 : 6);
    const int i = lane & ((1 << sh) - 1), jq = lane >> sh, ng = 64 >> sh;
    const int lr = lane < p ? lane : p - 1;
    // x2: from the parent's solved vector (inside the subtree) or from x (root: every ancestor is complete)
    if (lane < m) {
        const int idx = Mi[relo + lane];
        xg[lane] = (lds_par >= 0) ? ST[lds_par + idx] : x[idx];
    }
    const bool fast = (flags & 4) && p <= 16; // (wave-uniform) zeros below the diagonal of U11
    const double inv_d = 1.0 / Ub[lr + lr * us];
    wave_sync();
    if (fast && lane < p) Ub[lane + lane * us] = 0.0;
    double acc = 0.0;
    if (i < p) {
        const double *Ui = Ub + i + p * us;
        for (int j = jq; j < m; j += ng) acc += Ui[j * us] * xg[j];
    }
    for (int off = 1 << sh; off < 64; off <<= 1) acc += __shfl_xor(acc, off);
    double v = (lane < p) ? X[xoff + lane] - acc : 0.0;
    // x1 = U11^{-1} t, columns from right to left
    if (fast) {
        wave_sync();
        if (p <= 4) v = wt_bwd_steps<4>(L, pslot, lr, p, us, v, inv_d);
        else if (p <= 8) v = wt_bwd_steps<8>(L, pslot, lr, p, us, v, inv_d);
        else if (p <= 12) v = wt_bwd_steps<12>(L, pslot, lr, p, us, v, inv_d);
        else v = wt_bwd_steps<16>(L, pslot, lr, p, us, v, inv_d);
    } else {
        for (int jhi = p; jhi > 0; jhi -= 8) { // eight columns at a time
            double a8[8];
#pragma unroll
            for (int c = 0; c < 8; c++) a8[c] = Ub[lr + (jhi - 1 - c >= 0 ? jhi - 1 - c : 0) * us];
#pragma unroll
            for (int c = 0; c < 8; c++) {
                const int j = jhi - 1 - c;
                if (j >= 0) {
                    if (lane == j) v *= inv_d;
                    const double vj = wave_bcast(v, j);
                    if (lane < j) v -= a8[c] * vj;
                }
            }
        }
    }
    if (lane < p) x[first + lane] = v;
    if (lds_self >= 0) { // the children gather their x2 from [x1; x2]
        const double keep = (lane >= p && lane < f) ? xg[lane - p] : 0.0;
        wave_sync();
        if (lane < f) ST[lds_self + lane] = (lane < p) ? v : keep;
    }
    wave_sync();
}

// `arm` (round 6; tagged hand-offs, kernels_solve_fused.hpp): this is the LAST kernel of a pass pair, and by the time it starts nothing
// reads the words the tasks above the wave-subtrees handed to each other any more (their vectors in `work`, the shadow xt: the backward
// launches above are complete, this kernel reads x).  Every workgroup therefore re-arms its slice of those arm_words words with the tag
// pattern for the NEXT pass pair: the 16 MB hipMemsetAsync in front of every pass pair (5 - 6 us plus a launch gap inside the timed pair)
// is only needed before the first one.
__global__ void __launch_bounds__(64 * WT_WAVES) k_wt_bwd(const WtWave *__restrict__ waves, const WtHdr *__restrict__ hdrs,
                                                        const int32_t *__restrict__ meta, const double *__restrict__ pool, double *x,
                                                        double *arm, long long arm_words) {
    __shared__ __attribute__((aligned(16))) double lds[WT_WAVES][WT_LDS];
    if (arm) {
        const long long per = (arm_words + gridDim.x - 1) / gridDim.x, a0 = per * blockIdx.x;
        const long long a1 = a0 + per < arm_words ? a0 + per : arm_words;
        const double tagv = __longlong_as_double(-1LL); // (SF_TAG_BITS)
        for (long long i = a0 + threadIdx.x; i < a1; i += 64 * WT_WAVES) arm[i] = tagv;
    }
    const int lane = threadIdx.x & 63, wave = wave_uniform(threadIdx.x >> 6);
    const long long *WV = reinterpret_cast<const long long *>(waves + (blockIdx.x * WT_WAVES + wave));
    const long long wrec = WV[lane & 31]; // (see k_wt_fwd)
    const long long w16 = wave_bcast_i64(wrec, 16), w17 = wave_bcast_i64(wrec, 17);
    const int b0 = wave_uniform((int)(unsigned)(unsigned long long)w16), b1 = wave_uniform((int)(unsigned)((unsigned long long)w16 >> 32));
    const int xfirst = wave_uniform((int)(unsigned)(unsigned long long)w17), npiv = wave_uniform((int)(unsigned)((unsigned long long)w17 >> 32));
    if (b0 >= b1) return;
    double *L = lds[wave];
    L[WT_OFF_Z + lane] = 0.0;
    const long long *H = reinterpret_cast<const long long *>(hdrs);
    long long h0 = __shfl(wrec, lane & 15);
    long long h1 = H[16 * (int64_t)(b0 + 1 < b1 ? b0 + 1 : b0) + (lane & 15)];
    f64x2 xc[WT_X / 128];
#pragma unroll
    for (int c = 0; c < WT_X / 128; c++) xc[c] = ld_f64x2(x + xfirst + (128 * c < npiv ? 128 * c : 0) + 2 * lane);
    double pc[WT_NCH];
    i32x4 mc[2];
    wt_issue(h0, lane, pool, meta, pc, mc);
#pragma unroll
    for (int c = 0; c < WT_X / 128; c++) st_lds_f64x2(L + WT_OFF_X + 128 * c + 2 * lane, xc[c]);
    for (int j = b0; j < b1; j++) {
        wt_park(L, lane, pc, mc);
        const int nrec = wave_uniform((int)(unsigned)(unsigned long long)wave_bcast_i64(h0, WT_NCH + 1));
        if (j + 1 < b1) wt_issue(h1, lane, pool, meta, pc, mc);
        const long long h2 = H[16 * (int64_t)(j + 2 < b1 ? j + 2 : b1 - 1) + (lane & 15)];
        wave_sync();
        for (int q = 0; q < nrec; q++) wt_bwd_front(L, q, lane, x);
        h0 = h1;
        h1 = h2;
    }
}

} // namespace hipmf
