// kernels_factor.hpp -- dense partial factorisation of the fronts:  P F = [L11 0; L21 I] [U11 U12; 0 S]
//   k_small_factor   one or four wavefronts per front with f <= SMALL_F, whole front in LDS, blocked LU with the trailing
//                    update on MFMA (lds_lu_blocked)                                            (LDS / latency-bound)
//   k_panel          tiled path, step k0: every workgroup factorises the 32 x 32 diagonal tile in
//                    REGISTERS (one row per lane, v_readlane broadcasts; redundant per workgroup, which
//                    removes a dependent launch from the critical path) and then solves its own row /
//                    column tile against it                                                     (latency-bound)
//   k_update         tiled path, step k0: trailing update on v_mfma_f64_16x16x4_f64            (MFMA / HBM-bound)
// Big fronts are factorised augmented (kernels_common.hpp): the tiled kernels work on the index range
// [k0 + nb, f + k0 + nb) of both dimensions of [F Ic; Ir 0], which covers the not-yet-eliminated part of F, the columns
// of E that are already non-zero and the rows of E' that are already non-zero (AugView maps an index pair to F / E / E').
// Symmetric mode (template parameter SYM): L D L^T without interchanges on the lower triangle of F; the rows of U a step
// needs are D times the transposed columns of L, E' does not exist.
#pragma once
#include "kernels_common.hpp"

namespace hipmf {

// Partial LU of an f x f front (f <= 64) held in LDS (column-major, stride ld), p pivots, by a workgroup of NW wavefronts, LU_KB = 8
// pivots at a time -- the elimination of k_small_factor.  A rank-1 update per pivot moves every trailing entry through LDS three
// times per pivot (read, read of the pivot row, write) and needs three workgroup barriers per pivot.  Per block of eight pivots:
//   P1  wavefront 0 holds the eight panel columns in registers, one row per lane, and eliminates them with partial pivoting over
//       the rows c .. p-1 (implicit pivoting, v_readlane broadcasts, no barrier); it records where the interchanges of LAPACK's
//       row-swap convention put every row, writes the panel back in that order and leaves the eight interchanges in pivpos
//   P2  the interchanges are applied to all other columns (one thread per column)
//   P3  the eight rows of U right of the panel: substitution with the unit lower 8 x 8 block (one thread per column)
//   P4  trailing matrix -= L21 U12 on v_mfma_f64_16x16x4_f64, 16 x 16 tiles dealt to the wavefronts: every entry passes through LDS
//       once per eight pivots (this phase is bound by the LDS read-modify-write of the tiles, ~150 clocks per tile).
// lp[position] = front-local row that ended up there (identity on entry).  Static pivoting as everywhere: |pivot| < eps -> +-eps.
// (PIVOT = false: the pivot of step c is row c.)
constexpr int LU_KB = 8;
// PAIRED (see tile_lu32_z; the front's pivots and rows come in (real, imaginary) pairs, p and f even): the odd step of a pair takes
// the partner of the even step's pivot row -- physical positions 2 k, 2 k + 1 hold the two rows of ONE pair at all times (a pair moves
// to the pivot positions as a whole, the pair it displaces moves to where it came from), possibly in swapped order: which of the two
// is the real row is the parity of its original position, lp.  zd: where the complex pivots of this front go (2 doubles per pair).
template <int NW, bool PIVOT, bool PAIRED = false>
__device__ __forceinline__ void lds_lu_blocked(double *__restrict__ S, const int ld, const int f, const int p, int32_t *lp, int32_t *pivpos,
                                               const double eps, const double rep, FactorInfo *info, double *zd = nullptr) {
    constexpr int KB = LU_KB, T = 64 * NW;
    const int lin = threadIdx.x, lane = lin & 63, wave = lin >> 6;
    for (int c0 = 0; c0 < p; c0 += KB) {
        const int kb = (p - c0) < KB ? (p - c0) : KB;
        // ---- P1 ----
        if (wave == 0) {
            const int r = lane;
            const bool active = r >= c0 && r < f;
            double a[KB];
#pragma unroll
            for (int q = 0; q < KB; q++) a[q] = (active && q < kb) ? S[r + (c0 + q) * ld] : 0.0;
            int pos = r; // physical row this lane's row occupies after the interchanges so far
            bool chosen = false;
            int npert = 0, nzero = 0;
            const int lp_old = (PIVOT && r >= c0 && r < p) ? lp[r] : 0;
            int pv_prev = 0;
#pragma unroll
            for (int st = 0; st < KB; st++) {
                if (st < kb) { // (wave-uniform)
                    int pv;
                    // (every lane forms the reciprocal of its own entry while the arg-max runs: the divide is off the dependent chain)
                    const double myinv = fast_rcp(a[st]);
                    if (PAIRED && (st & 1)) {
                        pv = pv_prev ^ 1;
                    } else if (PIVOT) {
                        const bool cand = active && !chosen && r < p;
                        const unsigned mag = __float_as_uint((float)fabs(a[st]));
                        const unsigned key = cand ? ((mag & ~127u) | 64u | (unsigned)(63 - r)) : 0u;
                        pv = 63 - (int)(wave_max_u32(key) & 63u);
                    } else {
                        pv = c0 + st;
                    }
                    pv_prev = pv;
                    double d = wave_bcast(a[st], pv);
                    double inv = wave_bcast(myinv, pv);
                    if (fabs(d) < eps || d == 0.0) {
                        double dn = (d < 0.0) ? -rep : rep;
                        if (dn == 0.0) dn = 1.0; // eps == 0 requested and an exact zero: keep the factors finite
                        if (lane == pv) a[st] = dn;
                        npert++;
                        // An exactly zero pivot is what UMFPACK reports as a singular matrix (status 1, solver_umfpack.rs:492) -- when the
                        // whole column is zero.  Here the column goes on below the pivot block (rows that belong to the ancestors: a solver
                        // with dynamic pivoting would take one of them, interface_umfpack.c:167): a non-zero entry there means the
                        // matrix need not be singular -- the pivot counts as perturbed only (round 6: the +-1 family of the matrix zoo).
                        if (d == 0.0 && __ballot(active && r >= p && a[st] != 0.0) == 0ull) nzero++;
                        d = dn;
                        inv = 1.0 / dn;
                    }
                    if constexpr (PAIRED) {
                        if ((st & 1) == 0) {
                            // column c0 + st is a real-part column: the pair's real row holds Re there, its imaginary row Im
                            const double q = wave_bcast(a[st], pv ^ 1);
                            const int imag_row = wave_bcast_i32(lp_old, pv) & 1;
                            if (lane == 0) zd[c0 + st] = imag_row ? q : d, zd[c0 + st + 1] = imag_row ? d : q;
                        }
                    }
                    if (PIVOT) {
                        const int P = wave_bcast_i32(pos, pv);
                        if (pos == c0 + st) pos = P; // the row at the pivot position moves to where the chosen row was
                        if (lane == pv) pos = c0 + st;
                        if (lane == 0) pivpos[st] = P;
                    }
                    if (lane == pv) chosen = true;
                    const bool below = active && !chosen;
                    const double l = below ? a[st] * inv : 0.0;
                    if (below) a[st] = l;
#pragma unroll
                    for (int q = st + 1; q < KB; q++) a[q] -= l * wave_bcast(a[q], pv);
                }
            }
            if (active) {
#pragma unroll
                for (int q = 0; q < KB; q++)
                    if (q < kb) S[pos + (c0 + q) * ld] = a[q];
            }
            if (PIVOT && r >= c0 && r < p) lp[pos] = lp_old; // (all reads of lp happened above)
            if (lane == 0 && npert > 0) {
                atomicAdd(&info->n_perturbed, npert);
                if (nzero > 0) atomicAdd(&info->n_zero_pivot, nzero);
            }
        }
        __syncthreads();
        // ---- P2 ----
        if (PIVOT) {
            int pp[KB];
#pragma unroll
            for (int st = 0; st < KB; st++) pp[st] = pivpos[st]; // (stale beyond kb: not used)
            for (int col = lin; col < f; col += T) {
                if (col >= c0 && col < c0 + kb) continue;
#pragma unroll
                for (int st = 0; st < KB; st++) {
                    const int P = pp[st];
                    if (st < kb && P != c0 + st) {
                        const double t = S[(c0 + st) + col * ld];
                        S[(c0 + st) + col * ld] = S[P + col * ld];
                        S[P + col * ld] = t;
                    }
                }
            }
        }
        __syncthreads();
        const int R0 = c0 + kb; // first row / column of the trailing part
        if (R0 >= f) break;
        // ---- P3 ----
        for (int col = R0 + lin; col < f; col += T) {
            double x[KB];
#pragma unroll
            for (int t = 0; t < KB; t++) x[t] = (t < kb) ? S[(c0 + t) + col * ld] : 0.0;
#pragma unroll
            for (int t = 1; t < KB; t++) {
                if (t < kb) {
#pragma unroll
                    for (int q = 0; q < t; q++) x[t] -= S[(c0 + t) + (c0 + q) * ld] * x[q];
                }
            }
#pragma unroll
            for (int t = 1; t < KB; t++)
                if (t < kb) S[(c0 + t) + col * ld] = x[t];
        }
        __syncthreads();
        // ---- P4 ----
        {
            const int nt = (f - R0 + 15) >> 4, ntot = nt * nt;
            const int l15 = lane & 15, l4 = lane >> 4;
            // operands and the 16 x 16 tile itself with clamped addresses: every load is issued unconditionally (one LDS round trip per
            // tile), out-of-range operands are zeroed afterwards, stores are guarded.  C - L U is formed as (-L) U + C.
            const int ka = l4 < kb ? l4 : kb - 1, kb2 = 4 + l4 < kb ? 4 + l4 : kb - 1;
            const bool k0ok = l4 < kb, k1ok = 4 + l4 < kb;
            auto tile_load = [&](int ti, int tj, double &a0, double &a1, double &b0, double &b1, f64x4 &acc) {
                const int rA = R0 + 16 * ti + l15, cB = R0 + 16 * tj + l15;
                const int rAc = rA < f ? rA : f - 1, cBc = cB < f ? cB : f - 1;
                a0 = S[rAc + (c0 + ka) * ld], a1 = S[rAc + (c0 + kb2) * ld];
                b0 = S[(c0 + ka) + cBc * ld], b1 = S[(c0 + kb2) + cBc * ld];
#pragma unroll
                for (int g = 0; g < 4; g++) {
                    const int rr = R0 + 16 * ti + l4 + 4 * g;
                    acc[g] = S[(rr < f ? rr : f - 1) + cBc * ld];
                }
                if (rA >= f || !k0ok) a0 = 0.0;
                if (rA >= f || !k1ok) a1 = 0.0;
                if (cB >= f || !k0ok) b0 = 0.0;
                if (cB >= f || !k1ok) b1 = 0.0;
            };
            auto tile_store = [&](int ti, int tj, const f64x4 &acc) {
                const int cB = R0 + 16 * tj + l15;
#pragma unroll
                for (int g = 0; g < 4; g++) {
                    const int rr = R0 + 16 * ti + l4 + 4 * g;
                    if (rr < f && cB < f) S[rr + cB * ld] = acc[g];
                }
            };
            int ti = wave % nt, tj = wave / nt;
            for (int tile = wave; tile < ntot; tile += 2 * NW) { // two tiles per pass: their LDS round trips overlap
                int ti2 = ti + NW, tj2 = tj;
                while (ti2 >= nt) ti2 -= nt, tj2++;
                const bool two = tile + NW < ntot; // (wave-uniform)
                double a0, a1, b0, b1, c0a, c1a, d0, d1;
                f64x4 acc, acc2;
                tile_load(ti, tj, a0, a1, b0, b1, acc);
                if (two) tile_load(ti2, tj2, c0a, c1a, d0, d1, acc2);
                acc = mfma_f64_16x16x4(-a0, b0, acc);
                acc = mfma_f64_16x16x4(-a1, b1, acc);
                if (two) {
                    acc2 = mfma_f64_16x16x4(-c0a, d0, acc2);
                    acc2 = mfma_f64_16x16x4(-c1a, d1, acc2);
                }
                tile_store(ti, tj, acc);
                if (two) tile_store(ti2, tj2, acc2);
                ti = ti2 + NW, tj = tj2;
                while (ti >= nt) ti -= nt, tj++;
            }
        }
        __syncthreads();
    }
}

// Everything a small front needs to ASSEMBLE itself (fused into k_small_factor: no memset, scatter or extend-add
// traffic for the small fronts, which hold most of the fronts and half of the pool).
struct SmallAsm {
    const SmallDesc *sd;    // per position of the small lists
    const int32_t *sa_k;    // input entry k (>= 0), or ~k for the mirrored copy of a symmetric-lower entry
    const uint16_t *sa_pos; // row | column << 8 inside the front
    const double *vs, *vs2; // scaled values (k_absmax): vs2 = the mirrored entries of symmetric-lower storage
    const int32_t *child_idx, *rel;
    const int32_t *list0;   // start of the device array of lists (sd is indexed like it)
};

// One workgroup of NW wavefronts assembles and factorises one small front (f <= SMALL_F = 64) held entirely in LDS: thread
// (row r = tid mod 64, column group g = tid / 64).
//   assembly:  F = (scaled entries of A that belong to this front) + sum over the children of their contribution
//              (a tiled L D L^T child has had its block mirrored to a full one by k_mirror_cb)
//              blocks (children in ascending order, read from the pool where their own factorisation left them)
//   LU:        partial pivoting searches the whole remaining pivot block (rows c..p-1) with one 32-bit DPP max-reduction.
// No integer divisions and no per-element index arithmetic: every loop runs over columns with lane = row.
// NW = 1: one wavefront per front, the throughput shape of the leaf levels (tens of thousands of fronts per launch).
// NW = 4: the columns of the rank-1 updates (and the zero-fill, the gather of A's entries, the final store) are spread over four
//         wavefronts: a third of the latency per pivot, for the launches with few, large fronts between the leaves and the tiled
//         levels, where one front's LU (up to 64 pivots x ~1.3 us) is the whole launch.
template <int NW, bool PAIRED = false>
// (one wavefront per front: six waves per SIMD -- 78 registers instead of 96 -- measured 6.560 -> 6.545 ms; eight spill: 6.72)
__global__ void __launch_bounds__(64 * NW, NW == 1 ? 6 : 2) k_small_factor(const int32_t *__restrict__ list, const FrontDesc *__restrict__ FD,
                                                          double *__restrict__ pool, int32_t *__restrict__ lperm,
                                                          const unsigned long long *__restrict__ anorm_bits, double pivot_eps,
                                                          FactorInfo *info, int32_t ld, SmallAsm A, double *__restrict__ diag) {
    HIPMF_DYN_SHARED(double, sm);
    __shared__ int32_t lp[SMALL_F];
    __shared__ int32_t pivpos[LU_KB];
    const int tid = threadIdx.x & 63, grp = threadIdx.x >> 6; // row / lane, column group (wave)
    const int lin = threadIdx.x;
    const SmallDesc sdesc = A.sd[(list - A.list0) + blockIdx.x];
    const FrontDesc fd = sdesc.fd;
    const int p = fd.p, f = fd.p + fd.m;
    double *F = pool + fd.off;
    const double eps = pivot_eps * __longlong_as_double((long long)*anorm_bits);
    const double rep = pivot_replacement(pivot_eps, __longlong_as_double((long long)*anorm_bits));
    // descriptors of up to 64 children, one lane each (on their way while the front is zeroed and A's entries come in)
    const int nch = fd.child_end - fd.child_begin;
    int64_t d_cb = 0, d_ldc = 0, d_rel = 0;
    int d_m = 0;
    if (grp == 0 && tid < nch) { // (nch > 64: later batches are read below)
        const FrontDesc cd = FD[A.child_idx[fd.child_begin + tid]];
        d_ldc = cd.ld;
        d_cb = cd.off + cd.p + (int64_t)cd.p * cd.ld;
        d_rel = cd.rowptr;
        d_m = cd.m;
    }
    const int e0 = sdesc.e0, e1 = sdesc.e1;
    for (int e = lin; e < f * ld; e += 64 * NW) sm[e] = 0.0;
    __syncthreads();
    // entries of A, already scaled (LDS atomics: a caller's CSR may hold duplicates)
    for (int e = e0 + lin; e < e1; e += 64 * NW) {
        const int kk = A.sa_k[e];
        const int pos = A.sa_pos[e];
        const double v = kk < 0 ? A.vs2[~kk] : A.vs[kk];
        sm[(pos & 255) + (pos >> 8) * ld] = v; // (every entry has its own slot: validate_csr refuses duplicates)
    }
    __syncthreads();
    // children's contribution blocks (wavefront 0).  Per child the lanes are (row i, column group g) with 16 / 32 / 64 rows per pass
    // by the block's size, sixteen columns in flight per lane; rel of column j comes from the lane that holds it as a row.
    if (grp == 0) {
        for (int c0 = 0; c0 < nch; c0 += 64) {
            // descriptors of the children c0 .. c0 + 63, one per lane (the first batch was requested at the top of the kernel)
            if (c0 > 0) {
                d_m = 0;
                if (c0 + tid < nch) {
                    const FrontDesc cd = FD[A.child_idx[fd.child_begin + c0 + tid]];
                    d_ldc = cd.ld;
                    d_cb = cd.off + cd.p + (int64_t)cd.p * cd.ld;
                    d_rel = cd.rowptr;
                    d_m = cd.m;
                }
            }
            const int nbatch = nch - c0 < 64 ? nch - c0 : 64;
            if (nch > 64 && __ballot(d_m > 1) == 0ull) {
                // hub front: every child of this batch brings at most ONE entry (stars: a supply net and the pins that touch
                // nothing else).  One lane per child fetches it; lane 0 adds them in child order (fixed order: reproducible sums).
                const int myq = (tid < nbatch && d_m == 1) ? A.rel[d_rel] : -1;
                const double myv = myq >= 0 ? pool[d_cb] : 0.0;
                for (int l = 0; l < nbatch; l++) {
                    const int q = wave_bcast_i32(myq, l);
                    const double v = wave_bcast(myv, l);
                    if (tid == 0 && q >= 0) sm[q + q * ld] += v;
                }
                wave_sync();
                continue;
            }
            for (int cl = 0; cl < nbatch; cl++) {
                // (cl is wave-uniform: v_readlane broadcasts)
                const int64_t cbo = wave_bcast_i64(d_cb, cl), ldc = wave_bcast_i64(d_ldc, cl), relo = wave_bcast_i64(d_rel, cl);
                const int mc = wave_bcast_i32(d_m, cl);
                if (mc == 0) continue; // (wave-uniform)  mc <= f <= 64
                const double *CB = pool + cbo;
                const int sh = mc <= 16 ? 4 : (mc <= 32 ? 5 : 6);
                const int i = tid & ((1 << sh) - 1), g = tid >> sh, G = 64 >> sh;
                const int myrel = i < mc ? A.rel[relo + i] : 0; // lanes 0 .. mc-1 hold rel of rows (= columns) 0 .. mc-1
                for (int jb = 0; jb < mc; jb += 16 * G) { // (wave-uniform trip count: the shuffles below need every lane)
                    double cb[16];
#pragma unroll
                    for (int q = 0; q < 16; q++) {
                        const int j = jb + g + q * G;
                        cb[q] = (i < mc && j < mc) ? CB[i + (int64_t)j * ldc] : 0.0;
                    }
#pragma unroll
                    for (int q = 0; q < 16; q++) {
                        const int j = jb + g + q * G;
                        const int rj = __shfl(myrel, j & 63);
                        if (i < mc && j < mc) sm[myrel + rj * ld] += cb[q];
                    }
                }
                wave_sync(); // the next child may hit the same entries from other lanes
            }
        }
        if (tid < p) lp[tid] = tid;
    }
    __syncthreads();
    if constexpr (PAIRED) lds_lu_blocked<NW, true, true>(sm, ld, f, p, lp, pivpos, eps, rep, info, reinterpret_cast<const FactorInfoExt *>(info)->zdiag + fd.first);
    else lds_lu_blocked<NW, true>(sm, ld, f, p, lp, pivpos, eps, rep, info);
    // A front with a packed copy of its rows of U keeps L alone in its pivot block (zeros on and above the diagonal) and U alone in the
    // copy (zeros below the diagonal of U11): the wave-subtree solves (kernels_solve_tree.hpp) then need no per-lane tests in their
    // substitution steps; every other reader masks those entries anyway.
    const bool split = fd.epoff >= 0;
    if (tid < f) {
        int c = grp;
        for (; c + 7 * NW < f; c += 8 * NW) {
            double a[8];
#pragma unroll
            for (int q = 0; q < 8; q++) a[q] = sm[tid + (c + q * NW) * ld];
#pragma unroll
            for (int q = 0; q < 8; q++) F[tid + (c + q * NW) * f] = (split && c + q * NW < p && tid <= c + q * NW) ? 0.0 : a[q];
        }
        for (; c < f; c += NW) F[tid + c * f] = (split && c < p && tid <= c) ? 0.0 : sm[tid + c * ld];
    }
    if (split && tid < p) { // packed rows of U for the backward solve (lane = row: p-entry contiguous pieces)
        double *Up = pool + fd.epoff;
        for (int c = grp; c < f; c += NW) Up[tid + c * p] = (tid > c) ? 0.0 : sm[tid + c * ld];
    }
    if (grp == 0 && tid < p) {
        lperm[fd.first + tid] = lp[tid];
        diag[fd.first + tid] = sm[tid + tid * ld];
    }
}

// LU with partial pivoting of a 32 x 32 tile held one ROW PER LANE in registers (lanes 0..31).  A smaller
// tile is padded by the caller with identity rows / columns: its pivots are 1, they are chosen last and
// change nothing, so no size guards are needed in the (fully unrolled, branch-free) elimination.
// Pivoting is implicit: rows never move between lanes; `step` records at which elimination step this
// lane's row was chosen as the pivot row (ties go to the lowest row: deterministic).  On exit the row
// holds its multipliers in columns < step and its row of U in columns >= step; npert / nzero count the
// perturbed / exactly-zero pivots (wave-uniform).
// PAIRED (the real-equivalent form of a complex matrix: rows / columns 2 k, 2 k + 1 are the real and imaginary parts of complex row /
// column k, the tile starts at an even row): the pivot search runs at the EVEN steps only, the odd step takes the chosen row's partner
// (lane ^ 1) -- the two steps together are one step of a complex LU with partial pivoting (the pivot's real or imaginary part is the
// largest of the column: its modulus is within sqrt 2 of the largest), and the 2 x 2 blocks [a -b; b a] keep their shape in the
// Schur complement (to rounding).  zr + i zi: the complex pivot, valid in the lane that was chosen at an even step.
template <bool PIVOT = true, bool PAIRED = false>
__device__ __forceinline__ void tile_lu32_z(double (&a)[NB], int lane, double eps, double rep, int &step, int &npert, int &nzero, double &zr, double &zi) {
    step = -1;
    npert = 0;
    nzero = 0;
    int pv_prev = 0;
#pragma clang loop unroll(full)
    for (int c = 0; c < NB; c++) {
        // arg-max as ONE 32-bit max-reduction (4 DPP max steps): key = float(|a|) bits with the low 6 bits
        // replaced by a candidate flag and 31 - lane.  Candidates within 2^-18 of each other may be ordered by
        // lane instead of by magnitude, which is immaterial for the pivot choice.  Every candidate computes the
        // reciprocal of its own entry meanwhile, so the divide is off the dependent chain of the reduction.
        const bool cand = lane < NB && step < 0;
        const unsigned mag = __float_as_uint((float)fabs(a[c]));
        const unsigned key = cand ? ((mag & ~63u) | 32u | (unsigned)(31 - lane)) : 0u;
        const double myinv = fast_rcp(a[c]);
        // (PIVOT = false, the symmetric fronts: the pivot of step c is row c -- L D L^T in the guise of an LU without interchanges)
        int pv;
        if constexpr (PAIRED) pv = (c & 1) ? (pv_prev ^ 1) : 31 - (int)(wave_max_u32<2>(key) & 31u);
        else pv = PIVOT ? 31 - (int)(wave_max_u32<2>(key) & 31u) : c; // (candidates live in lanes 0..31: two rows)
        pv_prev = pv;
        if (lane == pv) step = c;
        double d = wave_bcast(a[c], pv);
        double inv = wave_bcast(myinv, pv);
        if (fabs(d) < eps || d == 0.0) {
            double dn = (d < 0.0) ? -rep : rep;
            if (dn == 0.0) dn = 1.0;
            if (lane == pv) a[c] = dn;
            npert++;
            if (d == 0.0) nzero++;
            inv = 1.0 / dn;
            d = dn;
        }
        if constexpr (PAIRED) {
            if ((c & 1) == 0) {
                // column c is a real-part column: the real row of the pair holds Re there, the imaginary row Im
                const double q = wave_bcast(a[c], pv ^ 1);
                if (lane == pv) zr = (pv & 1) ? q : d, zi = (pv & 1) ? d : q;
            }
        }
        const bool below = lane < NB && step < 0; // rows not yet chosen as pivot
        if (below) a[c] *= inv;
        // rows already chosen as pivots take part with a zero multiplier: the update needs no per-element
        // select (3 instructions per element: two v_readlane and one FMA), which also keeps the code small
        const double lmul = below ? a[c] : 0.0;
#pragma clang loop unroll(full)
        for (int cc = c + 1; cc < NB; cc++) a[cc] -= lmul * wave_bcast(a[cc], pv);
    }
}
template <bool PIVOT = true> __device__ __forceinline__ void tile_lu32(double (&a)[NB], int lane, double eps, double rep, int &step, int &npert, int &nzero) {
    double zr, zi;
    tile_lu32_z<PIVOT, false>(a, lane, eps, rep, step, npert, nzero, zr, zi);
}

// Tiled path, step 0 of the levels with MANY tiled fronts: one wavefront per front factorises the first diagonal tile and parks it
// (with its interchanges and pivots) where k_panel finds the tiles of the later steps.  k_panel's own "every workgroup factorises
// the tile itself" saves a dependent launch for the few large fronts near the root; with a thousand fronts in the level it is three
// redundant 5 us factorisations per front in workgroups whose other wavefront waits.
template <bool SYM, bool PAIRED = false>
__global__ void __launch_bounds__(64) k_diag0(const FrontDesc *__restrict__ LFD, double *__restrict__ pool, int32_t *__restrict__ lperm,
                                              double *__restrict__ dws, const unsigned long long *__restrict__ anorm_bits, double pivot_eps,
                                              FactorInfo *info, double *__restrict__ diag) {
    const int slot = blockIdx.x, tid = threadIdx.x;
    FrontDesc fd = LFD[slot];
    const int nb = fd.p < NB ? fd.p : NB;
    const double *F = pool + fd.off;
    const int64_t ld = fd.ld;
    double a[NB];
#pragma unroll
    for (int c = 0; c < NB; c++) {
        const int rr = (SYM && tid < c) ? c : tid, cc = (SYM && tid < c) ? tid : c; // (SYM: only the lower triangle of F is assembled)
        a[c] = (tid < nb && c < nb) ? F[rr + (int64_t)cc * ld] : (tid == c ? 1.0 : 0.0);
    }
    const double eps = pivot_eps * __longlong_as_double((long long)*anorm_bits);
    const double rep = pivot_replacement(pivot_eps, __longlong_as_double((long long)*anorm_bits));
    int step, npert, nzero;
    double zr = 0.0, zi = 0.0;
    tile_lu32_z<!SYM, PAIRED>(a, tid, eps, rep, step, npert, nzero, zr, zi);
    if (tid < nb) {
        double *dw = dws + (int64_t)slot * NB * NB; // step 0 uses buffer 0
        double dg = 1.0;
#pragma unroll
        for (int c = 0; c < NB; c++) {
            if (c < nb) dw[step + c * nb] = a[c];
            if (c == step) dg = a[c];
        }
        lperm[fd.first + step] = tid;
        diag[fd.first + step] = dg;
        store_zpivot<PAIRED>(info, fd.first, step, zr, zi);
    }
    if (tid == 0 && npert > 0) {
        atomicAdd(&info->n_perturbed, npert);
        if (nzero > 0) atomicAdd(&info->n_zero_pivot, nzero);
    }
}

// Memory access of the tiled kernels' bodies.  COH = false: plain loads and stores (one launch per step: the launch boundary orders
// everything).  COH = true: agent-scope (sc1) accesses for the chained launch (k_chain, kernels_factor_chain.hpp), whose workgroups
// consume what other workgroups of the SAME launch produced -- the eight XCDs' L2s are not coherent with each other inside a launch.
// Loads whose address is only valid under a condition come in two forms: `cond ? *p : 0` (the compiler predicates the load) and an
// unconditional load from a clamped address followed by a select.  An agent-scope load is never speculated, so the chained instance
// (COH) needs the second form everywhere -- one load per branch would wait for its own round trip (measured: 18 us instead of 8 for a
// panel tile).  For the plain instances, measured at 1000 x 1000 (profiles/r03_rejected_experiments.txt): the clamped form makes the panel
// solve of the L D L^T fronts faster (factorisation 6.68 -> 6.31 ms) and leaves LU where it was; in the look-ahead piece it costs
// 0.06 ms, in the tiles of the trailing update 15 % at 100^3 (address arithmetic in a throughput-bound loop).
#define HIPMF_CLAMP_PANEL 1
#define HIPMF_CLAMP_LA 0
template <bool COH> struct TileMem {
    static __device__ __forceinline__ double ld(const double *p) { return COH ? ld_agent(p) : *p; }
    static __device__ __forceinline__ void st(double *p, double v) {
        if (COH) st_agent(p, v);
        else *p = v;
    }
    static __device__ __forceinline__ int32_t ldi(const int32_t *p) { return COH ? flag_load(p) : *p; }
    static __device__ __forceinline__ void sti(int32_t *p, int32_t v) {
        if (COH) flag_store(p, v);
        else *p = v;
    }
};

// LDS of a panel workgroup.  D: L\U of the tile, row-major rows (16-byte aligned so that a thread can fetch a whole row of U with
// ds_read_b128 broadcasts); DT: its transpose (rows of DT = columns of L for the U-tile substitution)
struct PanelLds {
    __attribute__((aligned(16))) double D[NB][NB + 2];
    __attribute__((aligned(16))) double DT[NB][NB + 2];
    double T[NB][PANEL_T + 1];
    double dinv[NB];
    int32_t lp[NB];
};

// Tiled path, step k0 (base = k0 + nb, active range [base, f + base)):
//   every workgroup factorises the diagonal tile itself (wave 0, registers) while all its threads prefetch
//   the workgroup's own tile;  then
//   L tiles  (rows of the range, columns of the tile):   X <- X * U_kk^{-1}       (covers L21 and E')
//   U tiles  (columns of the range, rows of the tile):   X <- L_kk^{-1} * (P X)   (covers U12 and E)
// One thread owns one row (L) / one column (U) and runs a right-looking substitution in registers.
// The factorised tile is NOT written into F here (other workgroups still read the original): workgroup 0
// of each front parks it in dws, k_update moves it into place.
// SYM: the L tiles cover the rows of F below the tile only (and leave U12 = D L21^T in the upper triangle for the trailing update),
// the U tiles the columns of E only;
// the tile is factorised without interchanges from its lower triangle.
// One panel tile (tile t of the front in `slot`) of step k0.  NT = threads of the workgroup: PANEL_T of them work (k_chain's
// workgroups have 256: the others only take part in the barriers).
template <bool SYM, bool COH, int NT, bool PAIRED = false>
__device__ __forceinline__ void panel_body(PanelLds &sh, const int slot, const int t, const FrontDesc &fd, int32_t k0, double *__restrict__ pool,
                                           int32_t *__restrict__ lperm, double *__restrict__ dws, int32_t dws_stride,
                                           const unsigned long long *__restrict__ anorm_bits, double pivot_eps, FactorInfo *info,
                                           double *__restrict__ diag, int32_t pre_lu) {
    typedef TileMem<COH> M;
    constexpr bool CLP = COH || HIPMF_CLAMP_PANEL;
    double(&D)[NB][NB + 2] = sh.D;
    double(&DT)[NB][NB + 2] = sh.DT;
    double(&T)[NB][PANEL_T + 1] = sh.T;
    double(&dinv)[NB] = sh.dinv;
    int32_t(&lp)[NB] = sh.lp;
    const int tid = threadIdx.x;
    const bool act = NT == PANEL_T || tid < PANEL_T;
    const int f = fd.p + fd.m;
    const int nb = (fd.p - k0) < NB ? (fd.p - k0) : NB;
    const int base = k0 + nb, limit = f + base;
    const int nT = (f + PANEL_T - 1) / PANEL_T;
    const AugView A = aug_view(fd, pool);
    const bool ltile = t < nT;
    // first row (L) / column (U) of this tile and its extent.  A tile may straddle row / column f: an L tile's thread owns one row
    // (its own base pointer and stride: F or E'), a U tile's column is in F or in E (same stride, base picked per column).
    const int o0 = SYM ? (ltile ? base + t * PANEL_T : f + (t - nT) * PANEL_T) : base + (ltile ? t : t - nT) * PANEL_T;
    const int oend = (SYM && ltile) ? f : limit;
    const int ext = (oend - o0) < PANEL_T ? (oend - o0) : PANEL_T;
    const bool from_dws = k0 > 0 || pre_lu != 0; // the factorised diagonal tile is in dws (look-ahead of the previous step / k_diag0)
    if (ext <= 0 && (from_dws || t != 0)) return; // (workgroup 0 of step 0 still factorises and parks the diagonal tile)
    double *Lrow = A.at(o0 + tid, k0);                      // L tile: column k0 + u of this thread's row at Lrow[u * lstr]
    const int64_t lstr = (o0 + tid) >= f ? A.ps : A.ld;
    const bool lmixed = o0 < f && o0 + ext > f;             // (workgroup-uniform)
    const int64_t lstr_u = o0 >= f ? A.ps : A.ld;
    // 1. prefetch the tile, no interchange yet
    // (rows >= nb of T are zero: a partial tile is treated as a full one padded with identity)
    // (all 32 loads of a thread are issued before the first LDS store: one memory round trip, not four -- the step is a latency chain;
    //  the factorised diagonal tile of steps k0 > 0 and its row interchanges are requested first, in the same round trip)
    double tv[NB * NB / PANEL_T];
    int32_t lpv = 0;
    if (from_dws && act) {
        const double *dw = dws + ((int64_t)((k0 / NB) & 1) * dws_stride + slot) * NB * NB;
#pragma unroll
        for (int u = 0; u < NB * NB / PANEL_T; u++) {
            const int e = tid + u * PANEL_T, r = e % NB, c = e / NB;
            // (unconditional loads from clamped addresses, here and below: an agent-scope load is not speculated by the compiler, a
            //  load per branch would wait for its own round trip)
            const bool in = r < nb && c < nb;
            if (CLP) {
                const double dv = M::ld(dw + (in ? r + c * nb : 0));
                tv[u] = in ? dv : (r == c ? 1.0 : 0.0); // identity padding
            } else
                tv[u] = in ? dw[r + c * nb] : (r == c ? 1.0 : 0.0);
        }
        if (CLP) lpv = M::ldi(lperm + fd.first + k0 + (tid < nb ? tid : 0));
        else if (tid < nb) lpv = lperm[fd.first + k0 + tid];
    }
    if (ltile) {
        if (tid < ext) {
            double v[NB];
            if (lmixed) { // (the one tile per front that straddles row f: per-thread stride)
#pragma unroll
                for (int u = 0; u < NB; u++) {
                    const double lv = M::ld(Lrow + (int64_t)(u < nb ? u : 0) * lstr);
                    v[u] = (u < nb) ? lv : 0.0;
                }
            } else { // workgroup-uniform stride: the address arithmetic stays on the scalar unit
#pragma unroll
                for (int u = 0; u < NB; u++) {
                    const double lv = M::ld(Lrow + (int64_t)(u < nb ? u : 0) * lstr_u);
                    v[u] = (u < nb) ? lv : 0.0;
                }
            }
#pragma unroll
            for (int u = 0; u < NB; u++) T[u][tid] = v[u];
        }
    } else if (act && ext > 0) {
        const int k = tid & (NB - 1), cq = tid >> 5; // 4 columns x 32 rows per pass
        double v[PANEL_T / 4];
        const int kc = k < nb ? k : 0; // (clamped row: the load is unconditional, the value is dropped)
        const double *sF = A.F + (k0 + kc) + (int64_t)o0 * A.ld, *sE = A.Esh + (k0 + kc) + (int64_t)o0 * A.ld;
        // (a per-element choice between the two arrays costs ~1.2 us per launch in this latency chain -- measured: only the one tile
        //  per front that straddles column f pays it)
        if (o0 < f && o0 + ext > f) {
#pragma unroll
            for (int u = 0; u < PANEL_T / 4; u++) {
                const int cc = 4 * u + cq;
                if (CLP) {
                    const int ccl = cc < ext ? cc : 0;
                    const double uv = M::ld((o0 + ccl >= f ? sE : sF) + (int64_t)ccl * A.ld);
                    v[u] = (k < nb && cc < ext) ? uv : 0.0;
                } else
                    v[u] = (k < nb && cc < ext) ? (o0 + cc >= f ? sE : sF)[(int64_t)cc * A.ld] : 0.0;
            }
        } else {
            const double *sU = o0 >= f ? sE : sF;
#pragma unroll
            for (int u = 0; u < PANEL_T / 4; u++) {
                const int cc = 4 * u + cq;
                if (CLP) {
                    const double uv = M::ld(sU + (int64_t)(cc < ext ? cc : 0) * A.ld);
                    v[u] = (k < nb && cc < ext) ? uv : 0.0;
                } else
                    v[u] = (k < nb && cc < ext) ? sU[(int64_t)cc * A.ld] : 0.0;
            }
        }
#pragma unroll
        for (int u = 0; u < PANEL_T / 4; u++) T[k][4 * u + cq] = v[u];
    }
    // 2. the factorised diagonal tile.  Steps k0 > 0 find it in dws: workgroup 0 of the previous k_update factorised
    //    it right after updating it (look-ahead: that LU overlaps with the rest of the trailing update).
    if (from_dws) {
        if (act) {
#pragma unroll
            for (int u = 0; u < NB * NB / PANEL_T; u++) {
                const int e = tid + u * PANEL_T, r = e % NB, c = e / NB;
                D[r][c] = tv[u];
                DT[c][r] = tv[u];
                if (r == c) dinv[r] = 1.0 / tv[u];
            }
        }
        if (tid < NB) lp[tid] = (tid < nb) ? lpv - k0 : tid;
    } else if (tid < 64) {
        double a[NB];
        const double *F = A.F;
#pragma unroll
        for (int c = 0; c < NB; c++) {
            // (SYM: only the lower triangle of F is assembled)
            const int rr = (SYM && tid < c) ? c : tid, cc = (SYM && tid < c) ? tid : c;
            const bool in = tid < nb && c < nb;
            const double fv = M::ld(F + (k0 + (in ? rr : 0)) + (int64_t)(k0 + (in ? cc : 0)) * A.ld);
            a[c] = in ? fv : (tid == c ? 1.0 : 0.0);
        }
        const double eps = pivot_eps * __longlong_as_double((long long)*anorm_bits);
        const double rep = pivot_replacement(pivot_eps, __longlong_as_double((long long)*anorm_bits));
        int step, npert, nzero;
        double zr = 0.0, zi = 0.0;
        tile_lu32_z<!SYM, PAIRED>(a, tid, eps, rep, step, npert, nzero, zr, zi);
        if (tid < NB) {
            // rows go to LDS in pivot order: row `step` of the interchanged tile is this lane's row
            double dg = 1.0;
#pragma unroll
            for (int c = 0; c < NB; c++) {
                D[step][c] = a[c];
                DT[c][step] = a[c];
                if (c == step) dg = a[c];
            }
            dinv[step] = 1.0 / dg;
            lp[step] = tid;
            if (t == 0 && tid < nb) {
                diag[fd.first + k0 + step] = dg;
                store_zpivot<PAIRED>(info, fd.first + k0, step, zr, zi);
            }
        }
        if (t == 0 && tid == 0 && npert > 0) {
            atomicAdd(&info->n_perturbed, npert);
            if (nzero > 0) atomicAdd(&info->n_zero_pivot, nzero);
        }
    }
    __syncthreads();
    if (t == 0 && !from_dws && act) {
        double *dw = dws + (int64_t)slot * NB * NB; // step 0 uses buffer 0
        for (int e = tid; e < nb * nb; e += PANEL_T) M::st(dw + e, D[e % nb][e / nb]);
        if (tid < nb) M::sti(lperm + fd.first + k0 + tid, k0 + lp[tid]);
    }
    if (ext <= 0) return;
    // 3. substitution, right-looking, one row / column per thread, branch-free over the padded 32 steps
    if (tid < ext) {
        double x[NB];
        if (ltile) {
#pragma unroll
            for (int c = 0; c < NB; c++) x[c] = T[c][tid];
#pragma unroll
            for (int c = 0; c < NB; c++) {
                // the whole row of U is fetched before the FMA chain (otherwise every FMA waits on its own LDS read)
                double u[NB];
#pragma unroll
                for (int cc = 0; cc < NB; cc++) u[cc] = D[c][cc];
                x[c] *= dinv[c];
#pragma unroll
                for (int cc = c + 1; cc < NB; cc++) x[cc] -= x[c] * u[cc];
            }
#pragma unroll
            for (int c = 0; c < NB; c++) T[c][tid] = x[c];
        } else {
#pragma unroll
            for (int r = 0; r < NB; r++) x[r] = T[lp[r]][tid]; // row interchange applied here
#pragma unroll
            for (int k = 0; k < NB; k++) {
                double l[NB];
#pragma unroll
                for (int r = 0; r < NB; r++) l[r] = DT[k][r]; // column k of L, one aligned row of DT
#pragma unroll
                for (int r = k + 1; r < NB; r++) x[r] -= l[r] * x[k];
            }
#pragma unroll
            for (int r = 0; r < NB; r++) T[r][tid] = x[r];
        }
    }
    __syncthreads();
    if (ltile) {
        if (tid < ext) {
            if (lmixed) {
#pragma unroll
                for (int k = 0; k < NB; k++)
                    if (k < nb) M::st(Lrow + (int64_t)k * lstr, T[k][tid]);
            } else {
#pragma unroll
                for (int k = 0; k < NB; k++)
                    if (k < nb) M::st(Lrow + (int64_t)k * lstr_u, T[k][tid]);
            }
        }
        if (SYM && act) {
            // the rows of U the trailing update multiplies with: U12(k, c) = d_k L21(c, k) into the (otherwise unused) upper triangle
            // of F, rows k0 .. k0 + nb, columns of this tile (32-row segments of a column: the store pattern of the U tiles)
            const int k = tid & (NB - 1), cq = tid >> 5;
            const double dk = (k < nb) ? D[k][k] : 0.0;
            double *dst = A.F + (k0 + k) + (int64_t)o0 * A.ld;
#pragma unroll
            for (int cb = 0; cb < PANEL_T; cb += 4) {
                const int cc = cb + cq;
                if (k < nb && cc < ext) M::st(dst + (int64_t)cc * A.ld, dk * T[k][cc]);
            }
        }
    } else if (act) {
        const int k = tid & (NB - 1), cq = tid >> 5;
        double *dF = A.F + (k0 + k) + (int64_t)o0 * A.ld, *dE = A.Esh + (k0 + k) + (int64_t)o0 * A.ld;
        if (o0 < f && o0 + ext > f) {
#pragma unroll
            for (int cb = 0; cb < PANEL_T; cb += 4) {
                const int cc = cb + cq;
                if (k < nb && cc < ext) M::st((o0 + cc >= f ? dE : dF) + (int64_t)cc * A.ld, T[k][cc]);
            }
        } else {
            double *dU = o0 >= f ? dE : dF;
#pragma unroll
            for (int cb = 0; cb < PANEL_T; cb += 4) {
                const int cc = cb + cq;
                if (k < nb && cc < ext) M::st(dU + (int64_t)cc * A.ld, T[k][cc]);
            }
        }
    }
}

template <bool SYM, bool PAIRED = false>
__global__ void __launch_bounds__(PANEL_T) k_panel(const int32_t *__restrict__ pfx, int32_t nactive, const FrontDesc *__restrict__ LFD,
                                                   int32_t k0, double *__restrict__ pool,
                                                   int32_t *__restrict__ lperm, double *__restrict__ dws, int32_t dws_stride,
                                                   const unsigned long long *__restrict__ anorm_bits, double pivot_eps, FactorInfo *info,
                                                   double *__restrict__ diag, int32_t pre_lu, Pfx4 q4) {
    __shared__ PanelLds sh;
    int pfx_slot, slot;
    FrontDesc fd = load_front_pfx(pfx, nactive, LFD, blockIdx.x, q4, slot, pfx_slot); // (LFD: the descriptors of the level's tiled fronts in slot order)
    const int t = blockIdx.x - pfx_slot;
    fd_resident(fd);
    panel_body<SYM, false, PANEL_T, PAIRED>(sh, slot, t, fd, k0, pool, lperm, dws, dws_stride, anorm_bits, pivot_eps, info, diag, pre_lu);
}

// Tiled path, step k0: trailing update  A22 -= L21 * U12  on v_mfma_f64_16x16x4_f64 over the active
// range [base, f + base)^2 minus the (unused) corner where both indices are >= f.
// A 256-thread workgroup owns a 64 x 64 tile; each of its 4 waves owns 32 x 32 = 2 x 2 MFMA tiles.
// The product is formed transposed (D = U^T L^T) so that a result register of 16 adjacent lanes
// maps to 16 consecutive rows of one column: stores are 128-byte contiguous segments.
// LDS layouts: Ls[kk][r] (ld 80) and Us[c][kk] (ld 34) make the fragment reads of ds_read_b64
// conflict-free (banks = (dword address) mod 64) and both global->LDS copies conflict-free too.
//
// Two-level blocking (the read-modify-write of the trailing matrix is what bounds the large fronts, so it
// is done once per GROUP of G = fd.ugroup 32-wide panels; G = 2, or 4 / 8 for the largest fronts): step k0 is
// at position g = (k0 / 32) mod G of its group.
//   g < G - 1 and another step follows: NARROW update -- only the next panel's block column and block row
//                         (width nb2) receive the rank-32(g + 1) update of the group's panels so far;
//   g = G - 1:            the whole trailing matrix receives the update of all G panels, 32 columns of K at a
//                         time through the same LDS buffers;
//   g < G - 1 with nothing to follow: plain update of the whole trailing matrix with the panels so far.
// Workgroup 0 of every front also moves the factorised diagonal tile of this step from dws into the front
// and leaves the next diagonal tile to the look-ahead workgroup.
// Look-ahead: one extra workgroup per front (the last one, while a next diagonal tile exists) forms the NEXT
// diagonal tile  A - L U  from the 32 critical rows / columns with a single wavefront (lane r < 32 owns row r
// for columns 0..15, lane r + 32 for columns 16..31), factorises it in registers (tile_lu32) and parks it in
// dws (other buffer) with its row interchanges.  That 32 x 32 LU -- the longest serial piece of a tiled step --
// runs beside the trailing update instead of in front of the next panel solve.
// SYM: only entries of the lower triangle of F (r >= c) and of E are live; the rows of U inside F are U(k, c) = d_k L(c, k), written by
// k_panel into the otherwise unused upper triangle of the panel rows, so the update L D L^T is exactly symmetric and reads its operands
// exactly as the LU instance does; rows of E are read as they are.
// (Tried and rejected: 128 x 128 tiles, four 64 x 64 waves -- 204 VGPRs + 128 AGPRs, one workgroup per CU: 962 ms instead of
// 924 ms for the 128^3 Poisson factorisation.)
//
// LDS of an update workgroup (the look-ahead workgroup's tile buffer shares the space of Ls: it never touches Ls / Us)
// TS: tile edge.  64: a 256-thread workgroup, four waves of 32 x 32 (the large fronts); 32 (UPD_T_SMALL): ONE wave per tile -- the
// levels in the middle of the tree hold thousands of fronts of 65 - 200 rows, where a 64 x 64 grid wastes up to 44 % of every edge and
// three of a tile's four waves often have nothing live (k_update32).  A wave's work is the same in both: 2 x 2 MFMA tiles, same
// k order -- the two instances give bit-identical trailing matrices.
constexpr int UPD_T_SMALL = 32;
template <int TS> struct UpdateLdsT {
    static constexpr int LSLD = TS + 16; // (TS + 16) mod 32 == 16: the two kk rows of a ds_read_b64 pass fall into disjoint banks
    __attribute__((aligned(16))) double LsUM[(NB * LSLD > NB * (NB + 2)) ? NB * LSLD : NB * (NB + 2)];
    double Us[TS * US_LD];
};
typedef UpdateLdsT<UPD_T> UpdateLds;

// One tile (t < ntiles) or the look-ahead piece (t == ntiles) of the trailing update of step k0 of the front in `slot`.
// part (LU instances, full steps only): 0 = every tile; 1 = the tiles of the first block column and block row (what the next two panels
// and the look-ahead touch: the "critical strips") + the look-ahead piece; 2 = all other tiles.  A full step split this way runs its
// part 2 on a side stream beside the next group's panel steps (numeric.cpp); the tiles' arithmetic is the same: identical bits.
template <bool SYM, bool COH, int TS = UPD_T, bool PAIRED = false>
__device__ __forceinline__ void update_body(UpdateLdsT<TS> &sh, const int slot, const int t_in, const FrontDesc &fd, int32_t k0, double *__restrict__ pool,
                                            double *__restrict__ dws, int32_t dws_stride, int32_t *__restrict__ lperm,
                                            const unsigned long long *__restrict__ anorm_bits, double pivot_eps, FactorInfo *info,
                                            double *__restrict__ diag, const int part_in = 0, const int gshift = 0) {
    typedef TileMem<COH> M;
    const int part = part_in & 3;
    constexpr bool CLA = COH || HIPMF_CLAMP_LA;
    static_assert(TS == 64 || TS == 32, "tile edge");
    constexpr int NT = TS == 64 ? 256 : 64; // threads of the workgroup
    constexpr int LSLD = UpdateLdsT<TS>::LSLD;
    constexpr int MT = 2;                   // MFMA tiles per wave and dimension (a wave owns 32 x 32)
    constexpr int NE = TS * NB / NT;        // panel entries per thread and 32-column slice (L and U each)
    double *Ls = sh.LsUM;
    double *Us = sh.Us;
    double(*UM)[NB + 2] = reinterpret_cast<double(*)[NB + 2]>(sh.LsUM);
    const int tid = threadIdx.x;
    const int f = fd.p + fd.m;
    const int nb = (fd.p - k0) < NB ? (fd.p - k0) : NB;
    const int base = k0 + nb, limit = f + base;
    // Tiles never straddle row / column f: [base, f) (inside F) and [f, limit) (E' rows / E columns) are tiled separately, nt tiles per
    // dimension cover both parts, so a tile lives in ONE array with uniform base and stride.
    const int ntF = (f - base + TS - 1) / TS;                // tiles of the part inside F
    const int nt = ntF + (base + TS - 1) / TS;               // ... plus those of the E' rows / E columns (the host counts the same way)
    const int nb2 = (fd.p - base) < NB ? (fd.p - base) : NB; // size of the next diagonal tile (<= 0: none)
    const int gpos = (k0 / NB) % fd.ugroup;                  // position of this step in its group of panels
    const bool narrow = gpos < fd.ugroup - 1 && nb2 > 0;     // not the last step of the group and another step follows
    const int nhalf = gpos + 1;                              // 32-column slices of K: every panel of the group so far
    const int kfirst = k0 - gpos * NB;
    // SYM: only the tiles that hold live entries are enumerated (a grid of nt^2 workgroups of which half return at once is bound by the
    // workgroup dispatch rate on the large fronts: 38 000 empty workgroups cost 1.5 ms of a 3.9 ms launch at 100^3): the lower
    // triangle of the F part column by column, then the rows of F against the columns of E; narrow steps: block column, then E part of
    // the block row.  The host counts the same way.
    const int ntE = nt - ntF;
    const int ntri = ntF * (ntF + 1) / 2;
    const bool split = !SYM && part != 0 && !narrow; // this front's step is a full one and the launch carries one part of it
    const int ntiles_all = SYM ? (narrow ? nt : ntri + ntF * ntE) : (narrow ? 2 * nt : nt * nt);
    // tiles of this front in THIS launch (the host counts the same way): a narrow step belongs to part 1 as a whole
    const int ntiles = !split ? (part == 2 ? 0 : ntiles_all) : (part == 1 ? 2 * nt - 1 : (nt - 1) * (nt - 1));
    // index of the tile in the full enumeration
    int t = t_in;
    if (split && t_in < ntiles) {
        if (part == 1) t = t_in < nt ? t_in : (t_in - nt + 1) * nt;                                   // (ti, 0), then (0, tj), tj >= 1
        else t = (1 + t_in % (nt - 1)) + (1 + t_in / (nt - 1)) * nt;                                   // (ti, tj), both >= 1
    } else if (t_in == ntiles && part != 2)
        t = ntiles_all; // the look-ahead piece
    else if (t_in >= ntiles)
        return;
#ifndef HIPMF_XCD_MIN_NT
#define HIPMF_XCD_MIN_NT 8
#endif
    else if (!SYM && !narrow && (part_in & 4) && nt >= HIPMF_XCD_MIN_NT) {
        // XCD-aware order of a full step's tiles.  Workgroups go to the eight XCDs round-robin by their index in the launch (gshift = index
        // of this front's first workgroup mod 8), every XCD has its own L2, and a tile reads the block row of U of its tile column and the
        // block column of L of its tile row: with the plain order every XCD ends up fetching every panel.  Here the tiles are sorted by
        // (tile column mod 8, tile column, tile row) and XCD x takes the x-th eighth of that list: whole tile columns, so a block of U is
        // fetched into one L2 (its neighbours' boundary columns aside).  A permutation of the tiles: the arithmetic of a tile is untouched.
        const int x = (t_in + gshift) & 7;
        const int first_x = (x - gshift) & 7;
        int q = (t_in - first_x) >> 3; // how many of this XCD's workgroups came before
        for (int y = 0; y < x; y++) {
            const int first_y = (y - gshift) & 7;
            q += first_y < ntiles ? (ntiles - first_y + 7) >> 3 : 0;
        }
        int c = 0; // class of tile columns (column mod 8) the q-th tile of the sorted list belongs to
        for (; c < 7; c++) {
            const int size_c = ((nt - c + 7) >> 3) * nt;
            if (q < size_c) break;
            q -= size_c;
        }
        t = (q % nt) + (c + 8 * (q / nt)) * nt;
    }
    const AugView A = aug_view(fd, pool);
    double *F = A.F;
    if (t == ntiles_all) {
        // ---- look-ahead workgroup: only wave 0 works (its LDS phases are ordered by wave_sync: no workgroup barrier in here) ----
        if (tid >= 64) return;
        const int r = tid & 31, half = tid >> 5; // row, column half (16 columns each)
        double acc[16];
#pragma unroll
        for (int c = 0; c < 16; c++) {
            const int cc = half * 16 + c;
            const int rr = (SYM && r < cc) ? cc : r, c2 = (SYM && r < cc) ? r : cc; // (SYM: lower triangle)
            const bool in = r < nb2 && cc < nb2;
            if (CLA) {
                const double fv = M::ld(F + (base + (in ? rr : 0)) + (int64_t)(base + (in ? c2 : 0)) * A.ld);
                acc[c] = in ? fv : (r == cc ? 1.0 : 0.0);
            } else
                acc[c] = in ? F[(base + rr) + (int64_t)(base + c2) * A.ld] : (r == cc ? 1.0 : 0.0);
        }
        for (int h = 0; h < nhalf; h++) {
            const int kh = kfirst + h * NB, nbh = (h == nhalf - 1) ? nb : NB;
            if (h > 0) wave_sync();
            // U block (rows kh.., columns base..base+32) -> LDS, 16 elements per lane; the lane's row of L is requested in the same
            // round trip (the LDS stores wait for the U loads: everything this slice needs from memory is in flight before them)
            double um[16];
#pragma unroll
            for (int u = 0; u < 16; u++) {
                const int e = tid + 64 * u;
                const int kk = e & 31, c = e >> 5; // (SYM: k_panel left U12 = D L21^T in the upper triangle of the panel rows)
                const bool in = kk < nbh && c < nb2;
                if (CLA) {
                    const double fv = M::ld(F + (kh + (in ? kk : 0)) + (int64_t)(base + (in ? c : 0)) * A.ld);
                    um[u] = in ? fv : 0.0;
                } else
                    um[u] = in ? F[(kh + kk) + (int64_t)(base + c) * A.ld] : 0.0;
            }
            double lrow[NB];
#pragma unroll
            for (int kk = 0; kk < NB; kk++) {
                const bool in = r < nb2 && kk < nbh;
                if (CLA) {
                    const double fv = M::ld(F + (base + (in ? r : 0)) + (int64_t)(kh + (in ? kk : 0)) * A.ld);
                    lrow[kk] = in ? fv : 0.0;
                } else
                    lrow[kk] = in ? F[(base + r) + (int64_t)(kh + kk) * A.ld] : 0.0;
            }
#pragma unroll
            for (int u = 0; u < 16; u++) {
                const int e = tid + 64 * u;
                UM[e & 31][e >> 5] = um[u];
            }
            wave_sync();
#pragma unroll
            for (int kk = 0; kk < NB; kk++) {
                double u[16];
#pragma unroll
                for (int c = 0; c < 16; c++) u[c] = UM[kk][half * 16 + c];
#pragma unroll
                for (int c = 0; c < 16; c++) acc[c] -= lrow[kk] * u[c];
            }
        }
        // lanes 0..31 collect the other half of their row from lane + 32
        double a2[NB];
#pragma unroll
        for (int c = 0; c < 16; c++) {
            const double other = __shfl(acc[c], (tid + 32) & 63);
            a2[c] = half == 0 ? acc[c] : other;
            a2[16 + c] = half == 0 ? other : acc[c];
        }
        const double eps = pivot_eps * __longlong_as_double((long long)*anorm_bits);
        const double rep = pivot_replacement(pivot_eps, __longlong_as_double((long long)*anorm_bits));
        int step, npert, nzero;
        double zr = 0.0, zi = 0.0;
        tile_lu32_z<!SYM, PAIRED>(a2, tid, eps, rep, step, npert, nzero, zr, zi); // lanes >= 32 are not candidates and take no part
        if (tid < nb2) {
            double *dwo = dws + ((int64_t)(((k0 / NB) + 1) & 1) * dws_stride + slot) * NB * NB;
            double dgv = 1.0;
#pragma unroll
            for (int c = 0; c < NB; c++) {
                if (c < nb2) M::st(dwo + step + c * nb2, a2[c]);
                if (c == step) dgv = a2[c];
            }
            M::sti(lperm + fd.first + base + step, base + tid);
            diag[fd.first + base + step] = dgv;
            store_zpivot<PAIRED>(info, fd.first + base, step, zr, zi);
        }
        if (tid == 0 && npert > 0) {
            atomicAdd(&info->n_perturbed, npert);
            if (nzero > 0) atomicAdd(&info->n_zero_pivot, nzero);
        }
        return;
    }
    // tile of this workgroup; a narrow step has the nt tiles of the block column, then the nt tiles of the block row
    bool rowstrip;
    int ti, tj;
    if (!SYM) {
        rowstrip = narrow && t >= nt;
        ti = narrow ? (rowstrip ? 0 : t) : t % nt;
        tj = narrow ? (rowstrip ? t - nt : 0) : t / nt;
    } else if (narrow) {
        rowstrip = t >= ntF;
        ti = rowstrip ? 0 : t;
        tj = rowstrip ? t : 0; // (tiles ntF .. nt - 1 of the block row: the columns of E)
    } else {
        rowstrip = false;
        if (t < ntri) { // column tj of the lower triangle starts at off(tj) = tj ntF - tj (tj - 1) / 2
            const double bq = 2.0 * ntF + 1.0;
            int c = (int)((bq - sqrt(bq * bq - 8.0 * (double)t)) * 0.5);
            if (c < 0) c = 0;
            if (c > ntF - 1) c = ntF - 1;
            while (c > 0 && c * ntF - c * (c - 1) / 2 > t) c--;
            while (c + 1 < ntF && (c + 1) * ntF - (c + 1) * c / 2 <= t) c++;
            tj = c;
            ti = c + (t - (c * ntF - c * (c - 1) / 2));
        } else {
            const int e = t - ntri;
            ti = e % ntF;
            tj = ntF + e / ntF;
        }
    }
    const bool rowsE = ti >= ntF, colsE = tj >= ntF; // tile in the rows of E' / in the columns of E
    const int r0 = rowsE ? f + (ti - ntF) * TS : base + ti * TS, c0 = colsE ? f + (tj - ntF) * TS : base + tj * TS;
    const int rend = rowsE ? limit : f, cend = colsE ? limit : f;
    if (t == 0) {
        const double *dw = dws + ((int64_t)((k0 / NB) & 1) * dws_stride + slot) * NB * NB;
        double dv[NB * NB / NT];
#pragma unroll
        for (int u = 0; u < NB * NB / NT; u++) dv[u] = M::ld(dw + (tid + NT * u < nb * nb ? tid + NT * u : 0));
#pragma unroll
        for (int u = 0; u < NB * NB / NT; u++) {
            const int e = tid + NT * u;
            if (e < nb * nb) M::st(F + (k0 + e % nb) + (int64_t)(k0 + e / nb) * A.ld, dv[u]);
        }
    }
    if (r0 >= rend || c0 >= cend) return;          // no such tile
    if (rowsE && (SYM || colsE)) return;           // corner of the augmented front (SYM: there is no E'): never read
    if (SYM && !colsE && r0 + TS <= c0) return;    // strictly upper tile of F
    // entries this step may touch: rows < rmax, columns in [cmin, cmax)
    const int rmax = rowstrip ? base + nb2 : rend;
    const int cmax = (narrow && !rowstrip) ? base + nb2 : cend;
    const int cmin = rowstrip ? base + nb2 : 0;
    if (SYM && rowstrip && !colsE) return;         // row strip of a symmetric front: only the columns of E are live
    const int lane = tid & 63, wave = tid >> 6;
    const int wr = TS == 64 ? (wave & 1) * 32 : 0, wc = TS == 64 ? (wave >> 1) * 32 : 0; // this wave's 32 x 32 part of the tile
    const int l15 = lane & 15, l4 = lane >> 4;
    double lreg[NE], ureg[NE];
    const double *Lb = rowsE ? A.Epsh : F;             // rows of the L slice: (r, k) at Lb[r + k * lstr]
    const int64_t lstr = rowsE ? A.ps : A.ld;
    const double *Ub = colsE ? A.Esh : F;              // columns of the U slice: (k, c) at Ub[k + c * ld]
    double *Cb = rowsE ? A.Epsh : (colsE ? A.Esh : F); // the tile itself: (r, c) at Cb[r + c * cstr]
    const int64_t cstr = rowsE ? A.ps : A.ld;
    // slice h of the panels: global -> registers, registers -> LDS
#define HIPMF_FETCH_SLICE(h)                                                                                           \
    {                                                                                                                  \
        const int kh = kfirst + (h) * NB, nbh = ((h) == nhalf - 1) ? nb : NB;                                          \
        _Pragma("unroll") for (int u = 0; u < NE; u++) {                                                               \
            const int e = tid + NT * u;                                                                                \
            const int r = e % TS, kk = e / TS;                                                                         \
            const int k2 = e % NB, c = c0 + e / NB;                                                                    \
            const bool lin = r0 + r < rend && kk < nbh, uin = c < cend && k2 < nbh;                                    \
            if (COH) { /* (agent-scope loads are not speculated: unconditional, from clamped addresses) */            \
                const double lv = M::ld(Lb + (r0 + (lin ? r : 0)) + (int64_t)(kh + (lin ? kk : 0)) * lstr);            \
                const double uv = M::ld(Ub + (kh + (uin ? k2 : 0)) + (int64_t)(uin ? c : c0) * A.ld);                  \
                lreg[u] = lin ? lv : 0.0, ureg[u] = uin ? uv : 0.0;                                                    \
            } else { /* (the throughput-bound instance: the clamped form costs 15 % at 100^3) */                      \
                lreg[u] = lin ? Lb[(r0 + r) + (int64_t)(kh + kk) * lstr] : 0.0;                                        \
                ureg[u] = uin ? Ub[(kh + k2) + (int64_t)c * A.ld] : 0.0;                                               \
            }                                                                                                          \
        }                                                                                                              \
    }
#define HIPMF_STORE_SLICE()                                                                                            \
    {                                                                                                                  \
        _Pragma("unroll") for (int u = 0; u < NE; u++) {                                                               \
            const int e = tid + NT * u;                                                                                \
            Ls[(e / TS) * LSLD + e % TS] = lreg[u];                                                                    \
            Us[(e / NB) * US_LD + e % NB] = ureg[u];                                                                   \
        }                                                                                                              \
    }
    HIPMF_FETCH_SLICE(0)
    const bool owner0 = t == 0 && nb2 > 0 && wave == 0; // this wave's block holds the next diagonal tile
    // is entry (sub-tile a, b; register g) of this lane updated by this step?
    auto is_live = [&](int a, int b, int g) {
        const int r = r0 + wr + b * 16 + l15, c = c0 + wc + a * 16 + l4 + 4 * g;
        const bool corner = owner0 && (b * 16 + l15) < nb2 && (a * 16 + l4 + 4 * g) < nb2;
        const bool tri = !SYM || colsE || r >= c; // (SYM: lower triangle of F, all of E)
        return r < rmax && c < cmax && c >= cmin && !corner && tri;
    };
    // the 16 entries of the trailing matrix this lane updates are requested together with the first slice (one round trip for
    // both: a narrow step is a latency chain) and arrive while the MFMAs run
    double cur[MT][MT][4];
#pragma unroll
    for (int a = 0; a < MT; a++)
#pragma unroll
        for (int b = 0; b < MT; b++)
#pragma unroll
            for (int g = 0; g < 4; g++) {
                const int r = r0 + wr + b * 16 + l15, c = c0 + wc + a * 16 + l4 + 4 * g;
                const bool live = is_live(a, b, g);
                if (COH) {
                    const double cv = M::ld(Cb + (live ? r : r0) + (int64_t)(live ? c : c0) * cstr);
                    cur[a][b][g] = live ? cv : 0.0;
                } else
                    cur[a][b][g] = live ? Cb[r + (int64_t)c * cstr] : 0.0;
            }
    HIPMF_STORE_SLICE()
    __syncthreads();
    if (nhalf > 1) HIPMF_FETCH_SLICE(1) // in flight while the first slice is multiplied
    // strips of a narrow step: a wave whose quarter of the tile holds no live entry has nothing to multiply
    const bool wave_idle = (c0 + wc >= cmax) || (c0 + wc + 32 <= cmin) || (r0 + wr >= rmax) || (SYM && !colsE && r0 + wr + 32 <= c0 + wc);
    f64x4 acc[MT][MT];
#pragma unroll
    for (int a = 0; a < MT; a++)
#pragma unroll
        for (int b = 0; b < MT; b++) acc[a][b] = f64x4{0.0, 0.0, 0.0, 0.0};
    for (int h = 0; h < nhalf; h++) {
        if (h > 0) {
            __syncthreads();
            HIPMF_STORE_SLICE()
            __syncthreads();
            if (h + 1 < nhalf) HIPMF_FETCH_SLICE(h + 1) // the next slice, in flight while this one is multiplied
        }
        if (!wave_idle) {
#pragma unroll
            for (int kk0 = 0; kk0 < NB; kk0 += 4) {
                double ua[MT], lb[MT];
#pragma unroll
                for (int a = 0; a < MT; a++) ua[a] = Us[(wc + a * 16 + l15) * US_LD + kk0 + l4];
#pragma unroll
                for (int b = 0; b < MT; b++) lb[b] = Ls[(kk0 + l4) * LSLD + wr + b * 16 + l15];
#pragma unroll
                for (int a = 0; a < MT; a++)
#pragma unroll
                    for (int b = 0; b < MT; b++) acc[a][b] = mfma_f64_16x16x4(ua[a], lb[b], acc[a][b]);
            }
        }
    }
#undef HIPMF_FETCH_SLICE
#undef HIPMF_STORE_SLICE
    if (wave_idle) return;
#pragma unroll
    for (int a = 0; a < MT; a++)
#pragma unroll
        for (int b = 0; b < MT; b++)
#pragma unroll
            for (int g = 0; g < 4; g++) {
                const int r = r0 + wr + b * 16 + l15, c = c0 + wc + a * 16 + l4 + 4 * g;
                if (is_live(a, b, g)) M::st(Cb + r + (int64_t)c * cstr, cur[a][b][g] - acc[a][b][g]);
            }
}


#ifndef HIPMF_UPD_WAVES
#define HIPMF_UPD_BOUNDS __launch_bounds__(256)
#else
#define HIPMF_UPD_BOUNDS __launch_bounds__(256, HIPMF_UPD_WAVES)
#endif
#ifndef HIPMF_UPD32_WAVES
#define HIPMF_UPD32_BOUNDS __launch_bounds__(64)
#else
#define HIPMF_UPD32_BOUNDS __launch_bounds__(64, HIPMF_UPD32_WAVES)
#endif
template <bool SYM, bool PAIRED = false>
__global__ void HIPMF_UPD_BOUNDS k_update(const int32_t *__restrict__ pfx, int32_t nactive, const FrontDesc *__restrict__ LFD,
                                                int32_t k0, double *__restrict__ pool,
                                                double *__restrict__ dws, int32_t dws_stride, int32_t *__restrict__ lperm,
                                                const unsigned long long *__restrict__ anorm_bits, double pivot_eps, FactorInfo *info,
                                                double *__restrict__ diag, int32_t part, Pfx4 q4) {
    __shared__ UpdateLds sh;
    int pfx_slot, slot;
    FrontDesc fd = load_front_pfx(pfx, nactive, LFD, blockIdx.x, q4, slot, pfx_slot); // (LFD: the descriptors of the level's tiled fronts in slot order)
    const int t = blockIdx.x - pfx_slot;
    fd_resident(fd);
    update_body<SYM, false, UPD_T, PAIRED>(sh, slot, t, fd, k0, pool, dws, dws_stride, lperm, anorm_bits, pivot_eps, info, diag, SYM ? 0 : part, pfx_slot & 7);
}

// the same with 32 x 32 tiles, one wavefront per tile (levels whose largest tiled front has at most Solver::upd32_max_front rows)
template <bool SYM, bool PAIRED = false>
__global__ void HIPMF_UPD32_BOUNDS k_update32(const int32_t *__restrict__ pfx, int32_t nactive, const FrontDesc *__restrict__ LFD,
                                                 int32_t k0, double *__restrict__ pool,
                                                 double *__restrict__ dws, int32_t dws_stride, int32_t *__restrict__ lperm,
                                                 const unsigned long long *__restrict__ anorm_bits, double pivot_eps, FactorInfo *info,
                                                 double *__restrict__ diag, Pfx4 q4) {
    __shared__ UpdateLdsT<UPD_T_SMALL> sh;
    int pfx_slot, slot;
    FrontDesc fd = load_front_pfx(pfx, nactive, LFD, blockIdx.x, q4, slot, pfx_slot);
    const int t = blockIdx.x - pfx_slot;
    fd_resident(fd);
    update_body<SYM, false, UPD_T_SMALL, PAIRED>(sh, slot, t, fd, k0, pool, dws, dws_stride, lperm, anorm_bits, pivot_eps, info, diag);
}

} // namespace hipmf
