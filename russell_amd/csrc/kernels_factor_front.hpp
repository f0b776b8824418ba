// kernels_factor_front.hpp -- the fronts in the MIDDLE of the tree (f > 64, p <= 64, m <= 8 CM, LU mode): ONE workgroup carries a
// front through its whole partial factorisation in ONE launch per level and size class.
//
// The tiled path (kernels_factor.hpp) spends a launch pair per 32 pivots on these fronts -- k_panel + k_update, each a chain of
// memory round trips -- and a 64 x 64 tile grid of which a quarter is live (a front with f = 100, p = 30 is nine tiles for 9 100
// entries).  Here the p pivot ROWS of the front, [F11 F12] (p x f), stay on the CU -- F11 in LDS, F12 in the registers of eight
// wavefronts (lane = row, every wavefront holds every eighth column) -- and are reduced by Gauss-Jordan elimination with partial
// pivoting over the pivot block:
//
//      [F11 F12]  ->  [G  V],   G = inv(F11),  V = inv(F11) F12          (in place: column k of F11 becomes a column of G)
//
// after which the rest of the front is ONE product with the column panel F21 (fetched into LDS at the start of the kernel, beside the
// pivot rows: the elimination hides its latency):
//
//      [-W  S] = [0  F22] - F21 [G  V]           W = F21 inv(F11),  S = F22 - F21 inv(F11) F12  (the contribution block, in place)
//
// Every global load the kernel needs before its last phase is issued in its first microsecond; a dependent round trip to memory costs
// 1 - 2 us on this GPU and the first version of this kernel (round 4, profiles/r04_front_bench.txt) spent three quarters of its time
// in a dozen of them.
//
// What the solves need of a tiled front is a pair (E, E') with  [y1; u] = E b1 + [0; b2]  and  x1 = E' [y1; x2]  (two GEMVs without
// dependencies, kernels_common.hpp).  Any pair with  E'_left E_top = inv(F11),  E_bot = -F21 inv(F11),  E'_right = -inv(F11) F12
// serves; the tiled kernels leave (inv(L11) P, inv(U11)), this kernel leaves
//
//      E  = [G; -W]      (f x p, stride ld)              E' = [I | -V]      (p x f, stride p)
//
// Its E_top is a full p x p block (the tiled one is block lower triangular in 32-column blocks, which the forward slabs exploit):
// the descriptor carries FD_DENSE_TOP and the slabs read whole rows.  Pivots (diag) and interchanges (lperm) are those of an LU with
// the same pivot sequence: the entries of a Gauss-Jordan pivot column in the rows not chosen yet are the LU's.
//
// Per pivot k: the wavefront that owns column k (k mod 8) holds that column in a register since it updated it, picks the pivot (one
// 32-bit DPP max-reduction over the rows not chosen yet) and publishes the multipliers l_i = a_ik / d and the pivot row's lane
// through LDS -- one workgroup barrier per pivot; every wavefront then subtracts l_i x (pivot row) from its columns: those of F11
// by a read-modify-write of its own LDS columns (a column of F11 is only ever touched by its wavefront: no further barrier), those
// of F12 in registers with v_readlane broadcasts of the pivot row.  Pivot rows are NOT scaled inside the loop (a chosen row keeps d
// in its pivot column; every row is scaled by its own 1/d once at the end): the update is one FMA per entry for all lanes, the
// pivot lane takes part with a zero multiplier.
#pragma once
#include <type_traits>

#include "kernels_common.hpp"
#include "kernels_factor.hpp" // tile_lu32

namespace hipmf {

constexpr int MID_NW = 8;      // wavefronts per front
constexpr int MID_PMAX = 64;   // pivots at most: one pivot row per lane
constexpr int MID_MMAX = 192;  // off-diagonal rows at most (24 columns of F12 per wavefront and lane)
constexpr int MID_CHUNK = 64;  // columns of [G V] staged in LDS per pass of the product with F21 (8 per wavefront)
constexpr int MID_RBLD = 66;   // row stride of the staging buffer (even: 16-byte aligned rows for ds_read_b128 broadcasts;
                               // 132 dwords = 4 mod 64: the sixteen lanes of a ds_write_b64 group hit sixteen bank pairs)
constexpr int MID_LDS_DOUBLES = 12288; // dynamic LDS of a front at most (96 KB): F11 + staging buffer + F21

struct MidLds {
    double lm[2][64];     // multipliers of the current pivot, double-buffered: one barrier per pivot
    double dv[2];         // the pivot
    int32_t pv[2];        // lane of the pivot row
    int32_t rk[MID_PMAX]; // lane of the pivot row of every step (= the front-local row that became pivot row k)
};
// dynamic LDS of a front with p pivots and m off-diagonal rows, in doubles: F11 row-major with stride p | 1 (lane = row: an odd
// stride of doubles spreads the 32 lanes of a ds_read_b64 group over 32 bank pairs), the staging buffer of the product
// (p x MID_RBLD), F21 column-major (m x p)
__host__ __device__ inline int mid_s11_ld(int p) { return p | 1; }
__host__ __device__ inline int mid_rb_off(int p) { return (p * mid_s11_ld(p) + 1) & ~1; }
__host__ __device__ inline int mid_a21_off(int p) { return mid_rb_off(p) + p * MID_RBLD; }
__host__ __device__ inline int mid_lds_doubles(int p, int m) { return mid_a21_off(p) + p * m; }

template <int I, int N, class Fn> __device__ __forceinline__ void static_for(Fn &&fn) {
    if constexpr (I < N) {
        fn(std::integral_constant<int, I>{});
        static_for<I + 1, N>(fn);
    }
}

// CM: columns of F12 per wavefront and lane (fronts with m <= 8 CM off-diagonal rows)
template <int CM>
__global__ void __launch_bounds__(64 * MID_NW) k_front(const FrontDesc *__restrict__ LFD, double *__restrict__ pool, int32_t *__restrict__ lperm,
                                                       const unsigned long long *__restrict__ anorm_bits, double pivot_eps, FactorInfo *info,
                                                       double *__restrict__ diag) {
    HIPMF_DYN_SHARED(double, dyn);
    __shared__ MidLds sh;
    const int tid = threadIdx.x, lane = tid & 63, wave = wave_uniform(tid >> 6); // (the wavefront's number in a scalar register)
    HIPMF_STAMP(blockIdx.x, 0);
    const FrontDesc fd = LFD[blockIdx.x];
    fd_resident(fd);
    const int p = fd.p, m = fd.m, f = p + m;
    const int64_t ld = fd.ld;
    double *__restrict__ F = pool + fd.off;
    double *__restrict__ E = pool + fd.eoff;
    double *__restrict__ Ep = pool + fd.epoff;
    const int sld = mid_s11_ld(p);
    double *S11 = dyn;                  // (row, column) of F11 at S11[row * sld + column]
    double *RB = dyn + mid_rb_off(p);   // staging buffer of the product: [pivot step][MID_RBLD]
    double *A21 = dyn + mid_a21_off(p); // (row i, column k) of F21 at A21[i + k * m]
    const bool rowl = lane < p;                  // this lane holds a pivot row
    double *Srow = S11 + (rowl ? lane : 0) * sld; // (lanes beyond the pivot block read row 0 and never store)

    // ---- every load up front.  Pivot rows: lane = row; F12: this wavefront's columns p + wave, p + wave + 8, ... stay in registers;
    //      F11: its columns wave, wave + 8, ... go to LDS; F21: column k of the panel by wavefront k mod 8, 64 rows per load
    double a[CM];
    double pcol = 0.0; // column k of F11 while this wavefront owns the next pivot
    {
        const double *Fr = F + (lane < p ? lane : 0);
#pragma unroll
        for (int q = 0; q < CM; q++) {
            const int c = p + wave + MID_NW * q;
            const double v = Fr[(int64_t)(c < f ? c : 0) * ld]; // (clamped address: unconditional load)
            a[q] = (lane < p && c < f) ? v : 0.0;
        }
        constexpr int NJ = MID_PMAX / MID_NW; // columns of F11 / of F21 per wavefront at most
        double t[NJ];
#pragma unroll
        for (int u = 0; u < NJ; u++) {
            const int j = wave + MID_NW * u;
            t[u] = Fr[(int64_t)(j < p ? j : 0) * ld];
        }
        // F21: (MID_MMAX / 64) x NJ loads per thread at most, in two halves to bound the registers
        const double *Fp = F + p;
        constexpr int NRB = MID_MMAX / 64;
#pragma unroll
        for (int h = 0; h < 2; h++) {
            double w[NRB][NJ / 2];
#pragma unroll
            for (int rb = 0; rb < NRB; rb++)
#pragma unroll
                for (int u = 0; u < NJ / 2; u++) {
                    const int k = wave + MID_NW * (u + h * (NJ / 2)), i = rb * 64 + lane;
                    const bool in = k < p && i < m;
                    w[rb][u] = Fp[(in ? i : 0) + (int64_t)(in ? k : 0) * ld];
                }
            if (h == 0) {
#pragma unroll
                for (int u = 0; u < NJ; u++) {
                    const int j = wave + MID_NW * u;
                    if (j < p && rowl) Srow[j] = t[u];
                    if (j == 0) pcol = rowl ? t[u] : 0.0; // (wave 0 owns pivot 0)
                }
            }
#pragma unroll
            for (int rb = 0; rb < NRB; rb++)
#pragma unroll
                for (int u = 0; u < NJ / 2; u++) {
                    const int k = wave + MID_NW * (u + h * (NJ / 2)), i = rb * 64 + lane;
                    if (k < p && i < m) A21[i + k * m] = w[rb][u];
                }
        }
    }
    const double eps = pivot_eps * __longlong_as_double((long long)*anorm_bits);
    const double rep = pivot_replacement(pivot_eps, __longlong_as_double((long long)*anorm_bits));
    HIPMF_STAMP(blockIdx.x, 1);
    // ---- Gauss-Jordan over the pivot block ----
    int step = -1;     // the elimination step at which this lane's row was chosen
    double dval = 1.0; // its pivot
    int npert = 0, nzero = 0;
    for (int k = 0; k < p; k++) {
        const int ow = k & (MID_NW - 1), buf = k & 1;
        if (wave == ow) { // (wave-uniform)
            const double col = pcol;
            const bool cand = lane < p && step < 0;
            const unsigned mag = __float_as_uint((float)fabs(col));
            const unsigned key = cand ? ((mag & ~127u) | 64u | (unsigned)(63 - lane)) : 0u;
            const double myinv = fast_rcp(col); // (off the dependent chain of the reduction)
            const int pv = 63 - (int)(wave_max_u32(key) & 63u);
            double d = wave_bcast(col, pv);
            double inv = wave_bcast(myinv, pv);
            if (fabs(d) < eps || d == 0.0) {
                double dn = (d < 0.0) ? -rep : rep;
                if (dn == 0.0) dn = 1.0; // eps == 0 requested and an exact zero: keep the factors finite
                npert++;
                if (d == 0.0) nzero++;
                d = dn;
                inv = 1.0 / dn;
            }
            const double l = col * inv;
            sh.lm[buf][lane] = (lane == pv || !rowl) ? 0.0 : l;
            if (rowl) Srow[k] = (lane == pv) ? 1.0 : -l; // column k of the identity block, stored in place of column k of F11
            if (lane == 0) sh.pv[buf] = pv, sh.dv[buf] = d, sh.rk[k] = pv;
        }
        __syncthreads();
        const double lm = sh.lm[buf][lane];
        const int pv = wave_uniform(sh.pv[buf]);
        if (lane == pv) {
            step = k;
            dval = sh.dv[buf];
        }
        // this wavefront's columns of F11 (LDS; column k itself holds its final content already), the next pivot column among them
        const double *Spv = S11 + pv * sld;
        for (int j0 = wave; j0 < p; j0 += 4 * MID_NW) {
            double x[4], u[4];
#pragma unroll
            for (int t = 0; t < 4; t++) {
                const int j = j0 + MID_NW * t;
                const int jc = j < p ? j : wave; // (clamped: the loads are unconditional)
                x[t] = Srow[jc], u[t] = Spv[jc];
            }
#pragma unroll
            for (int t = 0; t < 4; t++) {
                const int j = j0 + MID_NW * t;
                const double y = __builtin_fma(-lm, u[t], x[t]);
                if (j < p && j != k && rowl) Srow[j] = y;
                if (j == k + 1) pcol = y;
            }
        }
#pragma unroll
        for (int q = 0; q < CM; q++) a[q] = __builtin_fma(-lm, wave_bcast(a[q], pv), a[q]);
    }
    HIPMF_STAMP(blockIdx.x, 2);
    if (lane == 0 && npert > 0) {
        atomicAdd(&info->n_perturbed, npert);
        if (nzero > 0) atomicAdd(&info->n_zero_pivot, nzero);
    }
    // every row by its own pivot: the lane that was chosen at step s holds row s of [G V]
    const double inv_own = 1.0 / dval;
#pragma unroll
    for (int q = 0; q < CM; q++) a[q] *= inv_own;
    __syncthreads(); // (sh.rk complete)
    // ---- E' = [I | -V], pivots, interchanges ----
    if (lane < p) {
#pragma unroll
        for (int q = 0; q < CM; q++) {
            const int c = p + wave + MID_NW * q;
            if (c < f) Ep[step + (int64_t)c * fd.ldp] = -a[q];
        }
        for (int j = wave; j < p; j += MID_NW) Ep[step + (int64_t)j * fd.ldp] = (j == step) ? 1.0 : 0.0;
        if (wave == 0) {
            diag[fd.first + step] = dval;
            lperm[fd.first + step] = lane;
        }
    }
    // ---- G in pivot order into the staging buffer (column position k of the in-place block is the column of front row rk[k]) and out:
    //      E_top(s, rk[k]) = G(s, k) ----
    if (lane < p) {
        for (int j = wave; j < p; j += MID_NW) {
            const double g = Srow[j] * inv_own;
            RB[step * MID_RBLD + j] = g;
            E[step + (int64_t)sh.rk[j] * ld] = g;
        }
    }
    __syncthreads();
    HIPMF_STAMP(blockIdx.x, 3);
    if (m == 0) return;
    // ---- [-W  S] = [0  F22] - F21 [G  V]: units of 64 rows of F21 x 8 staged columns, dealt to the wavefronts; lane = row ----
    const int nrb = (m + 63) >> 6;
    // the staged columns [0, nst) are columns of G (gpart: result -> E_bot, scattered by rk) or the columns c0 .. of the front
    // (result -> F22 in place: the entries are requested before the products and needed after them)
    auto product = [&](const int nst, const bool gpart, const int c0) {
        const int ngr = (nst + 7) >> 3, nun = ngr * nrb;
        for (int un = wave; un < nun; un += MID_NW) {
            const int rb = un / ngr, cs = (un - rb * ngr) * 8;
            const int ncol = (nst - cs) < 8 ? (nst - cs) : 8;
            const int i = rb * 64 + lane;
            const bool rowok = i < m;
            const int ic = rowok ? i : 0;
            double cin[8];
#pragma unroll
            for (int c = 0; c < 8; c++) {
                const double v = gpart ? 0.0 : F[(p + ic) + (int64_t)(c < ncol ? c0 + cs + c : c0) * ld];
                cin[c] = (!gpart && c < ncol) ? v : 0.0;
            }
            double acc[8];
#pragma unroll
            for (int c = 0; c < 8; c++) acc[c] = 0.0;
            const double *Rw = RB + cs;
            const double *Ai = A21 + ic;
            for (int k0 = 0; k0 < p; k0 += 4) {
                double ak[4];
#pragma unroll
                for (int kk = 0; kk < 4; kk++) {
                    const int k = k0 + kk;
                    const double v = Ai[(k < p ? k : 0) * m];
                    ak[kk] = k < p ? v : 0.0;
                }
#pragma unroll
                for (int kk = 0; kk < 4; kk++) {
                    const double *Rk = Rw + (k0 + kk < p ? k0 + kk : 0) * MID_RBLD; // (rows beyond p: multiplied by ak = 0)
#pragma unroll
                    for (int c = 0; c < 8; c++) acc[c] = __builtin_fma(ak[kk], Rk[c], acc[c]);
                }
            }
            if (rowok) {
#pragma unroll
                for (int c = 0; c < 8; c++) {
                    if (c < ncol) {
                        if (gpart) E[(p + i) + (int64_t)wave_uniform(sh.rk[cs + c]) * ld] = -acc[c];
                        else F[(p + i) + (int64_t)(c0 + cs + c) * ld] = cin[c] - acc[c];
                    }
                }
            }
        }
    };
    product(p, true, 0);
    HIPMF_STAMP(blockIdx.x, 4);
    // the columns of V, 64 at a time: every wavefront stages its own (every eighth) column
    static_for<0, (CM + 7) / 8>([&](auto chc) {
        constexpr int ch = decltype(chc)::value;
        if (ch * MID_CHUNK < m) { // (workgroup-uniform)
            __syncthreads();      // the block staged before has been consumed
            if (lane < p) {
#pragma unroll
                for (int qq = 0; qq < 8; qq++) {
                    constexpr int qbase = 8 * ch;
                    if (qbase + qq < CM) RB[step * MID_RBLD + MID_NW * qq + wave] = a[(qbase + qq) < CM ? (qbase + qq) : 0];
                }
            }
            __syncthreads();
            const int c0 = p + ch * MID_CHUNK;
            product((f - c0) < MID_CHUNK ? (f - c0) : MID_CHUNK, false, c0);
        }
    });
    HIPMF_STAMP(blockIdx.x, 5);
}

// ------------------------------------------------------------------------------------------------------------------------------------
// k_front_lu -- the same fronts with at most 32 pivots (most of levels 3 - 6 of a 2D mesh), second formulation (round 4).
// k_front above spends ~800 wavefront-instructions per pivot on 8 wavefronts (pivot search, hand-off through LDS, three instructions
// per entry of the rank-1 updates) for p f / 64 useful multiply-adds: it is bound by instruction issue, and so are the levels it runs
// on (thousands of fronts).  Here the only sequential piece is the inversion of the p x p pivot block by ONE wavefront in registers:
// in-place Gauss-Jordan with partial pivoting, one row per lane, the 32 columns in 32 register pairs (the pattern of tile_lu32, the tiled
// path's diagonal-tile factorisation; identity padding up to 32, ~0.35 us per pivot, no LDS, no barrier).  Everything else is a
// product with G = inv(F11) whose loops run over k with static accumulators -- compact code, ~1.5 instructions per multiply-add
// (the operands of G come as LDS broadcasts):
//
//      G = inv(F11)                     wavefront 0, rows in registers           (a first version factorised P F11 = L U and substituted:
//      W = F21 G                        lane = row of F21, 32 accumulators        64 unrolled substitution steps are 50 KB of straight-
//      V = G F12                        lane = column of F12, 32 accumulators     line code that runs once per wavefront -- instruction
//      S = F22 - W F12                  lane = row, 8 columns per unit            fetch made it slower than k_front, profiles/r04_front_bench.txt)
//
// The pair (E, E') it leaves is k_front's (FD_DENSE_TOP): E = [G; -W], E' = [I | -V]; pivots and interchanges are those of an LU with
// the same pivot sequence.  Column position k of the in-place block belongs to front row rk[k] (the row chosen at step k), as in k_front.
// LDS (doubles): Gs[p][ldd] = G by pivot step and column position, GsT its transpose; B12[p][ldb] the rows of F12; A21 / Wp [m x p]
// column-major: F21 first, then (in place, row by row: a row is read and written by the same lane) W by column position.
constexpr int MIDL_P = 32;      // pivots at most
constexpr int MIDL_NW = 4;      // wavefronts per front: the inversion is one wavefront's work; four wavefronts leave room for three fronts per CU
constexpr int MIDL_LDS_DOUBLES = 16384; // 128 KB
__host__ __device__ inline int midl_ldd(int) { return MIDL_P + 2; } // (rows of Gs / GsT are 32 wide whatever p: positions beyond p hold zeros, the product loops carry no tests)
__host__ __device__ inline int midl_ldb(int m) { return (m + 3) & ~1; }
__host__ __device__ inline int midl_off_dt(int p) { return MIDL_P * midl_ldd(p); }     // (Gs and GsT are 32 x 34 whatever p)
__host__ __device__ inline int midl_off_b(int p) { return 2 * MIDL_P * midl_ldd(p); }
__host__ __device__ inline int midl_off_y(int p, int m) { return midl_off_b(p) + p * midl_ldb(m); }
__host__ __device__ inline int midl_lds_doubles(int p, int m) { return midl_off_y(p, m) + p * m; }

struct MidlLds {
    int32_t rk[MIDL_P]; // front-local row that became pivot row k (= the front row column position k of the in-place block belongs to)
};

// In-place inversion of a 32 x 32 block held one ROW PER LANE in registers (lanes 0..31), Gauss-Jordan with partial pivoting; a smaller
// block is padded by the caller with identity rows / columns (pivots 1, chosen last).  Implicit pivoting: rows never move between lanes;
// `step` = the elimination step at which this lane's row was chosen, `dval` its pivot.  Pivot rows are not scaled inside the loop (the
// update is one FMA per entry for every lane, the pivot lane takes part with a zero multiplier); on exit row `step` of the inverse sits in
// this lane, column position k belonging to the row chosen at step k (rk, written by the caller's lane).  np: steps to run (<= 32).
// PAIRED: as tile_lu32_z (np even) -- the odd step takes the partner row of the even step's pivot; zr + i zi: the pair's complex pivot.
template <bool PAIRED = false>
__device__ __forceinline__ void tile_inv32(double (&a)[MIDL_P], int lane, int np, double eps, double rep, int &step, double &dval, int32_t *rk, int &npert, int &nzero,
                                           double &zr, double &zi) {
    step = -1;
    dval = 1.0;
    npert = 0;
    nzero = 0;
    int pv_prev = 0;
#pragma clang loop unroll(full)
    for (int c = 0; c < MIDL_P; c++) {
        if (c < np) { // (wave-uniform)
            const bool cand = lane < MIDL_P && step < 0;
            const unsigned mag = __float_as_uint((float)fabs(a[c]));
            const unsigned key = cand ? ((mag & ~63u) | 32u | (unsigned)(31 - lane)) : 0u;
            const double myinv = fast_rcp(a[c]);
            int pv;
            if constexpr (PAIRED) pv = (c & 1) ? (pv_prev ^ 1) : 31 - (int)(wave_max_u32<2>(key) & 31u);
            else pv = 31 - (int)(wave_max_u32<2>(key) & 31u);
            pv_prev = pv;
            double d = wave_bcast(a[c], pv);
            double inv = wave_bcast(myinv, pv);
            if (fabs(d) < eps || d == 0.0) {
                double dn = (d < 0.0) ? -rep : rep;
                if (dn == 0.0) dn = 1.0;
                npert++;
                if (d == 0.0) nzero++;
                d = dn;
                inv = 1.0 / dn;
            }
            if constexpr (PAIRED) {
                if ((c & 1) == 0) {
                    const double q = wave_bcast(a[c], pv ^ 1);
                    if (lane == pv) zr = (pv & 1) ? q : d, zi = (pv & 1) ? d : q;
                }
            }
            if (lane == pv) step = c, dval = d, rk[c] = lane;
            const double lm = (lane == pv) ? 0.0 : a[c] * inv;
#pragma clang loop unroll(full)
            for (int cc = 0; cc < MIDL_P; cc++)
                if (cc != c) a[cc] -= lm * wave_bcast(a[cc], pv);
            a[c] = (lane == pv) ? 1.0 : -lm; // column c of the identity block, in place of column c of the block
        }
    }
    const double inv_own = 1.0 / dval;
#pragma unroll
    for (int cc = 0; cc < MIDL_P; cc++) a[cc] *= inv_own;
}

template <bool PAIRED = false>
__global__ void __launch_bounds__(64 * MIDL_NW) k_front_lu(const FrontDesc *__restrict__ LFD, double *__restrict__ pool, int32_t *__restrict__ lperm,
                                                           const unsigned long long *__restrict__ anorm_bits, double pivot_eps, FactorInfo *info,
                                                           double *__restrict__ diag) {
    HIPMF_DYN_SHARED(double, dyn);
    __shared__ MidlLds sh;
    const int tid = threadIdx.x, lane = tid & 63, wave = wave_uniform(tid >> 6);
    HIPMF_STAMP(blockIdx.x, 0);
    const FrontDesc fd = LFD[blockIdx.x];
    fd_resident(fd);
    const int p = fd.p, m = fd.m;
    const int64_t ld = fd.ld;
    double *__restrict__ F = pool + fd.off;
    double *__restrict__ E = pool + fd.eoff;
    double *__restrict__ Ep = pool + fd.epoff;
    const int ldd = midl_ldd(p), ldb = midl_ldb(m);
    double *Gs = dyn, *GsT = dyn + midl_off_dt(p), *B12 = dyn + midl_off_b(p), *Wp = dyn + midl_off_y(p, m);
    const int nrb = (m + 63) >> 6;
    // ---- every load up front.  Wavefront 0: its row of F11 (registers).  Wavefronts 1 .. 3: F12 -> LDS row-major (a column of F12 is p
    //      contiguous doubles in memory: lanes = rows) and F21 -> LDS column-major (lanes = rows of F21), eight loads in flight per lane ----
    double a[MIDL_P];
    if (wave == 0) {
#pragma unroll
        for (int c = 0; c < MIDL_P; c++) {
            const bool in = lane < p && c < p;
            const double v = F[(in ? lane : 0) + (int64_t)(in ? c : 0) * ld];
            a[c] = in ? v : (lane == c ? 1.0 : 0.0); // identity padding: those pivots are 1 and are chosen last
        }
    } else {
        const double *Fr = F + (lane < p ? lane : 0);
        for (int j0 = wave - 1; j0 < m; j0 += 8 * (MIDL_NW - 1)) {
            double t[8];
#pragma unroll
            for (int u = 0; u < 8; u++) {
                const int j = j0 + (MIDL_NW - 1) * u;
                t[u] = Fr[(int64_t)(p + (j < m ? j : 0)) * ld];
            }
#pragma unroll
            for (int u = 0; u < 8; u++) {
                const int j = j0 + (MIDL_NW - 1) * u;
                if (j < m && lane < p) B12[lane * ldb + j] = t[u];
            }
        }
        for (int rb = 0; rb < nrb; rb++) {
            const int i = rb * 64 + lane;
            const double *Fi = F + p + (i < m ? i : 0);
            for (int k0 = wave - 1; k0 < p; k0 += 8 * (MIDL_NW - 1)) {
                double t[8];
#pragma unroll
                for (int u = 0; u < 8; u++) {
                    const int k = k0 + (MIDL_NW - 1) * u;
                    t[u] = Fi[(int64_t)(k < p ? k : 0) * ld];
                }
#pragma unroll
                for (int u = 0; u < 8; u++) {
                    const int k = k0 + (MIDL_NW - 1) * u;
                    if (k < p && i < m) Wp[i + k * m] = t[u];
                }
            }
        }
    }
    HIPMF_STAMP(blockIdx.x, 1);
    // ---- G = inv(F11) in the registers of wavefront 0; out: Gs / GsT (LDS), E_top, the identity part of E', pivots, interchanges ----
    if (wave == 0) {
        const double eps = pivot_eps * __longlong_as_double((long long)*anorm_bits);
        const double rep = pivot_replacement(pivot_eps, __longlong_as_double((long long)*anorm_bits));
        int step, npert, nzero;
        double dval, zr = 0.0, zi = 0.0;
        tile_inv32<PAIRED>(a, lane, p, eps, rep, step, dval, sh.rk, npert, nzero, zr, zi);
        wave_sync(); // (sh.rk of every step visible to the lanes of this wavefront)
        if (lane < p) { // (the rows of the block are the lanes 0 .. p-1: each was chosen at some step < p)
#pragma unroll
            for (int k = 0; k < MIDL_P; k++) {
                Gs[step * ldd + k] = a[k], GsT[k * ldd + step] = a[k]; // (k >= p: zeros -- the padded columns of a real row stay zero)
                if (k < p) {
                    E[step + (int64_t)sh.rk[k] * ld] = a[k];
                    Ep[step + (int64_t)k * fd.ldp] = (k == step) ? 1.0 : 0.0;
                }
            }
            diag[fd.first + step] = dval;
            lperm[fd.first + step] = lane;
            store_zpivot<PAIRED>(info, fd.first, step, zr, zi);
        }
        if (lane == 0 && npert > 0) {
            atomicAdd(&info->n_perturbed, npert);
            if (nzero > 0) atomicAdd(&info->n_zero_pivot, nzero);
        }
    }
    __syncthreads();
    HIPMF_STAMP(blockIdx.x, 2);
    // ---- W = F21 G (units: 64 rows of F21, lane = row; in place in LDS) and V = G (rk-rows of F12) (units: 64 columns, lane = column):
    //      the units are dealt to the wavefronts; 32 static accumulators, the loop runs over k ----
    {
        const int ncu = (m + 63) >> 6;
        for (int un = wave; un < nrb + ncu; un += MIDL_NW) {
            double acc[MIDL_P];
#pragma unroll
            for (int c = 0; c < MIDL_P; c++) acc[c] = 0.0;
            if (un < nrb) {
                const int i = un * 64 + lane;
                const double *Ai = Wp + (i < m ? i : 0);
                for (int k = 0; k < p; k++) {
                    const double av = Ai[k * m];
                    const double *Gk = Gs + k * ldd; // row k of G by column position
#pragma unroll
                    for (int c = 0; c < MIDL_P; c++) acc[c] = __builtin_fma(av, Gk[c], acc[c]);
                }
                if (i < m) {
#pragma unroll
                    for (int c = 0; c < MIDL_P; c++)
                        if (c < p) {
                            Wp[i + c * m] = acc[c]; // (this lane's row: read above, written here)
                            E[(p + i) + (int64_t)wave_uniform(sh.rk[c]) * ld] = -acc[c];
                        }
                }
            } else {
                const int j = (un - nrb) * 64 + lane;
                const double *Bj = B12 + (j < m ? j : 0);
                for (int k = 0; k < p; k++) {
                    const double bv = Bj[wave_uniform(sh.rk[k]) * ldb];
                    const double *Gk = GsT + k * ldd; // column position k of G, by pivot step
#pragma unroll
                    for (int c = 0; c < MIDL_P; c++) acc[c] = __builtin_fma(bv, Gk[c], acc[c]);
                }
                if (j < m) {
#pragma unroll
                    for (int c = 0; c < MIDL_P; c++)
                        if (c < p) Ep[c + (int64_t)(p + j) * fd.ldp] = -acc[c];
                }
            }
        }
    }
    __syncthreads();
    HIPMF_STAMP(blockIdx.x, 3);
    if (m == 0) return;
    // ---- S = F22 - W (rk-rows of F12): units of 64 rows x 8 columns dealt to the wavefronts; lane = row ----
    {
        const int ngr = (m + 7) >> 3, nun = ngr * nrb;
        for (int un = wave; un < nun; un += MIDL_NW) {
            const int rb = un / ngr, cs = (un - rb * ngr) * 8;
            const int ncol = (m - cs) < 8 ? (m - cs) : 8;
            const int i = rb * 64 + lane;
            const bool rowok = i < m;
            const int ic = rowok ? i : 0;
            double cin[8], acc[8];
#pragma unroll
            for (int c = 0; c < 8; c++) {
                cin[c] = F[(p + ic) + (int64_t)(p + cs + (c < ncol ? c : 0)) * ld];
                acc[c] = 0.0;
            }
            const double *Wi = Wp + ic;
            for (int k0 = 0; k0 < p; k0 += 4) {
                double wk[4];
                int pk[4];
#pragma unroll
                for (int kk = 0; kk < 4; kk++) {
                    const int k = k0 + kk < p ? k0 + kk : 0;
                    const double v = Wi[k * m];
                    wk[kk] = k0 + kk < p ? v : 0.0;
                    pk[kk] = wave_uniform(sh.rk[k]);
                }
#pragma unroll
                for (int kk = 0; kk < 4; kk++) {
                    const double *Bk = B12 + pk[kk] * ldb + cs;
#pragma unroll
                    for (int c = 0; c < 8; c++) acc[c] = __builtin_fma(wk[kk], Bk[c], acc[c]);
                }
            }
            if (rowok) {
#pragma unroll
                for (int c = 0; c < 8; c++)
                    if (c < ncol) F[(p + i) + (int64_t)(p + cs + c) * ld] = cin[c] - acc[c];
            }
        }
    }
    HIPMF_STAMP(blockIdx.x, 5);
}

} // namespace hipmf
