// kernels_factor_binv.hpp -- tiled path, LU fronts, ONE launch per step (round 4): block elimination with the explicit inverse of the
// 32 x 32 diagonal tile.
//
// The step of kernels_factor.hpp is two dependent launches, k_panel (triangular solves of the block column and block row against the
// factorised diagonal tile) and k_update (trailing update + look-ahead LU of the next tile): 10 + 17 us (narrow) or 10 + 24 us (full)
// for a level with a handful of fronts, and the twelve levels near the root of a 2D mesh are ~115 such steps in sequence
// (profiles/r04_factor_sequence.txt: 5.2 of 7.1 ms at 1000 x 1000).  Nothing in a step is long except the two sequential 32-pivot pieces
// (the LU of the tile in the look-ahead workgroup, the substitutions in every panel workgroup), each behind its own launch boundary.
//
// Here the front is eliminated by blocks without triangular factors:  with D the (updated) diagonal tile of step k and G = inv(D),
//      W = A(:, k) G          block column below the tile (rows of F and rows of E')
//      R = A(k, :)            block row right of it, AS IT IS
//      A22 -= W R
// i.e. F = L~ U~ with L~ unit BLOCK lower triangular (blocks W) and U~ block upper triangular with the tiles D on its diagonal.  The
// augmented elimination of [F Ic; Ir 0] then leaves  E = [inv(L~11); -L~21 inv(L~11)]  and  E' = [inv(U~11), -inv(U~11) U~12]: the same
// block-triangular shapes (32-wide blocks) the solve kernels stream, the same pivots as an LU with the tile's pivot order (the
// in-place Gauss-Jordan of tile_inv32 chooses them like tile_lu32 does), determinant and interchange record unchanged.
// What goes: the panel launch.  W = A G is a 64 x 32 x 32 product every update tile forms for ITSELF on the matrix cores (16 MFMAs per
// slice of K beside the 32 of the tile's update), the block row needs no work at all, and the look-ahead workgroup forms its 32 rows
// of W, the next tile and its inverse.  A step is one launch whose length is the look-ahead chain:
//      load (1 round trip) -> W rows (512 FMAs per lane) -> tile update -> Gauss-Jordan (32 pivots) -> store.
// NOTHING is overwritten that another workgroup of the same launch reads: the block columns stay as they are in F (a temporary; every tile
// of a block row forms the same W from them), the trailing tiles are written by their owners only.  inv(D) of step k is E'(k, k) -- the
// identity rows of Ir times G -- and that is where the look-ahead workgroup puts it and where the tiles fetch it (no work buffer); a tile
// that holds those rows of E' takes them as the identity they were.  What is left when a level's steps are through is the part of E'
// above its diagonal blocks, still A(:, k) where the solves expect A(:, k) G: k_eflush multiplies it in place, one launch per level.
// Steps are grouped as before (narrow steps update the next panel's strips only; the last step of a group applies all its panels to
// the whole trailing matrix).
#pragma once
#include "kernels_factor.hpp"
#include "kernels_factor_front.hpp"

namespace hipmf {

constexpr int DV_LD = NB + 2; // LDS stride of inv(D): the layout of the U slice (Us[c][kk], stride 34)

template <int TS> struct BstepLdsT {
    static constexpr int LSLD = TS + 16;
    __attribute__((aligned(16))) double LsUM[(NB * LSLD > NB * (NB + 2)) ? NB * LSLD : NB * (NB + 2)];
    double Us[TS * US_LD];
    __attribute__((aligned(16))) double Dv[NB * DV_LD]; // Dv[j * DV_LD + k] = inv(D)(k, j)
    int32_t rk[NB];
};

// tile_inv32 (kernels_factor_front.hpp) as a LOOP: the unrolled form is ~30 KB of straight-line code, which is fine where thousands of
// workgroups run it (k_front_lu) and slow where ONE wavefront per launch runs it once -- every instruction is an instruction-cache miss
// (measured: 27 us for the inversion of one tile in k_dinv0).  Rotating form: the pivot column is always register 0, every other column
// moves one register down as it is updated, the column of the inverse that replaces the pivot column enters at register 31; after 32
// steps every column is back in its place.  The body (~110 instructions) is fetched once.  Always 32 steps: the caller pads a smaller
// block with identity rows / columns (their pivots are 1, chosen at their own steps).  Results as tile_inv32.
__device__ __forceinline__ void tile_inv32_rot(double (&a)[NB], int lane, double eps, double rep, int &step, double &dval, int32_t *rk, int &npert, int &nzero) {
    step = -1;
    dval = 1.0;
    npert = 0;
    nzero = 0;
#pragma clang loop unroll(disable)
    for (int c = 0; c < NB; c++) {
        const bool cand = lane < NB && step < 0;
        const unsigned mag = __float_as_uint((float)fabs(a[0]));
        const unsigned key = cand ? ((mag & ~63u) | 32u | (unsigned)(31 - lane)) : 0u;
        const double myinv = fast_rcp(a[0]);
        const int pv = 31 - (int)(wave_max_u32<2>(key) & 31u);
        double d = wave_bcast(a[0], pv);
        double inv = wave_bcast(myinv, pv);
        if (fabs(d) < eps || d == 0.0) {
            double dn = (d < 0.0) ? -rep : rep;
            if (dn == 0.0) dn = 1.0;
            npert++;
            if (d == 0.0) nzero++;
            d = dn;
            inv = 1.0 / dn;
        }
        if (lane == pv) step = c, dval = d, rk[c] = lane;
        const double lm = (lane == pv) ? 0.0 : a[0] * inv;
#pragma unroll
        for (int j = 1; j < NB; j++) a[j - 1] = a[j] - lm * wave_bcast(a[j], pv);
        a[NB - 1] = (lane == pv) ? 1.0 : -lm;
    }
    const double inv_own = 1.0 / dval;
#pragma unroll
    for (int cc = 0; cc < NB; cc++) a[cc] *= inv_own;
}

// first diagonal tile of every tiled front of a level: inv(D) -> E'(0, 0), pivots, interchanges (one wavefront per front)
__global__ void __launch_bounds__(64) k_dinv0(const FrontDesc *__restrict__ LFD, double *__restrict__ pool, int32_t *__restrict__ lperm,
                                              const unsigned long long *__restrict__ anorm_bits, double pivot_eps, FactorInfo *info,
                                              double *__restrict__ diag) {
    __shared__ int32_t rk[NB];
    const int slot = blockIdx.x, tid = threadIdx.x;
    FrontDesc fd = LFD[slot];
    const int nb = fd.p < NB ? fd.p : NB;
    const double *F = pool + fd.off;
    const int64_t ld = fd.ld;
    double a[NB];
#pragma unroll
    for (int c = 0; c < NB; c++) {
        const bool in = tid < nb && c < nb;
        const double v = F[(in ? tid : 0) + (int64_t)(in ? c : 0) * ld];
        a[c] = in ? v : (tid == c ? 1.0 : 0.0);
    }
    const double eps = pivot_eps * __longlong_as_double((long long)*anorm_bits);
    const double rep = pivot_replacement(pivot_eps, __longlong_as_double((long long)*anorm_bits));
    int step, npert, nzero;
    double dval;
    tile_inv32_rot(a, tid, eps, rep, step, dval, rk, npert, nzero);
    wave_sync();
    if (tid < nb) {
        double *Ep = pool + fd.epoff;
#pragma unroll
        for (int k = 0; k < NB; k++)
            if (k < nb) Ep[step + (int64_t)fd.ldp * wave_uniform(rk[k])] = a[k]; // inv(D)(step, rk[k]): row = pivot step, column = the row chosen at step k
        lperm[fd.first + step] = tid;
        diag[fd.first + step] = dval;
    }
    if (tid == 0 && npert > 0) {
        atomicAdd(&info->n_perturbed, npert);
        if (nzero > 0) atomicAdd(&info->n_zero_pivot, nzero);
    }
}

// Rows [rbase, rbase + 16 NRB) of the slice in Ls (Ls[k][r], stride LSLD: A(r, k)) become W = A inv(D) in place, by ONE wavefront that
// is the only reader and writer of those rows; Dv[j][k] = inv(D)(k, j).  MFMA operands as in the update: D^T on the A side.
// Wg != nullptr: rows [rlo, rhi) (tile-relative) of W also go to memory, (r, j) at Wg[r + j wstr], j < nbj.
template <int NRB, int LSLD>
__device__ __forceinline__ void form_w(double *Ls, const double *Dv, const int rbase, const int l15, const int l4, double *Wg = nullptr,
                                       const int64_t wstr = 0, const int rlo = 0, const int rhi = 0, const int nbj = 0) {
    f64x4 wacc[NRB][2];
#pragma unroll
    for (int rb = 0; rb < NRB; rb++)
#pragma unroll
        for (int jb = 0; jb < 2; jb++) wacc[rb][jb] = f64x4{0.0, 0.0, 0.0, 0.0};
#pragma unroll
    for (int kk0 = 0; kk0 < NB; kk0 += 4) {
        double da[2], ab[NRB];
#pragma unroll
        for (int jb = 0; jb < 2; jb++) da[jb] = Dv[(jb * 16 + l15) * DV_LD + kk0 + l4];
#pragma unroll
        for (int rb = 0; rb < NRB; rb++) ab[rb] = Ls[(kk0 + l4) * LSLD + rbase + rb * 16 + l15];
#pragma unroll
        for (int rb = 0; rb < NRB; rb++)
#pragma unroll
            for (int jb = 0; jb < 2; jb++) wacc[rb][jb] = mfma_f64_16x16x4(da[jb], ab[rb], wacc[rb][jb]);
    }
    wave_sync();
#pragma unroll
    for (int rb = 0; rb < NRB; rb++)
#pragma unroll
        for (int jb = 0; jb < 2; jb++)
#pragma unroll
            for (int g = 0; g < 4; g++) {
                const int rr = rbase + rb * 16 + l15, j = jb * 16 + l4 + 4 * g;
                Ls[j * LSLD + rr] = wacc[rb][jb][g];
                if (Wg != nullptr && rr >= rlo && rr < rhi && j < nbj) Wg[rr + (int64_t)j * wstr] = wacc[rb][jb][g];
            }
}

// One tile (t < ntiles) or the look-ahead piece (t == ntiles) of step k0 of the front in `slot`; tile enumeration, live-entry rules and
// the MFMA operand layouts are update_body's (kernels_factor.hpp), LU instance.
template <int TS>
__device__ __forceinline__ void bstep_body(BstepLdsT<TS> &sh, const int t, const FrontDesc &fd, int32_t k0, double *__restrict__ pool,
                                           int32_t *__restrict__ lperm, const unsigned long long *__restrict__ anorm_bits, double pivot_eps,
                                           FactorInfo *info, double *__restrict__ diag) {
    static_assert(TS == 64 || TS == 32, "tile edge");
    constexpr int NT = TS == 64 ? 256 : 64;
    constexpr int LSLD = BstepLdsT<TS>::LSLD;
    constexpr int MT = 2;
    constexpr int NE = TS * NB / NT;
    constexpr int ND = NB * NB / NT; // entries of inv(D) per thread
    double *Ls = sh.LsUM;
    double *Us = sh.Us;
    double *Dv = sh.Dv;
    const int tid = threadIdx.x;
    const int f = fd.p + fd.m;
    const int nb = (fd.p - k0) < NB ? (fd.p - k0) : NB;
    const int base = k0 + nb, limit = f + base;
    const int ntF = (f - base + TS - 1) / TS;
    const int nt = ntF + (base + TS - 1) / TS;
    const int nb2 = (fd.p - base) < NB ? (fd.p - base) : NB;
    const int gpos = (k0 / NB) % fd.ugroup;
    const bool narrow = gpos < fd.ugroup - 1 && nb2 > 0;
    const int nhalf = gpos + 1; // slices of K: the panels of the group so far; W of every one of them is formed here from A(:, k) and inv(D_k)
    const int kfirst = k0 - gpos * NB;
    const int ntiles = narrow ? 2 * nt : nt * nt;
    const AugView A = aug_view(fd, pool);
    double *F = A.F;
    double *Ep = pool + fd.epoff; // E'(i, c) at Ep[i + c p]; inv(D) of the step at column kh is its block (kh, kh)
    const int64_t pstr = fd.ldp;
    if (t == ntiles) {
        // ---- look-ahead workgroup (wave 0): rows [base, base + nb2) of W, the next diagonal tile, its inverse -> E'(base, base) ----
        if (tid >= 64) return;
#ifdef HIPMF_NO_LA
        return;
#endif
        // the piece is a 32 x 32 tile of the step on the matrix cores (W rows, then the update), then the inversion one row per lane
        const int l15 = tid & 15, l4 = tid >> 4;
        const double *Fb = F + base + (int64_t)base * A.ld; // the tile
        const uint32_t loC = (uint32_t)(l15 + (int64_t)l4 * A.ld);
        const uint32_t loL = (uint32_t)((tid & 31) + (int64_t)(tid >> 5) * A.ld), loD = (uint32_t)((tid & 31) + (int64_t)(tid >> 5) * pstr);
        f64x4 acc[MT][MT]; // acc = W R - tile
#pragma unroll
        for (int a = 0; a < MT; a++)
#pragma unroll
            for (int b = 0; b < MT; b++)
#pragma unroll
                for (int g = 0; g < 4; g++) {
                    const int r = b * 16 + l15, c = a * 16 + l4 + 4 * g;
                    const bool in = r < nb2 && c < nb2;
                    const double *Cu = Fb + b * 16 + (int64_t)(a * 16 + 4 * g) * A.ld;
                    const double v = in ? Cu[loC] : 0.0;
                    acc[a][b][g] = in ? -v : (r == c ? -1.0 : 0.0); // identity padding
                }
        for (int h = 0; h < nhalf; h++) {
            const int kh = kfirst + h * NB, nbh = (h == nhalf - 1) ? nb : NB;
            if (h > 0) wave_sync();
            const double *Dh = Ep + kh + (int64_t)kh * pstr;
            double lr[16], ur[16], dr[16];
#pragma unroll
            for (int u = 0; u < 16; u++) {
                const int e = tid + 64 * u;
                const int lo = e & 31, hi = e >> 5;
                const bool lin = lo < nb2 && hi < nbh, uin = lo < nbh && hi < nb2, din = h == nhalf - 1 && lo < nbh && hi < nbh;
                const double *Lu = F + base + (int64_t)(kh + 2 * u) * A.ld, *Uu = F + kh + (int64_t)(base + 2 * u) * A.ld, *Du = Dh + (int64_t)(2 * u) * pstr;
                lr[u] = lin ? Lu[loL] : 0.0; // A(row base + lo, column kh + hi)
                ur[u] = uin ? Uu[loL] : 0.0; // A(row kh + lo, column base + hi)
                dr[u] = din ? Du[loD] : 0.0;
            }
#pragma unroll
            for (int u = 0; u < 16; u++) {
                const int e = tid + 64 * u;
                const int lo = e & 31, hi = e >> 5;
                Ls[hi * LSLD + lo] = lr[u];
                Us[hi * US_LD + lo] = ur[u];
                Dv[hi * DV_LD + lo] = dr[u];
            }
            wave_sync();
            if (h == nhalf - 1) { // (the earlier panels of the group: their steps left W in place of these rows of A)
                form_w<2, LSLD>(Ls, Dv, 0, l15, l4);
                wave_sync();
            }
#pragma unroll
            for (int kk0 = 0; kk0 < NB; kk0 += 4) {
                double ua[MT], lb[MT];
#pragma unroll
                for (int a = 0; a < MT; a++) ua[a] = Us[(a * 16 + l15) * US_LD + kk0 + l4];
#pragma unroll
                for (int b = 0; b < MT; b++) lb[b] = Ls[(kk0 + l4) * LSLD + b * 16 + l15];
#pragma unroll
                for (int a = 0; a < MT; a++)
#pragma unroll
                    for (int b = 0; b < MT; b++) acc[a][b] = mfma_f64_16x16x4(ua[a], lb[b], acc[a][b]);
            }
        }
        // the tile, one row per lane (lanes 32 .. 63 hold copies): through LDS
        wave_sync();
#pragma unroll
        for (int a = 0; a < MT; a++)
#pragma unroll
            for (int b = 0; b < MT; b++)
#pragma unroll
                for (int g = 0; g < 4; g++) Us[(a * 16 + l4 + 4 * g) * US_LD + b * 16 + l15] = -acc[a][b][g];
        wave_sync();
        double a2[NB];
#pragma unroll
        for (int c = 0; c < NB; c++) a2[c] = Us[c * US_LD + (tid & 31)];
        const double eps = pivot_eps * __longlong_as_double((long long)*anorm_bits);
        const double rep = pivot_replacement(pivot_eps, __longlong_as_double((long long)*anorm_bits));
        int step, npert, nzero;
        double dval;
        tile_inv32_rot(a2, tid, eps, rep, step, dval, sh.rk, npert, nzero); // lanes >= 32 are not candidates and take no part
        wave_sync();
        if (tid < nb2) {
            double *Dn = Ep + base + (int64_t)base * pstr;
#pragma unroll
            for (int k = 0; k < NB; k++)
                if (k < nb2) Dn[step + pstr * wave_uniform(sh.rk[k])] = a2[k];
            lperm[fd.first + base + step] = base + tid;
            diag[fd.first + base + step] = dval;
        }
        if (tid == 0 && npert > 0) {
            atomicAdd(&info->n_perturbed, npert);
            if (nzero > 0) atomicAdd(&info->n_zero_pivot, nzero);
        }
        return;
    }
    if (t > ntiles) return;
    // tile of this workgroup; a narrow step has the nt tiles of the block column, then the nt tiles of the block row
    const bool rowstrip = narrow && t >= nt;
    const int ti = narrow ? (rowstrip ? 0 : t) : t % nt;
    const int tj = narrow ? (rowstrip ? t - nt : 0) : t / nt;
    const bool rowsE = ti >= ntF, colsE = tj >= ntF;
    const int r0 = rowsE ? f + (ti - ntF) * TS : base + ti * TS, c0 = colsE ? f + (tj - ntF) * TS : base + tj * TS;
    const int rend = rowsE ? limit : f, cend = colsE ? limit : f;
    if (r0 >= rend || c0 >= cend) return;
    if (rowsE && colsE) return; // corner of the augmented front: never read
    const int rmax = rowstrip ? base + nb2 : rend;
    const int cmax = (narrow && !rowstrip) ? base + nb2 : cend;
    const int cmin = rowstrip ? base + nb2 : 0;
    const int lane = tid & 63, wave = tid >> 6;
    const int wr = TS == 64 ? (wave & 1) * 32 : 0, wc = TS == 64 ? (wave >> 1) * 32 : 0;
    const int l15 = lane & 15, l4 = lane >> 4;
    double lreg[NE], ureg[NE], dvreg[ND];
    double *Lb = rowsE ? A.Epsh : F;
    const int64_t lstr = rowsE ? A.ps : A.ld;
    const double *Ub = colsE ? A.Esh : F;
    double *Cb = rowsE ? A.Epsh : (colsE ? A.Esh : F);
    const int64_t cstr = rowsE ? A.ps : A.ld;
    // slice h: the block column A(rows of the tile, kh ..) as it is (rows kh .. of E' hold inv(D) where the identity of Ir was: taken as
    // the identity), the block row A(kh .., columns of the tile), inv(D_kh)
    // (addresses: a 64-bit base that is the same for the whole workgroup -- scalar registers -- plus ONE 32-bit per-lane offset per array,
    //  shared by all loads of the array: the per-load 64-bit lane addresses of the straightforward form cost ~100 registers here)
    const uint32_t loffL = (uint32_t)((tid % TS) + (int64_t)(tid / TS) * lstr);
    const uint32_t loffU = (uint32_t)((tid % NB) + (int64_t)(tid / NB) * A.ld);
    const uint32_t loffD = (uint32_t)((tid % NB) + (int64_t)(tid / NB) * pstr);
#define HIPMF_FETCH_SLICE(h)                                                                                           \
    {                                                                                                                  \
        const int kh = kfirst + (h) * NB, nbh = ((h) == nhalf - 1) ? nb : NB;                                          \
        _Pragma("unroll") for (int u = 0; u < NE; u++) {                                                               \
            const int e = tid + NT * u;                                                                                \
            const int r = e % TS, kk = e / TS;                                                                         \
            const int k2 = e % NB, c = c0 + e / NB;                                                                    \
            const int ie = r0 + r - f - kh; /* (rows of E': position inside the step's identity block) */             \
            const bool ident = (h) == nhalf - 1 && rowsE && ie >= 0 && ie < nbh;                                       \
            const bool lin = r0 + r < rmax && kk < nbh && !ident, uin = c < cend && k2 < nbh;                          \
            const double *Lu = Lb + r0 + (int64_t)(kh + (NT / TS) * u) * lstr;                                         \
            const double *Uu = Ub + kh + (int64_t)(c0 + (NT / NB) * u) * A.ld;                                         \
            const double lv = lin ? Lu[loffL] : 0.0;                                                                   \
            lreg[u] = ident ? (ie == kk ? 1.0 : 0.0) : lv;                                                             \
            ureg[u] = uin ? Uu[loffU] : 0.0;                                                                           \
        }                                                                                                              \
        const double *Dh = Ep + kh + (int64_t)kh * pstr;                                                               \
        _Pragma("unroll") for (int u = 0; u < ND; u++) {                                                               \
            const int e = tid + NT * u;                                                                                \
            const bool in = (h) == nhalf - 1 && (e & 31) < nbh && (e >> 5) < nbh;                                      \
            const double *Du = Dh + (int64_t)((NT / NB) * u) * pstr;                                                   \
            dvreg[u] = in ? Du[loffD] : 0.0;                                                                           \
        }                                                                                                              \
    }
#define HIPMF_STORE_SLICE()                                                                                            \
    {                                                                                                                  \
        _Pragma("unroll") for (int u = 0; u < NE; u++) {                                                               \
            const int e = tid + NT * u;                                                                                \
            Ls[(e / TS) * LSLD + e % TS] = lreg[u];                                                                    \
            Us[(e / NB) * US_LD + e % NB] = ureg[u];                                                                   \
        }                                                                                                              \
        _Pragma("unroll") for (int u = 0; u < ND; u++) {                                                               \
            const int e = tid + NT * u;                                                                                \
            Dv[(e >> 5) * DV_LD + (e & 31)] = dvreg[u];                                                                \
        }                                                                                                              \
    }
    HIPMF_FETCH_SLICE(0)
    const bool owner0 = t == 0 && nb2 > 0 && wave == 0; // this wave's block holds the next diagonal tile (the look-ahead workgroup's)
    auto is_live = [&](int a, int b, int g) {
        const int r = r0 + wr + b * 16 + l15, c = c0 + wc + a * 16 + l4 + 4 * g;
        const bool corner = owner0 && (b * 16 + l15) < nb2 && (a * 16 + l4 + 4 * g) < nb2;
        return r < rmax && c < cmax && c >= cmin && !corner;
    };
    const uint32_t loffC = (uint32_t)((wr + l15) + (int64_t)(wc + l4) * cstr);
    // the tile's entries are requested with the first slice and go straight into the accumulators: acc = W R - C
    f64x4 acc[MT][MT];
#pragma unroll
    for (int a = 0; a < MT; a++)
#pragma unroll
        for (int b = 0; b < MT; b++)
#pragma unroll
            for (int g = 0; g < 4; g++) {
                const double *Cu = Cb + (r0 + b * 16) + (int64_t)(c0 + a * 16 + 4 * g) * cstr;
                acc[a][b][g] = is_live(a, b, g) ? -Cu[loffC] : 0.0;
            }
    HIPMF_STORE_SLICE()
    __syncthreads();
    if (nhalf > 1) HIPMF_FETCH_SLICE(1) // in flight while the first slice is worked on
    // The column-strip tile of a narrow step leaves its rows of W where A(:, k0 ..) was: the later steps of the group read W there, and so
    // do the solves (rows of E').  Not the rows of the next diagonal tile (the row-strip tiles and the look-ahead workgroup of THIS launch
    // read them as A; no later step needs them as W), not the rows of E' that hold inv(D) (they are their own W).
    double *const Wg = (narrow && !rowstrip) ? Lb + r0 + (int64_t)k0 * lstr : nullptr;
    const int wlo = rowsE ? 0 : base + nb2 - r0, whi = rowsE ? f + k0 - r0 : rend - r0;
    if (nhalf == 1) {
        form_w<TS == 64 ? 1 : 2, LSLD>(Ls, Dv, TS == 64 ? wave * 16 : 0, l15, l4, Wg, lstr, wlo, whi, nb);
        __syncthreads();
    }
    const bool wave_idle = (c0 + wc >= cmax) || (c0 + wc + 32 <= cmin) || (r0 + wr >= rmax);
    for (int h = 0; h < nhalf; h++) {
        if (h > 0) {
            __syncthreads();
            HIPMF_STORE_SLICE()
            __syncthreads();
            if (h + 1 < nhalf) HIPMF_FETCH_SLICE(h + 1)
            else {
                form_w<TS == 64 ? 1 : 2, LSLD>(Ls, Dv, TS == 64 ? wave * 16 : 0, l15, l4, Wg, lstr, wlo, whi, nb);
                __syncthreads();
            }
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
                double *Cu = Cb + (r0 + b * 16) + (int64_t)(c0 + a * 16 + 4 * g) * cstr;
                if (is_live(a, b, g)) Cu[loffC] = -acc[a][b][g];
            }
}

__global__ void __launch_bounds__(256, 2) k_bstep(const int32_t *__restrict__ pfx, int32_t nactive, const FrontDesc *__restrict__ LFD, int32_t k0,
                                               double *__restrict__ pool, int32_t *__restrict__ lperm, const unsigned long long *__restrict__ anorm_bits,
                                               double pivot_eps, FactorInfo *info, double *__restrict__ diag) {
    __shared__ BstepLdsT<UPD_T> sh;
    int pfx_slot;
    const int slot = find_slot_pfx(pfx, nactive, blockIdx.x, pfx_slot);
    const int t = blockIdx.x - pfx_slot;
    FrontDesc fd = LFD[slot];
    fd_resident(fd);
    bstep_body<UPD_T>(sh, t, fd, k0, pool, lperm, anorm_bits, pivot_eps, info, diag);
}

__global__ void __launch_bounds__(64, 2) k_bstep32(const int32_t *__restrict__ pfx, int32_t nactive, const FrontDesc *__restrict__ LFD, int32_t k0,
                                                double *__restrict__ pool, int32_t *__restrict__ lperm, const unsigned long long *__restrict__ anorm_bits,
                                                double pivot_eps, FactorInfo *info, double *__restrict__ diag) {
    __shared__ BstepLdsT<UPD_T_SMALL> sh;
    int pfx_slot;
    const int slot = find_slot_pfx(pfx, nactive, blockIdx.x, pfx_slot);
    const int t = blockIdx.x - pfx_slot;
    FrontDesc fd = LFD[slot];
    fd_resident(fd);
    bstep_body<UPD_T_SMALL>(sh, t, fd, k0, pool, lperm, anorm_bits, pivot_eps, info, diag);
}

// After a level's steps: the part of E' above its diagonal blocks is still A(:, k); the solves read W = A(:, k) inv(D_k).  One wavefront
// per 64 rows x one block column: lane = row, the row's 32 entries in registers, inv(D_k) = E'(k, k) through LDS (broadcast reads).
// Task t of a front: block column t / nrt, row tile t % nrt (nrt = tiles of 64 rows of the p rows; tiles at or below the diagonal block
// have nothing to do).
__global__ void __launch_bounds__(64) k_eflush(const int32_t *__restrict__ pfx, int32_t nfronts, const FrontDesc *__restrict__ LFD,
                                               double *__restrict__ pool) {
    __shared__ double Dv[NB * DV_LD];
    int pfx_slot;
    const int slot = find_slot_pfx(pfx, nfronts, blockIdx.x, pfx_slot);
    const int t = blockIdx.x - pfx_slot;
    const FrontDesc fd = LFD[slot];
    const int p = fd.p, nrt = (p + 63) >> 6;
    const int kb = t / nrt, rt = t - kb * nrt;
    const int k0 = kb * NB, i0 = rt * 64;
    if (i0 >= k0) return;
    if (k0 + NB < p && (kb % fd.ugroup) != fd.ugroup - 1) return; // a narrow step: its column-strip tiles stored W
    const int nbk = (p - k0) < NB ? (p - k0) : NB;
    const int lane = threadIdx.x;
    double *Ep = pool + fd.epoff;
    const int64_t pstr = fd.ldp;
    const double *Dk = Ep + k0 + (int64_t)k0 * pstr;
    const int i = i0 + lane;
    const bool rowok = i < k0;
    double *Ei = Ep + (rowok ? i : 0) + (int64_t)k0 * pstr;
    double a[NB];
#pragma unroll
    for (int kk = 0; kk < NB; kk++) {
        const double v = Ei[(int64_t)(kk < nbk ? kk : 0) * pstr];
        a[kk] = (rowok && kk < nbk) ? v : 0.0;
    }
#pragma unroll
    for (int u = 0; u < NB * NB / 64; u++) {
        const int e = lane + 64 * u;
        const bool in = (e & 31) < nbk && (e >> 5) < nbk;
        const double v = Dk[(in ? (e & 31) : 0) + (int64_t)(in ? (e >> 5) : 0) * pstr];
        Dv[(e >> 5) * DV_LD + (e & 31)] = in ? v : 0.0;
    }
    wave_sync();
    // (the row is in registers: a column of W can go straight to where the column of A was)
#pragma unroll 1
    for (int j = 0; j < nbk; j++) {
        const double *dcol = Dv + j * DV_LD;
        double w = 0.0;
#pragma unroll
        for (int kk = 0; kk < NB; kk++) w = __builtin_fma(a[kk], dcol[kk], w);
        if (rowok) Ei[(int64_t)j * pstr] = w;
    }
}

} // namespace hipmf
