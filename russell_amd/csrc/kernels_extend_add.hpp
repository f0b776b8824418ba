// kernels_extend_add.hpp -- the extend-add of the LU fronts with first touch (round 4).
#pragma once
#include "kernels_assembly.hpp"
#include "kernels_factor.hpp"

namespace hipmf {

// extend-add with FIRST TOUCH (LU fronts, round 4): the task owns an EA_TILE_C-column x EA_TILE_R-row tile (32 x 64: sixteen KB of LDS; measured 6.93 ms of numeric LU with 32 x 256, 6.78 with 16 x 256, 6.73 with 32 x 128, 6.67 with 16 x 128, 6.65 with 32 x 64, 6.80 with 8 x 128, 6.95 with 8 x 64) of the parent's working block and builds it
// in LDS -- zero, the entries of A that land in the tile (per-task lists: what k_scatter did for the level), the children's contribution
// blocks in child order -- and writes the WHOLE tile once.  The order of the additions is k_zero + k_scatter + k_extend_add's: the same
// bits.  What it saves: the block was zero-filled (8 bytes per entry), then read and written once per child that hits an entry (16 bytes
// beside the 8 of the child's entry); now an entry costs 8 bytes per child + 8.  The zero-fill / scatter launches of the level and the
// side stream they ran on are gone with their cross-stream edge (5 - 7 us at each of the upper levels' boundaries).  Every tile of every
// big front of the level has a task (a tile no child touches is still zero + A).  Dynamic LDS: EA_TILE_C x EA_TILE_R doubles.
#ifndef HIPMF_EA_TILE_C
#define HIPMF_EA_TILE_C 32
#endif
#ifndef HIPMF_EA_TILE_R
#define HIPMF_EA_TILE_R 64
#endif
constexpr int EA_TILE_C = HIPMF_EA_TILE_C, EA_TILE_R = HIPMF_EA_TILE_R; // (EA_TILE_R: a power of two, 32 .. 256; EA_TILE_C >= 32: the first tile of a front holds its first diagonal tile)
static_assert(EA_TILE_C >= NB && EA_TILE_R >= NB && EA_TILE_R <= 256 && (EA_TILE_R & (EA_TILE_R - 1)) == 0 && EA_TILE_C * EA_TILE_R <= 65536, "extend-add tile");
__global__ void __launch_bounds__(256) k_extend_add_lds(const EaTask *__restrict__ tasks, const EaRange *__restrict__ ranges, const int32_t *__restrict__ rel,
                                                        double *__restrict__ pool, const int32_t *__restrict__ ea_sc, const int32_t *__restrict__ sc_k,
                                                        const uint16_t *__restrict__ sc_pos, const double *__restrict__ vs, const double *__restrict__ vs2,
                                                        double *__restrict__ dws, int32_t *__restrict__ lperm, const unsigned long long *__restrict__ anorm_bits,
                                                        double pivot_eps, FactorInfo *info, double *__restrict__ diag) {
    HIPMF_DYN_SHARED(double, T); // T[(c - c0) * EA_TILE_R + (r - r0)]
    const EaTask t = tasks[blockIdx.x];
    const int tid = threadIdx.x;
    const int s0 = ea_sc[blockIdx.x], s1 = ea_sc[blockIdx.x + 1];
    EaRange rg = {};
    if (t.piece_begin < t.piece_end) rg = ranges[t.piece_begin];
    for (int e = tid; e < t.nc * EA_TILE_R; e += 256) T[e] = 0.0;
    __syncthreads();
    for (int e = s0 + tid; e < s1; e += 256) {
        const int32_t k = sc_k[e];
        T[sc_pos[e]] = k < 0 ? vs2[~k] : vs[k];
    }
    __syncthreads();
    const int c0 = t.c0, r0 = t.r0;
    for (int pc = t.piece_begin; pc < t.piece_end; pc++) {
        const EaRange nxt = ranges[pc + 1 < t.piece_end ? pc + 1 : pc];
        const int64_t ldc = rg.ldc;
        const double *CB = pool + rg.cb_off;
        const int32_t *relc = rel + rg.rel_off;
        const int jlo = rg.jlo, jhi = rg.jhi, ilo = rg.ilo, ihi = rg.ihi;
        const int ni = ihi - ilo;
        const int sh = ni <= 16 ? 4 : (ni <= 32 ? 5 : 6);
        const int tx = tid & ((1 << sh) - 1), ty = tid >> sh, ng = 256 >> sh;
        for (int i = ilo + tx; i < ihi; i += (1 << sh)) {
            const int ri = relc[i] - r0;
            for (int j0 = jlo + ty; j0 < jhi; j0 += 8 * ng) {
                double cb[8];
                int at[8];
#pragma unroll
                for (int q = 0; q < 8; q++) {
                    const int j = j0 + q * ng;
                    at[q] = j < jhi ? ri + (relc[j] - c0) * EA_TILE_R : -1;
                }
#pragma unroll
                for (int q = 0; q < 8; q++)
                    if (at[q] >= 0) cb[q] = CB[i + (int64_t)(j0 + q * ng) * ldc];
#pragma unroll
                for (int q = 0; q < 8; q++)
                    if (at[q] >= 0) T[at[q]] += cb[q]; // (within one child the targets are distinct: rel is strictly increasing)
            }
        }
        __syncthreads(); // the next child may hit the same entries from other threads
        rg = nxt;
    }
    // The first tile of a tiled front now holds the front's first 32 x 32 diagonal tile: wavefront 0 factorises it here (what k_diag0 or
    // the first panel launch did, 18 - 22 us in front of every level's first panel step) while the other wavefronts write the tile out and
    // the other workgroups are still adding.  Result where the panel step expects it: dws buffer 0, interchanges, pivots.
    if (t.lu_slot >= 0 && tid < 64) {
        const int nb = t.lu_nb;
        double a[NB];
#pragma unroll
        for (int c = 0; c < NB; c++) a[c] = (tid < nb && c < nb) ? T[c * EA_TILE_R + tid] : (tid == c ? 1.0 : 0.0);
        const double eps = pivot_eps * __longlong_as_double((long long)*anorm_bits);
        int step, npert, nzero;
        tile_lu32<true>(a, tid, eps, step, npert, nzero);
        if (tid < nb) {
            double *dw = dws + (int64_t)t.lu_slot * NB * NB;
            double dg = 1.0;
#pragma unroll
            for (int c = 0; c < NB; c++) {
                if (c < nb) dw[step + c * nb] = a[c];
                if (c == step) dg = a[c];
            }
            lperm[t.lu_first + step] = tid;
            diag[t.lu_first + step] = dg;
        }
        if (tid == 0 && npert > 0) {
            atomicAdd(&info->n_perturbed, npert);
            if (nzero > 0) atomicAdd(&info->n_zero_pivot, nzero);
        }
    }
    double *F = pool + t.f_off + r0 + (int64_t)c0 * t.ld;
    for (int e = tid; e < t.nc * EA_TILE_R; e += 256) {
        const int c = e / EA_TILE_R, r = e % EA_TILE_R;
        if (r < t.nr) F[r + (int64_t)c * t.ld] = T[e];
    }
}

} // namespace hipmf
