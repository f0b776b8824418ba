// kernels_extend_add.hpp -- the extend-add of the LU fronts with first touch (round 4).
#pragma once
#include "kernels_assembly.hpp"
#include "kernels_factor.hpp"

namespace hipmf {

// extend-add with FIRST TOUCH (round 4; SYM: L D L^T fronts -- only entries on or below the parent's diagonal are added, the tiles strictly
// above it have no task and are never read): the task owns an EA_TILE_C-column x EA_TILE_R-row tile (32 x 64: sixteen KB of LDS; measured 6.93 ms of numeric LU with 32 x 256, 6.78 with 16 x 256, 6.73 with 32 x 128, 6.67 with 16 x 128, 6.65 with 32 x 64, 6.80 with 8 x 128, 6.95 with 8 x 64) of the parent's working block and builds it
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
template <bool SYM, bool PAIRED = false>
__global__ void __launch_bounds__(256) k_extend_add_lds(const EaTask *__restrict__ tasks, const EaRange *__restrict__ ranges, const int32_t *__restrict__ rel,
                                                        double *__restrict__ pool, const int32_t *__restrict__ ea_sc, const int32_t *__restrict__ sc_k,
                                                        const uint16_t *__restrict__ sc_pos, const double *__restrict__ vs, const double *__restrict__ vs2,
                                                        double *__restrict__ dws, int32_t *__restrict__ lperm, const unsigned long long *__restrict__ anorm_bits,
                                                        double pivot_eps, FactorInfo *info, double *__restrict__ diag) {
    HIPMF_DYN_SHARED(double, T); // T[(c - c0) * EA_TILE_R + (r - r0)]
    const EaTask t = tasks[blockIdx.x];
    const int tid = threadIdx.x;
    const int s0 = ea_sc[blockIdx.x], s1 = ea_sc[blockIdx.x + 1];
    const int c0 = t.c0, r0 = t.r0;
    // A child's part of the tile: at most EA_TILE_R rows x EA_TILE_C columns of its contribution block.  Thread = (row tx, column group ty),
    // EA_PQ columns each.  The children are taken two at a time and the loads of a pair -- descriptors, then the relative indices of both,
    // then the entries of both -- are issued together, the first pair BEFORE the tile is zeroed and A's entries are placed: a workgroup is
    // a chain of dependent round trips to memory (2 us each) and used to make two per child, one child after the other.
    constexpr int EA_NG = 256 / EA_TILE_R, EA_PQ = (EA_TILE_C + EA_NG - 1) / EA_NG;
    static_assert(EA_TILE_R <= 256 && EA_PQ <= 16, "extend-add thread map");
    const int tx = tid % EA_TILE_R, ty = tid / EA_TILE_R;
    double cbA[EA_PQ], cbB[EA_PQ];
    int atA[EA_PQ], atB[EA_PQ];
    auto load_pair = [&](int pc) {
        const bool hasA = pc < t.piece_end, hasB = pc + 1 < t.piece_end;
        EaRange ra = {}, rb = {};
        if (hasA) ra = ranges[pc];
        if (hasB) rb = ranges[pc + 1];
        const int32_t *relA = rel + ra.rel_off, *relB = rel + rb.rel_off;
        const int iA = ra.ilo + tx, iB = rb.ilo + tx;
        const bool okA = hasA && iA < ra.ihi, okB = hasB && iB < rb.ihi;
        // (which entries exist is known from the ranges alone: the entries are requested together with the relative indices, not after them)
        int riA = 0, riB = 0, rjA[EA_PQ], rjB[EA_PQ];
        if (okA) riA = relA[iA];
        if (okB) riB = relB[iB];
        const double *CA = pool + ra.cb_off + iA, *CBb = pool + rb.cb_off + iB;
#pragma unroll
        for (int q = 0; q < EA_PQ; q++) {
            const int jA = ra.jlo + ty + q * EA_NG, jB = rb.jlo + ty + q * EA_NG;
            // (SYM: a child's block is valid on and below its diagonal; rel is increasing, so those entries land on or below the parent's)
            const bool inA = okA && jA < ra.jhi && (!SYM || jA <= iA), inB = okB && jB < rb.jhi && (!SYM || jB <= iB);
            rjA[q] = inA ? relA[jA] : -1;
            rjB[q] = inB ? relB[jB] : -1;
            cbA[q] = inA ? CA[(int64_t)jA * ra.ldc] : 0.0;
            cbB[q] = inB ? CBb[(int64_t)jB * rb.ldc] : 0.0;
        }
#pragma unroll
        for (int q = 0; q < EA_PQ; q++) {
            atA[q] = rjA[q] >= 0 ? (riA - r0) + (rjA[q] - c0) * EA_TILE_R : -1;
            atB[q] = rjB[q] >= 0 ? (riB - r0) + (rjB[q] - c0) * EA_TILE_R : -1;
        }
    };
    load_pair(t.piece_begin);
    for (int e = tid; e < t.nc * EA_TILE_R; e += 256) T[e] = 0.0;
    __syncthreads();
    for (int e = s0 + tid; e < s1; e += 256) {
        const int32_t k = sc_k[e];
        T[sc_pos[e]] = k < 0 ? vs2[~k] : vs[k];
    }
    __syncthreads();
    for (int pc = t.piece_begin; pc < t.piece_end; pc += 2) {
        if (pc > t.piece_begin) load_pair(pc);
#pragma unroll
        for (int q = 0; q < EA_PQ; q++)
            if (atA[q] >= 0) T[atA[q]] += cbA[q]; // (within one child the targets are distinct: rel is strictly increasing)
        __syncthreads(); // the next child may hit the same entries from other threads
        if (pc + 1 < t.piece_end) {
#pragma unroll
            for (int q = 0; q < EA_PQ; q++)
                if (atB[q] >= 0) T[atB[q]] += cbB[q];
            __syncthreads();
        }
    }
    // The first tile of a tiled front now holds the front's first 32 x 32 diagonal tile: wavefront 0 factorises it here (what k_diag0 or
    // the first panel launch did, 18 - 22 us in front of every level's first panel step) while the other wavefronts write the tile out and
    // the other workgroups are still adding.  Result where the panel step expects it: dws buffer 0, interchanges, pivots.
    if (t.lu_slot >= 0 && tid < 64) {
        const int nb = t.lu_nb;
        double a[NB];
#pragma unroll
        for (int c = 0; c < NB; c++) {
            const int rr = (SYM && tid < c) ? c : tid, cc = (SYM && tid < c) ? tid : c; // (SYM: only the lower triangle is assembled)
            a[c] = (tid < nb && c < nb) ? T[cc * EA_TILE_R + rr] : (tid == c ? 1.0 : 0.0);
        }
        const double eps = pivot_eps * __longlong_as_double((long long)*anorm_bits);
        const double rep = pivot_replacement(pivot_eps, __longlong_as_double((long long)*anorm_bits));
        int step, npert, nzero;
        double zr = 0.0, zi = 0.0;
        tile_lu32_z<!SYM, PAIRED>(a, tid, eps, rep, step, npert, nzero, zr, zi);
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
            store_zpivot<PAIRED>(info, t.lu_first, step, zr, zi);
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
