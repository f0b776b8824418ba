// kernels_factor_chain.hpp -- the tiled steps of the levels near the root in ONE launch per level.
//
// Near the root a level holds a few large fronts, and its factorisation is a chain of dependent steps (panel solve -> trailing update +
// look-ahead LU of the next diagonal tile -> next panel solve ...) whose launches carry a few dozen workgroups each: 94 steps x 2
// launches x (launch hand-over + a latency chain of ~5 us) made 3.1 ms of the 7.3 ms of the 1000 x 1000 Poisson factorisation
// (profiles/r03_factor_sequence.txt).  k_chain runs the SAME tile bodies (panel_body / update_body, kernels_factor.hpp: same
// arithmetic in the same order, bit-identical factors) as tasks of one launch, in the order the launches had; a task waits for the
// counters of what it consumes and bumps the counters of what it produced:
//   cP[front][step]  panel tiles done;  cC[front][step]  critical update pieces done (the tiles that hold the next panel's block
//   column / block row, and the look-ahead piece);  cU[front][step]  all update pieces done.
//   panel(k)   waits for cC[k-1]            (its diagonal tile, its block column / row)
//   update(k)  waits for cP[k] and cU[k-1]  (the panels it multiplies; every earlier writer of its tile: the tiling shifts by 32 rows
//                                            per step, a tile overlaps up to four tiles of the step before)
// so the rest of a wide trailing update runs beside the next panel solve instead of in front of it.
// Workgroups are dispatched in index order and a task only waits for tasks with smaller indices: no task can wait for one that is not
// resident yet (the argument of the dependency-driven solves, kernels_solve_fused.hpp).  Data handed over inside the launch is written
// and read with agent-scope (sc1) accesses (TileMem<true>): the XCDs' L2s are not coherent with each other inside a launch.
// A wait that times out sets *err; the host then repeats the factorisation with one launch per step.
#pragma once
#include "kernels_factor.hpp"
#include "kernels_solve_fused.hpp"

namespace hipmf {

struct ChainTask {
    int32_t slot, k0, t, kind; // front (slot of the level's descriptor table), step, piece; kind 0: panel tile, 1: update piece
    int32_t w0, n0, w1, n1;    // counters to wait for (index, target); index < 0: none
    int32_t pub0, pub1;        // counters to bump when done (pub1 < 0: none)
    int32_t pad[2];
};

template <bool SYM>
__global__ void __launch_bounds__(256) k_chain(const ChainTask *__restrict__ tasks, const FrontDesc *__restrict__ LFD, double *__restrict__ pool,
                                               int32_t *__restrict__ lperm, double *__restrict__ dws, int32_t dws_stride,
                                               const unsigned long long *__restrict__ anorm_bits, double pivot_eps, FactorInfo *info,
                                               double *__restrict__ diag, int32_t pre_lu, int *cnt, int *err) {
    __shared__ union {
        PanelLds p;
        UpdateLds u;
    } sh;
    const ChainTask T = tasks[blockIdx.x];
    FrontDesc fd = LFD[T.slot];
    fd_resident(fd);
    if (threadIdx.x == 0) {
        if (T.w0 >= 0) sf_wait(cnt + T.w0, T.n0, err);
        if (T.w1 >= 0) sf_wait(cnt + T.w1, T.n1, err);
    }
    __syncthreads();
    if (T.kind == 0) panel_body<SYM, true, 256>(sh.p, T.slot, T.t, fd, T.k0, pool, lperm, dws, dws_stride, anorm_bits, pivot_eps, info, diag, pre_lu);
    else update_body<SYM, true>(sh.u, T.slot, T.t, fd, T.k0, pool, dws, dws_stride, lperm, anorm_bits, pivot_eps, info, diag);
    drain_stores();
    __syncthreads();
    if (threadIdx.x == 0) {
        flag_add(cnt + T.pub0, 1);
        if (T.pub1 >= 0) flag_add(cnt + T.pub1, 1);
    }
}

} // namespace hipmf
