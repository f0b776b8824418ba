// kernels_solve.hpp -- level-set sparse triangular solves in multifrontal form (HBM-bound).
//   small fronts (f <= SMALL_F): k_fwd / k_bwd, one wavefront per supernode, panel staged in LDS, substitution by wave shuffles
//   big fronts (augmented):      k_fwd_big / k_bwd_big, GEMVs against the inverse-based panels E / E'
//                                split into row slabs, one workgroup per slab: 64 rows x 4 column groups (256 threads),
//                                or 32 rows x 32 column groups (1024 threads) on the levels near the root, where
//                                the fronts are few and large and the per-thread column chain sets the latency
// Forward pass, leaves to root:   w = [b1; 0] + sum_children u_c;  y1 = L11^{-1} P w1;  u = w2 - L21 y1
// Backward pass, root to leaves:  x1 = U11^{-1} (y1 - U12 x2),  x2 gathered from the ancestors
// Every sum has a fixed order (children ascending, columns ascending, 4 fixed column groups), so the
// solves are bit-reproducible.
#pragma once
#include "kernels_common.hpp"

namespace hipmf {

// Small fronts (f <= SMALL_F = 64), one wavefront per supernode.  The factor panel the step needs is copied
// to LDS first with every load in flight at once (the substitution would otherwise pay one HBM round trip per
// column); all arithmetic then runs out of LDS.  Dynamic LDS: ldp * pmax doubles (ldp = fmax | 1).
__global__ void __launch_bounds__(64) k_fwd(const int32_t *__restrict__ list, const FrontDesc *__restrict__ FD,
                                            const double *__restrict__ pool, const int32_t *__restrict__ lperm,
                                            const int32_t *__restrict__ child_idx, const int32_t *__restrict__ rel,
                                            double *__restrict__ work, double *__restrict__ x, int32_t ldp) {
    HIPMF_DYN_SHARED(double, P); // P[i + j * ldp] = F(i, j), i < f, j < p  (L11 and L21)
    __shared__ double w[SMALL_F];
    const int tid = threadIdx.x;
    FrontDesc fd = FD[list[blockIdx.x]];
    const int p = fd.p, f = fd.p + fd.m;
    const double *F = pool + fd.off;
    double *W = work + fd.woff;
    double *xs = x + fd.first;
    if (tid < f)
        for (int j = 0; j < p; j++) P[tid + j * ldp] = F[tid + (int64_t)j * f];
    w[tid] = (tid < p) ? xs[tid] : 0.0;
    __syncthreads();
    for (int ci = fd.child_begin; ci < fd.child_end; ci++) {
        FrontDesc cd = FD[child_idx[ci]];
        const double *uc = work + cd.woff + cd.p;
        const int32_t *relc = rel + cd.rowptr;
        for (int i = tid; i < cd.m; i += 64) w[relc[i]] += uc[i];
        __syncthreads();
    }
    // row interchanges of the pivot block, then y1 = L11^{-1} (P w1) column by column, u = w2 - L21 y1
    double v = (tid < p) ? w[lperm[fd.first + tid]] : ((tid < f) ? w[tid] : 0.0);
    for (int j = 0; j < p; j++) {
        const double vj = wave_bcast(v, j);
        if (tid > j && tid < f) v -= P[tid + j * ldp] * vj;
    }
    if (tid < p) xs[tid] = v;
    else if (tid < f) W[tid] = v;
}

__global__ void __launch_bounds__(64) k_bwd(const int32_t *__restrict__ list, const FrontDesc *__restrict__ FD,
                                            const double *__restrict__ pool, const int32_t *__restrict__ rows,
                                            double *__restrict__ work, double *__restrict__ x, int32_t ldp) {
    HIPMF_DYN_SHARED(double, P); // P[i + j * ldp] = U11(i, j), i, j < p
    __shared__ double xg[SMALL_F]; // x2 gathered from the ancestors
    const int tid = threadIdx.x;
    FrontDesc fd = FD[list[blockIdx.x]];
    const int p = fd.p, m = fd.m, f = fd.p + fd.m;
    const double *F = pool + fd.off;
    double *xs = x + fd.first;
    const int32_t *rws = rows + fd.rowptr;
    // lanes are arranged as (row i, column group jq): pw = p rounded up to 16 / 32 / 64 lanes per column
    const int sh = p <= 16 ? 4 : (p <= 32 ? 5 : 6);
    const int i = tid & ((1 << sh) - 1), jq = tid >> sh, ng = 64 >> sh;
    // (the rows of U come from the packed p x f copy when the front has one -- m > 0 --, else from the front itself: same values)
    const double *Ub = fd.epoff >= 0 ? pool + fd.epoff : F;
    const int64_t us = fd.epoff >= 0 ? p : f;
    if (tid < m) xg[tid] = x[rws[tid]];
    for (int j = jq; j < p; j += ng)
        if (i < p) P[i + j * ldp] = Ub[i + (int64_t)j * us];
    __syncthreads();
    // t = y1 - U12 x2: U12 is streamed from HBM exactly once, ng columns per pass, 4 passes in flight
    double acc = 0.0;
    if (i < p) {
        const double *Ui = Ub + i + (int64_t)p * us;
        int j = jq;
        for (; j + 3 * ng < m; j += 4 * ng) {
            const double e0 = Ui[(int64_t)j * us], e1 = Ui[(int64_t)(j + ng) * us], e2 = Ui[(int64_t)(j + 2 * ng) * us], e3 = Ui[(int64_t)(j + 3 * ng) * us];
            acc += e0 * xg[j];
            acc += e1 * xg[j + ng];
            acc += e2 * xg[j + 2 * ng];
            acc += e3 * xg[j + 3 * ng];
        }
        for (; j < m; j += ng) acc += Ui[(int64_t)j * us] * xg[j];
    }
    for (int off = 1 << sh; off < 64; off <<= 1) acc += __shfl_xor(acc, off); // sum over the column groups (fixed order)
    double v = (tid < p) ? xs[tid] - acc : 0.0;
    // x1 = U11^{-1} t: columns from right to left among lanes 0..p-1.  The reciprocal of the lane's own pivot is formed once: a
    // double-precision division per pivot (a dozen quarter-rate instructions executed by the whole wavefront) was the larger part
    // of this loop's instruction count.
    const double inv_d = (tid < p) ? 1.0 / P[tid + tid * ldp] : 1.0;
    for (int j = p - 1; j >= 0; j--) {
        if (tid == j) v *= inv_d;
        const double vj = wave_bcast(v, j);
        if (tid < j) v -= P[tid + j * ldp] * vj;
    }
    if (tid < p) xs[tid] = v;
    (void)work;
}

// 8-way unrolled strided dot product: acc += sum_j col[j * ld] * w[j], j = j0, j0 + step, ... < j1.
// Eight independent loads are in flight per lane before the first FMA (the loop is HBM-latency-bound otherwise).
__device__ __forceinline__ double strided_dot(const double *__restrict__ col, int64_t ld, const double *w, int j0, int j1, int step) {
    double acc0 = 0.0, acc1 = 0.0;
    int j = j0;
    for (; j + 7 * step < j1; j += 8 * step) {
        double e[8];
#pragma unroll
        for (int u = 0; u < 8; u++) e[u] = col[(int64_t)(j + u * step) * ld];
#pragma unroll
        for (int u = 0; u < 8; u += 2) {
            acc0 += e[u] * w[j + u * step];
            acc1 += e[u + 1] * w[j + (u + 1) * step];
        }
    }
    for (; j < j1; j += step) acc0 += col[(int64_t)j * ld] * w[j];
    return acc0 + acc1;
}

// Sum of the G column-group partial sums of row rr, pairwise in a fixed order.
template <int SLAB, int G>
__device__ __forceinline__ double group_sum(const double (*red)[SLAB], int rr) {
    double t[G];
#pragma unroll
    for (int g = 0; g < G; g++) t[g] = red[g][rr];
#pragma unroll
    for (int w = 1; w < G; w <<= 1)
#pragma unroll
        for (int g = 0; g + w < G; g += 2 * w) t[g] += t[g + w];
    return t[0];
}

// Forward step of a big (augmented) front, rows [r0, r1) of its f-vector:
//   [y1; -delta] = E * w1,  E(r, j) = E[r + j f]   ->   work[r] = y1[r] (r < p),  work[r] = w2[r] + (E w1)[r] (r >= p)
// Every workgroup of the front assembles w1 = b1 + (children's updates to the pivot rows) in LDS itself;
// the children's entries are swept linearly (no searches), in child order.  Dynamic LDS: p doubles.
template <int SLAB, int G>
__global__ void __launch_bounds__(SLAB *G) k_fwd_big(const SolveTask *__restrict__ tasks, const FrontDesc *__restrict__ FD,
                                                 const double *__restrict__ pool, const int32_t *__restrict__ child_idx,
                                                 const int32_t *__restrict__ rel, double *__restrict__ work,
                                                 const double *__restrict__ x) {
    HIPMF_DYN_SHARED(double, w1);
    __shared__ double wsl[SLAB];
    __shared__ double red[G][SLAB];
    constexpr int T = SLAB * G;
    const int tid = threadIdx.x;
    SolveTask tk = tasks[blockIdx.x];
    FrontDesc fd = FD[tk.s];
    const int p = fd.p, f = fd.p + fd.m;
    const int64_t ld = fd.ld;
    const double *E = pool + fd.eoff;
    double *W = work + fd.woff;
    const int r0 = tk.r0, r1 = tk.r1;
    for (int i = tid; i < p; i += T) w1[i] = x[fd.first + i];
    if (tid < SLAB) wsl[tid] = 0.0;
    __syncthreads();
    for (int ci = fd.child_begin; ci < fd.child_end; ci++) {
        FrontDesc cd = FD[child_idx[ci]];
        const double *uc = work + cd.woff + cd.p;
        const int32_t *relc = rel + cd.rowptr;
        for (int i = tid; i < cd.m; i += T) {
            const int r = relc[i];
            if (r < p) w1[r] += uc[i];
            else if (r >= r0 && r < r1) wsl[r - r0] += uc[i];
        }
        __syncthreads();
    }
    const int rr = tid & (SLAB - 1), g = tid / SLAB;
    const int r = r0 + rr;
    // rows of inv(L11) P are zero right of their own 32-column block
    int jmax = p;
    if (r1 <= p && !(fd.flags & FD_DENSE_TOP)) jmax = ((r1 - 1) / NB + 1) * NB < p ? ((r1 - 1) / NB + 1) * NB : p; // (k_front leaves a full block)
    double acc = 0.0;
    if (r < r1) acc = strided_dot(E + r, ld, w1, g, jmax, G);
    red[g][rr] = acc;
    __syncthreads();
    if (g == 0 && r < r1) {
        const double tot = group_sum<SLAB, G>(red, rr);
        W[r] = (r < p) ? tot : wsl[rr] + tot;
    }
}

// Backward step of a big front, pivot rows [r0, r1):
//   x1 = E' * [y1; x2],  E'(i, j) = E'[i + j p],  y1 = work[0..p),  x2 = x[rows]
// Dynamic LDS: f doubles.
template <int SLAB, int G>
__global__ void __launch_bounds__(SLAB *G) k_bwd_big(const SolveTask *__restrict__ tasks, const FrontDesc *__restrict__ FD,
                                                 const double *__restrict__ pool, const int32_t *__restrict__ rows,
                                                 const double *__restrict__ work, double *__restrict__ x) {
    HIPMF_DYN_SHARED(double, v);
    __shared__ double red[G][SLAB];
    constexpr int T = SLAB * G;
    const int tid = threadIdx.x;
    SolveTask tk = tasks[blockIdx.x];
    FrontDesc fd = FD[tk.s];
    const int p = fd.p, f = fd.p + fd.m;
    const int64_t ld = fd.ldp;
    const double *Ep = pool + fd.epoff;
    const double *W = work + fd.woff;
    const int32_t *rws = rows + fd.rowptr;
    const int r0 = tk.r0, r1 = tk.r1;
    // columns of inv(U11) left of the slab's first 32-column block are zero
    const int jmin = (r0 / NB) * NB;
    for (int j = jmin + tid; j < f; j += T) v[j] = (j < p) ? W[j] : x[rws[j - p]];
    __syncthreads();
    const int rr = tid & (SLAB - 1), g = tid / SLAB;
    const int i = r0 + rr;
    double acc = 0.0;
    if (i < r1) acc = strided_dot(Ep + i, ld, v, jmin + g, f, G);
    red[g][rr] = acc;
    __syncthreads();
    if (g == 0 && i < r1) x[fd.first + i] = group_sum<SLAB, G>(red, rr);
}

} // namespace hipmf
