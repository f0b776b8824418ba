// kernels_solve_fused.hpp -- dependency-driven sparse triangular solves: ONE launch per direction.
//
// The level-set kernels of kernels_solve.hpp pay one launch (and its chain of dependent loads) per level of the
// assembly tree, 2 x ~25 launches for the 1M-DOF Poisson factor, while the data of the upper levels are tiny.
// Here every front (small: one wavefront, four per workgroup; big: one 256-thread workgroup per row slab) is a
// task in one launch.  Tasks are ordered by level and task = workgroup index, so a task only ever waits for
// workgroups with smaller indices.  The hardware places workgroups in index order (observed, not promised by
// HIP), which makes the waits deadlock-free; correctness does NOT rest on that: every spin is bounded and a
// timeout sends the solve back to the level-set launches.  (A ticket counter would make the order a guarantee,
// but one atomic word hands out only ~90 tickets per microsecond: 500 us for the 41 000 forward tasks of the
// 1M-DOF Poisson factor, more than the whole pass takes.)  A task
//   1. polls the completion counters of the fronts it depends on (children in the forward pass, the parent in the
//      backward pass: the parent has itself waited for its ancestors),
//   2. computes exactly what k_fwd / k_bwd / k_fwd_big / k_bwd_big compute, in the same summation order
//      (the two paths give bit-identical results; tests compare them),
//   3. publishes its results and bumps its front's counter.
// Visibility (MI355X: private L2 per XCD, private L1 per CU): everything exchanged inside the launch (the
// solve workspace `work` and the vector `x`) is written write-through and read around the L1 with agent-scope
// accesses (st_agent / ld_agent); every storing wave drains its stores before the counter is bumped.  The factor
// panels, descriptors and index lists are read-only here and use plain loads.
// Every spin is bounded: on a timeout the error word is set, the waiters give up and the host falls back to
// the level-set path.
#pragma once
#include "kernels_common.hpp"

namespace hipmf {

constexpr int SF_SYNC_HEADER = 16;       // ints in front of the completion counters (reserved)
constexpr unsigned SF_SPIN_LIMIT = 1u << 19;
constexpr int SF_CHUNK = 1024;           // doubles of the big fronts' vectors staged in LDS at a time

struct SfTask {
    int32_t kind;       // 0: group of small fronts, one per wavefront (a, b, c, d; -1 = none)
                        // 4, 5, 6: slab of a big front, kind = log2(rows per slab): a = front, rows [b, c)
    int32_t a, b, c, d;
    int32_t pad;
};

// wait until *cnt >= need (relaxed agent-scope polls); false on timeout or when another waiter timed out
__device__ __forceinline__ bool sf_wait(const int *cnt, int need, int *err) {
    unsigned spins = 0;
    while (flag_load(cnt) < need) {
        poll_nap();
        spins++;
        if (spins > SF_SPIN_LIMIT) {
            flag_store(err, 1);
            return false;
        }
        if ((spins & 1023u) == 0 && flag_load(err) != 0) return false;
    }
    return true;
}

// ---- forward step of one small front by one wavefront; w = 64 doubles of LDS owned by this wave ----
__device__ __forceinline__ void sf_fwd_small(int s, int lane, double *w, const FrontDesc *__restrict__ FD, const double *__restrict__ pool,
                                             const int32_t *__restrict__ lperm, const int32_t *__restrict__ child_idx,
                                             const int32_t *__restrict__ rel, const int32_t *__restrict__ need, int *done, int *err,
                                             double *work, double *x) {
    const FrontDesc fd = FD[s];
    const int p = fd.p, f = fd.p + fd.m;
    const double *F = pool + fd.off;
    double *W = work + fd.woff;
    double *xs = x + fd.first;
    w[lane] = (lane < p) ? ld_agent(xs + lane) : 0.0;
    const int lp = (lane < p) ? lperm[fd.first + lane] : 0;
    // lane c looks after child c: descriptor, completion counter
    const int nch = fd.child_end - fd.child_begin;
    int64_t c_woff = 0, c_rowptr = 0;
    int c_p = 0, c_m = 0;
    for (int c0 = 0; c0 < nch; c0 += 64) { // (more than 64 children: only the last 64 descriptors stay in registers, see below)
        if (c0 + lane < nch) {
            const int ch = child_idx[fd.child_begin + c0 + lane];
            const FrontDesc cd = FD[ch];
            c_woff = cd.woff, c_rowptr = cd.rowptr, c_p = cd.p, c_m = cd.m;
            sf_wait(done + ch, need[ch], err);
        }
    }
    wave_sync();
    for (int ci = 0; ci < nch; ci++) {
        int64_t woff, rowptr;
        int cp, cm;
        if (nch <= 64) {
            woff = __shfl(c_woff, ci), rowptr = __shfl(c_rowptr, ci), cp = __shfl(c_p, ci), cm = __shfl(c_m, ci);
        } else {
            const FrontDesc cd = FD[child_idx[fd.child_begin + ci]];
            woff = cd.woff, rowptr = cd.rowptr, cp = cd.p, cm = cd.m;
        }
        if (lane < cm) w[rel[rowptr + lane]] += ld_agent(work + woff + cp + lane); // cm <= f <= 64
        wave_sync();
    }
    // row interchanges of the pivot block, then y1 = L11^{-1} (P w1) column by column, u = w2 - L21 y1;
    // the lane's row of [L11; L21] comes 16 columns at a time (all 16 loads in flight)
    double v = (lane < p) ? w[lp] : ((lane < f) ? w[lane] : 0.0);
    for (int j0 = 0; j0 < p; j0 += 16) {
        double a[16];
#pragma unroll
        for (int q = 0; q < 16; q++) a[q] = (lane < f && j0 + q < p) ? F[lane + (int64_t)(j0 + q) * f] : 0.0;
#pragma unroll
        for (int q = 0; q < 16; q++) {
            const int j = j0 + q;
            const double vj = __shfl(v, j & 63);
            if (lane > j) v -= a[q] * vj; // a[q] == 0 for j >= p and for lanes >= f
        }
    }
    if (lane < p) st_agent(xs + lane, v);
    else if (lane < f) st_agent(W + lane, v);
    drain_stores();
    if (lane == 0) flag_add(done + s, 1);
}

// ---- backward step of one small front by one wavefront; xg = 64 doubles of LDS owned by this wave ----
__device__ __forceinline__ void sf_bwd_small(int s, int lane, double *xg, const FrontDesc *__restrict__ FD, const double *__restrict__ pool,
                                             const int32_t *__restrict__ rows, const int32_t *__restrict__ need, int *done, int *err,
                                             double *x) {
    const FrontDesc fd = FD[s];
    const int p = fd.p, m = fd.m, f = fd.p + fd.m;
    const double *F = pool + fd.off;
    double *xs = x + fd.first;
    const int32_t *rws = rows + fd.rowptr;
    const int myrow = (lane < m) ? rws[lane] : 0;
    const double y1 = (lane < p) ? ld_agent(xs + lane) : 0.0; // from the forward launch
    if (fd.parent >= 0 && lane == 0) sf_wait(done + fd.parent, need[fd.parent], err);
    wave_sync();
    if (lane < m) xg[lane] = ld_agent(x + myrow);
    wave_sync();
    const int sh = p <= 16 ? 4 : (p <= 32 ? 5 : 6);
    const int i = lane & ((1 << sh) - 1), jq = lane >> sh, ng = 64 >> sh;
    double acc = 0.0;
    if (i < p) {
        const double *Ui = F + i + (int64_t)p * f;
        int j = jq;
        for (; j + 3 * ng < m; j += 4 * ng) {
            const double e0 = Ui[(int64_t)j * f], e1 = Ui[(int64_t)(j + ng) * f], e2 = Ui[(int64_t)(j + 2 * ng) * f], e3 = Ui[(int64_t)(j + 3 * ng) * f];
            acc += e0 * xg[j];
            acc += e1 * xg[j + ng];
            acc += e2 * xg[j + 2 * ng];
            acc += e3 * xg[j + 3 * ng];
        }
        for (; j < m; j += ng) acc += Ui[(int64_t)j * f] * xg[j];
    }
    for (int off = 1 << sh; off < 64; off <<= 1) acc += __shfl_xor(acc, off);
    double v = (lane < p) ? y1 - acc : 0.0;
    // x1 = U11^{-1} t, columns from right to left, the lane's row of U11 16 columns at a time
    for (int jhi = p; jhi > 0; jhi -= 16) {
        double a[16];
#pragma unroll
        for (int q = 0; q < 16; q++) {
            const int j = jhi - 1 - q;
            a[q] = (lane < p && j >= 0) ? F[lane + (int64_t)j * f] : 1.0;
        }
#pragma unroll
        for (int q = 0; q < 16; q++) {
            const int j = jhi - 1 - q;
            if (j >= 0) { // wave-uniform
                if (lane == j) v /= a[q];
                const double vj = __shfl(v, j);
                if (lane < j) v -= a[q] * vj;
            }
        }
    }
    if (lane < p) st_agent(xs + lane, v);
    drain_stores();
    if (lane == 0) flag_add(done + s, 1);
}

// Strided dot product against an LDS vector chunk: acc += sum_j col[j * ld] * w[j - c0], j = j0, j0 + step, ... < j1.
__device__ __forceinline__ void sf_dot(double &acc0, double &acc1, const double *__restrict__ col, int64_t ld, const double *w, int c0, int j0,
                                       int j1, int step) {
    int j = j0;
    for (; j + 7 * step < j1; j += 8 * step) {
        double e[8];
#pragma unroll
        for (int u = 0; u < 8; u++) e[u] = col[(int64_t)(j + u * step) * ld];
#pragma unroll
        for (int u = 0; u < 8; u += 2) {
            acc0 += e[u] * w[j + u * step - c0];
            acc1 += e[u + 1] * w[j + (u + 1) * step - c0];
        }
    }
    for (; j < j1; j += step) acc0 += col[(int64_t)j * ld] * w[j - c0];
}

// Forward pass, one launch.  sync[SF_SYNC_HEADER + s] = completed tasks of front s (zeroed before
// every pass); *err is sticky: set when a wait timed out.
__global__ void __launch_bounds__(256) k_fwd_fused(const SfTask *__restrict__ tasks, const FrontDesc *__restrict__ FD,
                                                   const double *__restrict__ pool, const int32_t *__restrict__ lperm,
                                                   const int32_t *__restrict__ child_idx, const int32_t *__restrict__ rel,
                                                   const int32_t *__restrict__ need, int *sync, int *err, double *work, double *x) {
    __shared__ double wv[4][64];
    __shared__ double wc[SF_CHUNK];
    __shared__ double wsl[SOLVE_SLAB];
    __shared__ double red[256];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    int *done = sync + SF_SYNC_HEADER;
    const SfTask t = tasks[blockIdx.x];
    if (t.kind == 0) {
        const int s = wave == 0 ? t.a : (wave == 1 ? t.b : (wave == 2 ? t.c : t.d));
        if (s >= 0) sf_fwd_small(s, lane, wv[wave], FD, pool, lperm, child_idx, rel, need, done, err, work, x);
        return;
    }
    // ---- slab [r0, r1) of the big front t.a:  [y1; -delta] = E w1,  work[r] = y1[r] (r < p) or w2[r] + (E w1)[r] ----
    const FrontDesc fd = FD[t.a];
    const int p = fd.p, f = fd.p + fd.m;
    const int64_t ld = fd.ld;
    const double *E = pool + fd.off + (int64_t)f * ld;
    double *W = work + fd.woff;
    const int r0 = t.b, r1 = t.c, sh = t.kind;
    const int rr = tid & ((1 << sh) - 1), g = tid >> sh, G = 256 >> sh;
    const int r = r0 + rr;
    if (tid < SOLVE_SLAB) wsl[tid] = 0.0;
    // wave 0 waits for the children, one child per lane
    const int nch = fd.child_end - fd.child_begin;
    if (wave == 0)
        for (int c0 = 0; c0 < nch; c0 += 64)
            if (c0 + lane < nch) {
                const int ch = child_idx[fd.child_begin + c0 + lane];
                sf_wait(done + ch, need[ch], err);
            }
    __syncthreads();
    // rows of inv(L11) P are zero right of their own 32-column block
    int jmax = p;
    if (r1 <= p) jmax = ((r1 - 1) / NB + 1) * NB < p ? ((r1 - 1) / NB + 1) * NB : p;
    double acc0 = 0.0, acc1 = 0.0;
    for (int c0 = 0; c0 < jmax; c0 += SF_CHUNK) {
        const int c1 = c0 + SF_CHUNK < jmax ? c0 + SF_CHUNK : jmax;
        // w1[c0, c1) = b1 + the children's updates to these pivot rows (children in ascending order)
        for (int i = c0 + tid; i < c1; i += 256) wc[i - c0] = ld_agent(x + fd.first + i);
        __syncthreads();
        for (int ci = fd.child_begin; ci < fd.child_end; ci++) {
            const FrontDesc cd = FD[child_idx[ci]];
            const double *uc = work + cd.woff + cd.p;
            const int32_t *relc = rel + cd.rowptr;
            for (int i = tid; i < cd.m; i += 256) {
                const int q = relc[i];
                if (q >= c0 && q < c1) wc[q - c0] += ld_agent(uc + i);
                else if (c0 == 0 && q >= p && q >= r0 && q < r1) wsl[q - r0] += ld_agent(uc + i);
            }
            __syncthreads();
        }
        // the group's columns of this chunk: g, g + G, ... continue across chunks (SF_CHUNK is a multiple of every G)
        if (r < r1) sf_dot(acc0, acc1, E + r, ld, wc, c0, c0 + g, c1, G);
        __syncthreads();
    }
    if (jmax <= 0) { // p == 0 cannot happen for a front; keeps wsl complete if it ever does
        __syncthreads();
    }
    red[g * (1 << sh) + rr] = acc0 + acc1;
    __syncthreads();
    if (g == 0 && r < r1) {
        // pairwise sum over the G column groups in a fixed order
        double tsum[16];
#pragma unroll
        for (int q = 0; q < 16; q++) tsum[q] = (q < G) ? red[q * (1 << sh) + rr] : 0.0;
#pragma unroll
        for (int wdt = 1; wdt < 16; wdt <<= 1)
#pragma unroll
            for (int q = 0; q + wdt < 16; q += 2 * wdt) tsum[q] += tsum[q + wdt];
        st_agent(W + r, (r < p) ? tsum[0] : wsl[rr] + tsum[0]);
    }
    drain_stores();
    __syncthreads();
    if (tid == 0) flag_add(done + t.a, 1);
}

// Backward pass, one launch (tasks ordered root first).
__global__ void __launch_bounds__(256) k_bwd_fused(const SfTask *__restrict__ tasks, const FrontDesc *__restrict__ FD,
                                                   const double *__restrict__ pool, const int32_t *__restrict__ rows,
                                                   const int32_t *__restrict__ need, int *sync, int *err, const double *work, double *x) {
    __shared__ double wv[4][64];
    __shared__ double wc[SF_CHUNK];
    __shared__ double red[256];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    int *done = sync + SF_SYNC_HEADER;
    const SfTask t = tasks[blockIdx.x];
    if (t.kind == 0) {
        const int s = wave == 0 ? t.a : (wave == 1 ? t.b : (wave == 2 ? t.c : t.d));
        if (s >= 0) sf_bwd_small(s, lane, wv[wave], FD, pool, rows, need, done, err, x);
        return;
    }
    // ---- pivot rows [r0, r1) of the big front t.a:  x1 = E' [y1; x2] ----
    const FrontDesc fd = FD[t.a];
    const int p = fd.p, f = fd.p + fd.m;
    const int64_t ld = fd.ld;
    const double *Ep = pool + fd.off + f;
    const double *W = work + fd.woff; // y1: written by the forward launch
    const int32_t *rws = rows + fd.rowptr;
    const int r0 = t.b, r1 = t.c, sh = t.kind;
    const int rr = tid & ((1 << sh) - 1), g = tid >> sh, G = 256 >> sh;
    const int i = r0 + rr;
    if (fd.parent >= 0 && tid == 0) sf_wait(done + fd.parent, need[fd.parent], err);
    __syncthreads();
    // columns of inv(U11) left of the slab's first 32-column block are zero
    const int jmin = (r0 / NB) * NB;
    double acc0 = 0.0, acc1 = 0.0;
    for (int c0 = jmin; c0 < f; c0 += SF_CHUNK) {
        const int c1 = c0 + SF_CHUNK < f ? c0 + SF_CHUNK : f;
        for (int j = c0 + tid; j < c1; j += 256) wc[j - c0] = (j < p) ? W[j] : ld_agent(x + rws[j - p]);
        __syncthreads();
        if (i < r1) sf_dot(acc0, acc1, Ep + i, ld, wc, c0, c0 + g, c1, G);
        __syncthreads();
    }
    red[g * (1 << sh) + rr] = acc0 + acc1;
    __syncthreads();
    if (g == 0 && i < r1) {
        double tsum[16];
#pragma unroll
        for (int q = 0; q < 16; q++) tsum[q] = (q < G) ? red[q * (1 << sh) + rr] : 0.0;
#pragma unroll
        for (int wdt = 1; wdt < 16; wdt <<= 1)
#pragma unroll
            for (int q = 0; q + wdt < 16; q += 2 * wdt) tsum[q] += tsum[q + wdt];
        st_agent(x + fd.first + i, tsum[0]);
    }
    drain_stores();
    __syncthreads();
    if (tid == 0) flag_add(done + t.a, 1);
}

} // namespace hipmf
