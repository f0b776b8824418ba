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
constexpr unsigned long long SF_WAIT_TICKS = 10ull * 100000000ull; // a wait gives up after 10 s of the constant 100 MHz device clock
// (a poll COUNT was the limit until round 3: 2^19 polls are ~0.2 s, which the upper levels of a 200^3 factor exceed when two blocks of
//  16 right-hand sides share the GPU -- the timeout sent that solve to the one-column fallback, 26 s instead of 3 s)
constexpr int SF_CHUNK = 1024;           // doubles of a big front's vector staged in LDS at a time, per right-hand side (the children are
                                         // re-scanned for every chunk; 1024 doubles per right-hand side up to K = 4, 512 at K = 8: 32 KB of LDS)
constexpr int SF_SYMC = 4;               // columns of E per wavefront in the backward slabs of the symmetric (L D L^T) fronts: slabs of 16 pivots
constexpr int SF_KMAX = 16;              // right-hand sides solved together by the widest blocked instance (the factor is read once per block):
                                         // a full 16-column MFMA tile; narrower blocks (2..8 columns) use the SF_KMID instance
constexpr int SF_KMID = 8;
constexpr int SF_ASM_ROWS = 256;         // rows per ASSEMBLE task (kind 1): the smallest chunk of any instance

struct SfTask {
    int32_t kind;       // 0: group of small fronts, one per wavefront (a, b, c, d; -1 = none)
                        // 3..7: slab of a big front, kind = log2(rows per slab): a = front, rows [b, c); forward pass: d = number of
                        //       ASSEMBLE tasks of the front (0: the slab gathers the children's vectors itself)
                        // 1:    forward pass only, fronts of thousands of rows: assemble rows [b, c) of front a's vector ONCE
                        //       (right-hand side + the children's updates, children in ascending order) for all of its slabs
                        // 2:    forward pass, single right-hand side: group of WAVE FRONTS (a, b, c, d; -1 = none) -- big fronts (stored as
                        //       E) of at most SF_WF_ROWS rows and SF_WF_PIV pivots, one per wavefront (sf_fwd_wave)
    int32_t a, b, c, d;
    int32_t part;       // backward slabs of the blocked instances, levels of few tasks: 0 = the slab's whole dot product; otherwise
                        // q | Q << 8: this task forms part q of Q of it (a contiguous range of positions; the Q parts are consecutive tasks),
                        // d = the group's first 256-double unit in the scratch of partial sums (see k_bwd_fused)
};

// ---- block groups (round 6): several blocks of K right-hand sides in ONE launch of a blocked (K > 1) instance ----
// A pass over the upper levels of a 2D factor is a chain of hand-offs (17 levels per direction at ~18 us per level with 16 columns at the
// 1M-DOF Poisson factor: profiles/r05_solve_trace_c2_16col.txt) during which most of the device idles.  A launch therefore carries `ngrp`
// independent blocks ("groups") of K columns each: group g owns the columns [g K, g K + K) of x and of the workspace block and its own
// set of completion counters; the grid holds every task ngrp times and the chains of the groups overlap.  Placement: workgroup b runs on
// XCD b mod 8 (observed, MI355X_MICROARCH.md), so the ngrp copies of the tasks t .. t + 7 sit in 8 ngrp consecutive workgroups, copy g of
// task t at (t / 8) 8 ngrp + 8 g + t mod 8: the copies of one task land on ONE XCD right after each other and share the task's piece of
// the factor through that XCD's L2.  A task still only waits for tasks of its own group with smaller task numbers, which sit at smaller
// workgroup indices: the order argument at the head of this file is untouched.  Per column the arithmetic is that of one group per launch.
constexpr int SF_GMAX = 4;               // groups a launch carries at most
struct SfGroups {
    int32_t ngrp;          // groups in this launch (1: the grid is the task list itself)
    int32_t ntask;         // tasks of the list (the grid is padded to whole sets of 8 ngrp workgroups)
    int32_t nk_total;      // live columns over all groups (group g carries min(K, nk_total - g K))
    uint32_t mask;         // groups with live work (a refinement step whose active columns all sit in other groups skips a group)
    int64_t sync_stride;   // ints between the completion counters of consecutive groups
    int64_t split_stride;  // units of the split-dot-product scratch between consecutive groups
};
// resolves blockIdx.x into (task, group); false: nothing to do for this workgroup
template <int K>
__device__ __forceinline__ bool sf_group_of(const SfGroups &G, int &bid, int &grp, int &nk) {
    bid = blockIdx.x, grp = 0;
    if (K == 1 || G.ngrp <= 1) return true;
    const int per = 8 * G.ngrp;
    grp = (bid % per) >> 3;
    bid = (bid / per) * 8 + (bid & 7);
    if (bid >= G.ntask || !((G.mask >> grp) & 1u)) return false;
    nk = G.nk_total - grp * K < K ? G.nk_total - grp * K : K;
    return nk > 0;
}

// wait until *cnt >= need (relaxed agent-scope polls); false on timeout or when another waiter timed out
__device__ __forceinline__ bool sf_wait(const int *cnt, int need, int *err) {
    unsigned spins = 0;
    unsigned long long t0 = 0;
    while (flag_load(cnt) < need) {
        poll_nap();
        spins++;
        if ((spins & 1023u) == 0) {
            if (flag_load(err) != 0) return false;
            const unsigned long long now = dev_clock();
            if (t0 == 0) t0 = now;
            else if (now - t0 > SF_WAIT_TICKS) {
                flag_store(err, 1);
                return false;
            }
        }
    }
    return true;
}

// Fronts of the top levels are cut into up to ~170 slabs, and up to ~170 slabs of the parent (forward) / of the children (backward)
// wait for them: that many pollers on ONE counter word serialise with each other and with the arrivals (measured: 1.3 - 7.6 us from
// the last publish of a level to the last wake-up of the next).  Such a front has SF_REP replicas of a "complete" word, one per 64-byte
// line, written by whoever makes the counter reach its target; a waiting slab polls the replica its workgroup index selects.
constexpr int SF_REP = 16;
__device__ __forceinline__ void sf_wait_front(int s, const int32_t *__restrict__ need, const int *done, const int32_t *__restrict__ rep_idx,
                                              const int *rep, int *err) {
    const int ri = rep_idx ? rep_idx[s] : -1;
    if (ri >= 0) sf_wait(rep + ((size_t)ri * SF_REP + (blockIdx.x % SF_REP)) * 16, 1, err);
    else sf_wait(done + s, need[s], err);
}
__device__ __forceinline__ void sf_publish_front(int s, const int32_t *__restrict__ need, int *done, const int32_t *__restrict__ rep_idx, int *rep) {
    const int old = flag_add(done + s, 1);
    if (rep_idx) {
        const int ri = rep_idx[s];
        if (ri >= 0 && old + 1 == need[s])
            for (int k = 0; k < SF_REP; k++) flag_store(rep + ((size_t)ri * SF_REP + k) * 16, 1);
    }
}

// ---- data-tagged hand-offs (TAG instances: single right-hand side above the wave-subtrees) ----
// A hand-off through a completion counter costs the producer a drain of its stores and an atomic, and the consumer a poll of the counter
// FOLLOWED by a dependent round trip for the data (measured per level of the 1M-DOF Poisson factor: ~0.5-1 us publish, 1.5-2.5 us until
// the waiter sees the counter, 2-3 us to gather the children's vectors -- profiles/r04_solve_trace.txt).  In the TAG instances the data
// ARE the flag (MI355X_MICROARCH.md, handoff-1to1 against handoff-flag): every 8-byte word a task of the launch hands to another one
// (forward: the update part of a front's vector in `work`; backward: the solved pivot entries, in a shadow copy `xt` of x) is set to
// SF_TAG_BITS by a memset on the stream before the launch; producers store each word once (one aligned 8-byte write-through store, never
// torn); consumers load the words they need and re-load the ones that still hold the tag.  No drain, no counter, no second round trip.
// SF_TAG_BITS is a NaN no arithmetic produces (all ones; generated NaNs are the canonical quiet NaN, and k_perm_in canonicalises the
// NaNs of a right-hand side), so a value never looks like "not yet written"; every re-load loop is bounded like sf_wait.
constexpr long long SF_TAG_BITS = -1LL;
__device__ __forceinline__ bool sf_is_tag(double v) { return __double_as_longlong(v) == SF_TAG_BITS; }
__device__ __forceinline__ double sf_tag_wait(const double *p, double v, int *err) {
    unsigned spins = 0;
    unsigned long long t0 = 0;
    while (sf_is_tag(v)) {
        poll_nap();
        v = ld_agent(p);
        spins++;
        if ((spins & 1023u) == 0) {
            if (flag_load(err) != 0) return 0.0;
            const unsigned long long now = dev_clock();
            if (t0 == 0) t0 = now;
            else if (now - t0 > SF_WAIT_TICKS) {
                flag_store(err, 1);
                return 0.0;
            }
        }
    }
    return v;
}
__device__ __forceinline__ double sf_tag_load(const double *p, int *err) { return sf_tag_wait(p, ld_agent(p), err); }

// One entry per column of a block of right-hand sides, column c < nk at base + c * cstr + off: K agent-scope loads issued back to back.
// The compiler never speculates an agent-scope load: written as `cond ? ld_agent(p) : 0` (or under `if (c < nk)`) every load sits in a
// branch of its own and waits for its own round trip -- sixteen columns cost sixteen round trips (measured on the blocked small-front
// step: 9 us from entry to the last of its sixteen loads, 8.5 us per child gathered; tools/ab stamps, round 3).  Callers clamp `off` to
// an address that is always valid and drop what they do not need; the columns >= nk re-read column 0.
template <int K> __device__ __forceinline__ void ld_cols(double (&v)[K], const double *base, int64_t cstr, int64_t off, int nk) {
#pragma unroll
    for (int c = 0; c < K; c++) v[c] = ld_agent(base + (int64_t)(c < nk ? c : 0) * cstr + off);
}

// PLAIN instances (round 6, blocked solves: the all-small band at the bottom of the tree as ONE LAUNCH PER LEVEL): everything a front reads was
// written by an EARLIER launch of the stream, so ordinary loads and stores do -- which the compiler batches (sixteen columns in one
// round trip without the clamping tricks), which need no drain before a counter, and which are not one fabric transaction per lane and
// column like the write-through stores (measured on the blocked small-front step: stores + drain 7 - 9 us of a 16 - 23 us front,
// profiles/r03_rejected_experiments.txt).
template <int K, bool PLAIN> __device__ __forceinline__ void ld_cols_p(double (&v)[K], const double *base, int64_t cstr, int64_t off, int nk) {
    if constexpr (PLAIN) {
#pragma unroll
        for (int c = 0; c < K; c++) v[c] = base[(int64_t)(c < nk ? c : 0) * cstr + off];
    } else
        ld_cols<K>(v, base, cstr, off, nk);
}
template <bool PLAIN> __device__ __forceinline__ void st_p(double *p, double v) {
    if constexpr (PLAIN) *p = v;
    else st_agent(p, v);
}

// All kernels are templates on K = the number of right-hand sides a launch carries (1: the instances the
// benchmark path uses; SF_KMAX: the many-RHS instances, which read every factor entry ONCE for K columns -- the
// solves are HBM-bound, so K columns cost little more than one).  Column c of x lives at x + c * xstr, its solve
// workspace at work + c * wstr; nk <= K columns are live.  Small fronts: per column the arithmetic and its order are those of K = 1;
// big-front slabs of the blocked instances run on MFMA tiles (sf_mma_chunk): another summation order, equal to rounding.

// ---- forward step of one small front by one wavefront; w = K x 64 doubles of LDS owned by this wave ----
template <int K, bool TAG = false, bool PLAIN = false>
__device__ __forceinline__ void sf_fwd_small(int s, int lane, double (*w)[64], const FrontDesc *__restrict__ FD, const double *__restrict__ pool,
                                             const int32_t *__restrict__ lperm, const int32_t *__restrict__ child_idx,
                                             const int32_t *__restrict__ rel, const int32_t *__restrict__ need, int *done, int *err,
                                             double *work, double *x, int nk, int64_t xstr, int64_t wstr) {
    const FrontDesc fd = FD[s];
    const int p = fd.p, f = fd.p + fd.m;
    const double *F = pool + fd.off;
    double *W = work + fd.woff;
    double *xs = x + fd.first;
    {
        double xv[K];
        ld_cols_p<K, PLAIN>(xv, xs, xstr, lane < p ? lane : 0, nk);
#pragma unroll
        for (int c = 0; c < K; c++)
            if (c < nk) w[c][lane] = (lane < p) ? xv[c] : 0.0;
    }
    const int lp = (lane < p) ? lperm[fd.first + lane] : 0;
    // the lane's row of [L11; L21], first 8 columns: on their way before the waits (the panel is read-only)
    constexpr int CH = 8;
    double a[CH];
#pragma unroll
    for (int q = 0; q < CH; q++) a[q] = (lane < f && q < p) ? F[lane + (int64_t)q * f] : 0.0;
    // lane c looks after child c: descriptor, completion counter
    const int nch = fd.child_end - fd.child_begin;
    int64_t c_woff = 0, c_rowptr = 0;
    int c_p = 0, c_m = 0;
    for (int c0 = 0; c0 < nch; c0 += 64) { // children in batches of 64: descriptor and completion counter, one child per lane
        c_m = 0;
        if (c0 + lane < nch) {
            const int ch = child_idx[fd.child_begin + c0 + lane];
            const FrontDesc cd = FD[ch];
            c_woff = cd.woff, c_rowptr = cd.rowptr, c_p = cd.p, c_m = cd.m;
            if constexpr (!TAG && !PLAIN) sf_wait(done + ch, need[ch], err); // (TAG: the gathered words themselves say when they are there)
        }
        wave_sync();
        const int nbatch = nch - c0 < 64 ? nch - c0 : 64;
        if (nch > 64 && __ballot(c_m > 1) == 0ull) {
            // hub front: every child of this batch brings at most one entry; one lane per child fetches it, the owning row adds
            // them in child order (fixed order: reproducible sums)
            const int myr = (lane < nbatch && c_m == 1) ? rel[c_rowptr] : -1;
            double myv[K];
            ld_cols_p<K, PLAIN>(myv, work, wstr, myr >= 0 ? c_woff + c_p : 0, nk);
            if constexpr (TAG) {
                if (myr >= 0) myv[0] = sf_tag_wait(work + c_woff + c_p, myv[0], err);
            }
#pragma unroll
            for (int c = 0; c < K; c++) myv[c] = (c < nk && myr >= 0) ? myv[c] : 0.0;
            double add[K]; // starts from the row's current value: the same association order as adding child after child
#pragma unroll
            for (int c = 0; c < K; c++) add[c] = (c < nk) ? w[c][lane] : 0.0;
            for (int l = 0; l < nbatch; l++) {
                const int r = wave_bcast_i32(myr, l);
#pragma unroll
                for (int c = 0; c < K; c++) {
                    const double v = wave_bcast(myv[c], l);
                    if (lane == r) add[c] += v;
                }
            }
#pragma unroll
            for (int c = 0; c < K; c++)
                if (c < nk) w[c][lane] = add[c];
            wave_sync();
            continue;
        }
        for (int cl = 0; cl < nbatch; cl++) {
            // (cl is wave-uniform: v_readlane, not a trip through the LDS crossbar)
            const int64_t woff = wave_bcast_i64(c_woff, cl), rowptr = wave_bcast_i64(c_rowptr, cl);
            const int cp = wave_bcast_i32(c_p, cl), cm = wave_bcast_i32(c_m, cl);
            if constexpr (K == 1) {
                if constexpr (TAG) {
                    if (lane < cm) w[0][rel[rowptr + lane]] += sf_tag_load(work + woff + cp + lane, err);
                } else if (lane < cm) w[0][rel[rowptr + lane]] += ld_agent(work + woff + cp + lane); // cm <= f <= 64
            } else if (cm > 0) { // (wave-uniform)
                const int gl = lane < cm ? lane : 0;
                const int r = rel[rowptr + gl];
                double gv[K];
                ld_cols_p<K, PLAIN>(gv, work, wstr, woff + cp + gl, nk);
                if (lane < cm) {
#pragma unroll
                    for (int c = 0; c < K; c++)
                        if (c < nk) w[c][r] += gv[c];
                }
            }
            wave_sync();
        }
    }
    // row interchanges of the pivot block, then y1 = L11^{-1} (P w1) column by column, u = w2 - L21 y1;
    // (a front without children reaches this point without a wave-level barrier since w was written: the interchange reads
    //  ANOTHER lane's entry.  The hardware runs a wave's LDS operations in order; the barrier costs nothing there and makes the
    //  dependence explicit -- tools/hipemu, which runs the lanes one after the other, needs it)
    wave_sync();
    double v[K];
#pragma unroll
    for (int c = 0; c < K; c++) v[c] = (c < nk) ? ((lane < p) ? w[c][lp] : ((lane < f) ? w[c][lane] : 0.0)) : 0.0;
    for (int j0 = 0; j0 < p; j0 += CH) {
        double an[CH]; // next chunk: its loads fly while this chunk's substitution steps run
#pragma unroll
        for (int q = 0; q < CH; q++) an[q] = (lane < f && j0 + CH + q < p) ? F[lane + (int64_t)(j0 + CH + q) * f] : 0.0;
#pragma unroll
        for (int q = 0; q < CH; q++) {
            const int j = j0 + q;
#pragma unroll
            for (int c = 0; c < K; c++) {
                const double vj = wave_bcast(v[c], j & 63); // (j is wave-uniform: v_readlane)
                if (lane > j) v[c] -= a[q] * vj; // a[q] == 0 for j >= p and for lanes >= f
            }
        }
#pragma unroll
        for (int q = 0; q < CH; q++) a[q] = an[q];
    }
#pragma unroll
    for (int c = 0; c < K; c++)
        if (c < nk) {
            if (lane < p) st_p<PLAIN>(xs + c * xstr + lane, v[c]);
            else if (lane < f) st_p<PLAIN>(W + c * wstr + lane, v[c]);
        }
    if constexpr (!TAG && !PLAIN) {
        drain_stores();
        if (lane == 0) flag_add(done + s, 1);
    }
}

// ---- forward step of one WAVE FRONT by one wavefront (round 5) ----
// The levels right above the wave-subtrees hold thousands of big fronts that are big in name only: f = 65 ... 128 rows, ~16 pivots, two to
// six children (1000 x 1000 Poisson: 2 700 of them on levels 3 - 5).  As 256-thread slab tasks they fill the device's workgroup slots
// (four workgroups per compute unit) several times over, each task a chain of dependent round trips for a few hundred multiply-adds:
// those levels were bound by slots x task latency, 11 - 15 us per level (profiles/r05_solve_trace_step1_tagged.txt).  Here such a front is the
// work of ONE wavefront, four fronts per workgroup: the lane owns rows lane and lane + 64; the first SF_WF_PRE columns of its rows of E
// are requested before anything is waited for (registers), the children's update vectors are gathered four children at a time (all
// loads of a batch in flight together) into the wave's LDS copy of the front's vector, in child order; then
//   [y1; -delta] = E w1  (columns one after the other, even ones into one accumulator, odd ones into another),  u = w2 + (E w1)_2.
// Same result as the slab tasks to rounding (another summation order); HIPMF_SOLVE_SLAB64=1 / HIPMF_WAVE_FRONTS=0 keep the slab tasks.
constexpr int SF_WF_ROWS = 128, SF_WF_PIV = 32, SF_WF_PRE = 16;
template <bool TAG>
__device__ __forceinline__ void sf_fwd_wave(int s, int lane, double *w, const FrontDesc *__restrict__ FD, const double *__restrict__ pool,
                                            const int32_t *__restrict__ child_idx, const int32_t *__restrict__ rel,
                                            const int32_t *__restrict__ need, int *done, int *err, double *work, const double *x) {
    const FrontDesc fd = FD[s];
    const int p = fd.p, f = fd.p + fd.m;
    const int64_t ld = fd.ld;
    const double *E = pool + fd.eoff;
    double *W = work + fd.woff;
    const int r0 = lane, r1 = lane + 64;
    const bool h1 = r1 < f; // (r0 < f always: f > 64)
    // the front's vector: b1 on the pivot rows, zeros below (x is set before the launch: the children write `work`)
    const double b1 = x[fd.first + (r0 < p ? r0 : 0)];
    // the lane's rows of E, first SF_WF_PRE columns (clamped addresses: unconditional loads, all in flight at once)
    double e0[SF_WF_PRE], e1[SF_WF_PRE];
    {
        const double *E0 = E + r0, *E1 = E + (h1 ? r1 : r0);
#pragma unroll
        for (int c = 0; c < SF_WF_PRE; c++) {
            const int64_t cc = (int64_t)(c < p ? c : p - 1) * ld;
            e0[c] = E0[cc], e1[c] = E1[cc];
        }
    }
    w[r0] = (r0 < p) ? b1 : 0.0;
    w[r1] = 0.0;
    // lane c looks after child c (at most 64 children: the planner sends other fronts to the slab tasks)
    const int nch = fd.child_end - fd.child_begin;
    int64_t c_src = 0, c_rel = 0;
    int c_m = 0;
    if (lane < nch) {
        const int ch = child_idx[fd.child_begin + lane];
        const FrontDesc cd = FD[ch];
        c_src = cd.woff + cd.p, c_rel = cd.rowptr, c_m = cd.m;
        if constexpr (!TAG) sf_wait(done + ch, need[ch], err);
    }
    wave_sync();
    for (int cb = 0; cb < nch; cb += 4) {
        int q[4][2];
        double v[4][2];
#pragma unroll
        for (int k = 0; k < 4; k++) {
            const int cl = cb + k < nch ? cb + k : cb; // (wave-uniform; a batch's unused places re-read its first child and drop the values)
            const int64_t src = wave_bcast_i64(c_src, cl), rl = wave_bcast_i64(c_rel, cl);
            const int m = cb + k < nch ? wave_bcast_i32(c_m, cl) : 0;
#pragma unroll
            for (int e = 0; e < 2; e++) {
                const int i = lane + 64 * e, ic = i < m ? i : 0;
                const int qq = rel[rl + ic];
                v[k][e] = ld_agent(work + src + ic); // (`work` and `rel` are padded: index 0 of a child without update rows is inside the allocation)
                q[k][e] = i < m ? qq : -1;
            }
        }
#pragma unroll
        for (int k = 0; k < 4; k++) {
            if (cb + k < nch) { // (wave-uniform) children in ascending order: the order fixes the floating-point sums
#pragma unroll
                for (int e = 0; e < 2; e++)
                    if (q[k][e] >= 0) {
                        if constexpr (TAG) {
                            if (sf_is_tag(v[k][e])) v[k][e] = sf_tag_wait(work + wave_bcast_i64(c_src, cb + k) + lane + 64 * e, v[k][e], err);
                        }
                        w[q[k][e]] += v[k][e];
                    }
                wave_sync();
            }
        }
    }
    wave_sync();
    // E w1: even columns into a, odd ones into b
    double a0 = 0.0, b0 = 0.0, a1 = 0.0, b1s = 0.0;
#pragma unroll
    for (int c = 0; c < SF_WF_PRE; c += 2) {
        if (c < p) { // (wave-uniform)
            const double wj = w[c];
            a0 += e0[c] * wj, a1 += e1[c] * wj;
        }
        if (c + 1 < p) {
            const double wj = w[c + 1];
            b0 += e0[c + 1] * wj, b1s += e1[c + 1] * wj;
        }
    }
    if (p > SF_WF_PRE) { // (wave-uniform) the columns past the prefetched ones
        const double *E0 = E + r0, *E1 = E + (h1 ? r1 : r0);
#pragma unroll
        for (int c = 0; c < SF_WF_PIV - SF_WF_PRE; c++) {
            const int64_t cc = (int64_t)(SF_WF_PRE + c < p ? SF_WF_PRE + c : p - 1) * ld;
            e0[c] = E0[cc], e1[c] = E1[cc];
        }
#pragma unroll
        for (int c = 0; c < SF_WF_PIV - SF_WF_PRE; c += 2) {
            if (SF_WF_PRE + c < p) {
                const double wj = w[SF_WF_PRE + c];
                a0 += e0[c] * wj, a1 += e1[c] * wj;
            }
            if (SF_WF_PRE + c + 1 < p) {
                const double wj = w[SF_WF_PRE + c + 1];
                b0 += e0[c + 1] * wj, b1s += e1[c + 1] * wj;
            }
        }
    }
    const double t0 = a0 + b0, t1 = a1 + b1s;
    st_agent(W + r0, (r0 < p) ? t0 : w[r0] + t0);
    if (h1) st_agent(W + r1, w[r1] + t1);
    if constexpr (!TAG) {
        drain_stores();
        if (lane == 0) flag_add(done + s, 1);
    }
}

// ---- backward step of one small front by one wavefront; xg = K x 64 doubles of LDS owned by this wave ----
template <int K, bool TAG = false, bool PLAIN = false>
__device__ __forceinline__ void sf_bwd_small(int s, int lane, double (*xg)[64], const FrontDesc *__restrict__ FD, const double *__restrict__ pool,
                                             const int32_t *__restrict__ rows, const int32_t *__restrict__ need, int *done, int *err,
                                             double *x, int nk, int64_t xstr, double *xt = nullptr) {
    const FrontDesc fd = FD[s];
    const int p = fd.p, m = fd.m, f = fd.p + fd.m;
    const double *F = pool + fd.off;
    double *xs = x + fd.first;
    const int32_t *rws = rows + fd.rowptr;
    const int myrow = (lane < m) ? rws[lane] : 0;
    double y1[K]; // from the forward launch
    ld_cols_p<K, PLAIN>(y1, xs, xstr, lane < p ? lane : 0, nk);
#pragma unroll
    for (int c = 0; c < K; c++) y1[c] = (c < nk && lane < p) ? y1[c] : 0.0;
    const int sh = p <= 16 ? 4 : (p <= 32 ? 5 : 6);
    const int i = lane & ((1 << sh) - 1), jq = lane >> sh, ng = 64 >> sh;
    // read-only factor data on their way before the wait: the first 8 of this lane's U12 entries and the
    // rightmost 16 columns of its row of U11
    // (the rows of U come from the packed p x f copy when the front has one -- m > 0 --, else from the front itself: same values)
    const double *Ub = fd.epoff >= 0 ? pool + fd.epoff : F;
    const int64_t us = fd.epoff >= 0 ? p : f; // column stride of U
    const double *Ui = Ub + i + (int64_t)p * us;
    double e[8], a[16];
#pragma unroll
    for (int k = 0; k < 8; k++) e[k] = (i < p && jq + k * ng < m) ? Ui[(int64_t)(jq + k * ng) * us] : 0.0;
#pragma unroll
    for (int q = 0; q < 16; q++) {
        const int j = p - 1 - q;
        a[q] = (lane < p && j >= 0) ? Ub[lane + (int64_t)j * us] : 1.0;
    }
    // the reciprocal of the lane's own pivot, once (a division per pivot step was the larger part of the substitution's instructions);
    // requested and formed before the wait
    const double inv_d = (lane < p) ? 1.0 / Ub[lane + (int64_t)lane * us] : 1.0;
    if constexpr (!TAG && !PLAIN) {
        if (fd.parent >= 0 && lane == 0) sf_wait(done + fd.parent, need[fd.parent], err);
    }
    wave_sync();
    {
        double xv[K];
        if constexpr (TAG) { // the ancestors' solved entries come from the tagged shadow of x
            xv[0] = ld_agent(xt + myrow);
            if (lane < m) xv[0] = sf_tag_wait(xt + myrow, xv[0], err);
        } else
            ld_cols_p<K, PLAIN>(xv, x, xstr, myrow, nk); // (myrow = 0 for the lanes past the front's rows: a valid address, the value is dropped)
        if (lane < m) {
#pragma unroll
            for (int c = 0; c < K; c++)
                if (c < nk) xg[c][lane] = xv[c];
        }
    }
    wave_sync();
    double acc[K];
#pragma unroll
    for (int c = 0; c < K; c++) acc[c] = 0.0;
    if (i < p) {
#pragma unroll
        for (int k = 0; k < 8; k++)
            if (jq + k * ng < m) {
#pragma unroll
                for (int c = 0; c < K; c++)
                    if (c < nk) acc[c] += e[k] * xg[c][jq + k * ng];
            }
        for (int j = jq + 8 * ng; j < m; j += ng) {
            const double u = Ui[(int64_t)j * us];
#pragma unroll
            for (int c = 0; c < K; c++)
                if (c < nk) acc[c] += u * xg[c][j];
        }
    }
    double v[K];
#pragma unroll
    for (int c = 0; c < K; c++) {
        for (int off = 1 << sh; off < 64; off <<= 1) acc[c] += __shfl_xor(acc[c], off);
        v[c] = (lane < p) ? y1[c] - acc[c] : 0.0;
    }
    // x1 = U11^{-1} t, columns from right to left, the lane's row of U11 16 columns at a time
    for (int jhi = p; jhi > 0; jhi -= 16) {
#pragma unroll
        for (int q = 0; q < 16; q++) {
            const int j = jhi - 1 - q;
            if (j >= 0) { // wave-uniform
#pragma unroll
                for (int c = 0; c < K; c++) {
                    if (lane == j) v[c] *= inv_d;
                    const double vj = wave_bcast(v[c], j);
                    if (lane < j) v[c] -= a[q] * vj;
                }
            }
        }
        if (jhi > 16) {
#pragma unroll
            for (int q = 0; q < 16; q++) {
                const int j = jhi - 17 - q;
                a[q] = (lane < p && j >= 0) ? Ub[lane + (int64_t)j * us] : 1.0;
            }
        }
    }
    if (lane < p) {
#pragma unroll
        for (int c = 0; c < K; c++)
            if (c < nk) st_p<PLAIN>(xs + c * xstr + lane, v[c]);
        if constexpr (TAG) st_agent(xt + fd.first + lane, v[0]);
    }
    if constexpr (!TAG && !PLAIN) {
        drain_stores();
        if (lane == 0) flag_add(done + s, 1);
    }
}

// Strided dot products against K LDS vector chunks (column c at w + c * wld):
//   acc[c] += sum_j col[j * ld] * w[c * wld + j - c0],  j = j0, j0 + step, ... < j1.
// Sixteen unconditional loads are in flight per lane (then eight, then the last partial group of eight with clamped
// addresses: a loop of predicated loads compiles to a wait per load); the order of the additions is the one of
// kernels_solve.hpp's strided_dot: even positions into acc0, odd ones into acc1, the tail into acc0.
template <int K, bool WIDE = (K == 1)>
__device__ __forceinline__ void sf_dot(double (&acc0)[K], double (&acc1)[K], const double *__restrict__ col, int64_t ld, const double *w, int wld,
                                       int c0, int j0, int j1, int step) {
    int j = j0;
    const int nfull = (j1 - j0 + step - 1) / step / 8 * 8; // positions covered by whole groups of 8
    const int jend8 = j0 + nfull * step;
    // (K > 1: eight in flight -- the blocked instances sit at the edge of a register-occupancy step)
    for (; WIDE && j + 15 * step < jend8; j += 16 * step) {
        double e[16];
#pragma unroll
        for (int u = 0; u < 16; u++) e[u] = col[(int64_t)(j + u * step) * ld];
#pragma unroll
        for (int c = 0; c < K; c++)
#pragma unroll
            for (int u = 0; u < 16; u += 2) {
                acc0[c] += e[u] * w[c * wld + j + u * step - c0];
                acc1[c] += e[u + 1] * w[c * wld + j + (u + 1) * step - c0];
            }
    }
    for (; j + 7 * step < j1; j += 8 * step) {
        double e[8];
#pragma unroll
        for (int u = 0; u < 8; u++) e[u] = col[(int64_t)(j + u * step) * ld];
#pragma unroll
        for (int c = 0; c < K; c++)
#pragma unroll
            for (int u = 0; u < 8; u += 2) {
                acc0[c] += e[u] * w[c * wld + j + u * step - c0];
                acc1[c] += e[u + 1] * w[c * wld + j + (u + 1) * step - c0];
            }
    }
    if (j < j1) {
        double e[8];
#pragma unroll
        for (int u = 0; u < 8; u++) {
            const int jj = j + u * step < j1 ? j + u * step : j;
            e[u] = col[(int64_t)jj * ld];
        }
#pragma unroll
        for (int c = 0; c < K; c++)
#pragma unroll
            for (int u = 0; u < 8; u++)
                if (j + u * step < j1) acc0[c] += e[u] * w[c * wld + j + u * step - c0];
    }
}

// The children's update vectors are added into the LDS copies of w1 (chunk [c0, c1), column c at wc + c * wld) and, on
// the first chunk, into the slab's own rows wsl (column c at wsl + c * 128): NK children at a time, NE entries per
// thread and child fetched together (one round trip), then added child by child in ascending order (the order fixes
// the floating-point sums).  Longer children finish in a plain loop.
template <int NK, int NE, int K, bool TAG = false>
__device__ __forceinline__ void sf_children(int tid, int nch, int ncd, const int64_t *cd_woff, const int64_t *cd_rel, const int32_t *cd_m,
                                            const FrontDesc &fd, const FrontDesc *__restrict__ FD, const int32_t *__restrict__ child_idx,
                                            const int32_t *__restrict__ rel, const double *work, double *wc, int wld, double *wsl, int c0, int c1,
                                            int p, int r0, int r1, int nk, int64_t wstr, int *err = nullptr) {
    static_assert(!TAG || K == 1, "the tagged hand-offs carry one right-hand side");
    for (int cb = 0; cb < nch; cb += NK) {
        int qv[NK][NE];
        double uv[NK][NE][K];
        int cm[NK];
#pragma unroll
        for (int k = 0; k < NK; k++) {
            const int c = cb + k;
            cm[k] = 0;
            if (c < nch) {
                int64_t woff, relo;
                if (c < ncd) {
                    woff = cd_woff[c], relo = cd_rel[c], cm[k] = cd_m[c];
                } else {
                    const FrontDesc cd = FD[child_idx[fd.child_begin + c]];
                    woff = cd.woff + cd.p, relo = cd.rowptr, cm[k] = cd.m;
                }
#pragma unroll
                for (int e = 0; e < NE; e++) {
                    // (clamped index: the loads are unconditional -- see ld_cols; `work` and `rel` are padded by 64 entries, so
                    //  index 0 of a child without update rows is still inside the allocation)
                    const int i = tid + 256 * e;
                    if constexpr (K == 1) { // (one load per child and entry: the single-column instances keep the predicated form they were tuned with;
                                            //  the unconditional form costs 20 registers and an occupancy step there: forward 230 -> 250 us)
                        qv[k][e] = -1;
                        if (i < cm[k]) {
                            qv[k][e] = rel[relo + i];
                            uv[k][e][0] = ld_agent(work + woff + i);
                        }
                    } else {
                        const int ic = i < cm[k] ? i : 0;
                        const int qq = rel[relo + ic];
                        qv[k][e] = i < cm[k] ? qq : -1;
                        ld_cols<K>(uv[k][e], work, wstr, woff + ic, nk);
                    }
                }
                if (cm[k] > 256 * NE) cm[k] = -cm[k]; // the rest of this child's list goes through the plain loop below
            }
        }
#pragma unroll
        for (int k = 0; k < NK; k++) {
            if (cb + k < nch) { // workgroup-uniform
#pragma unroll
                for (int e = 0; e < NE; e++) {
                    const int q = qv[k][e];
                    if constexpr (TAG) {
                        // (a word that still holds the tag is re-loaded until the producing task has stored it; only the words this
                        //  thread is going to use are waited for)
                        const bool used = (q >= c0 && q < c1) || (c0 == 0 && q >= p && q >= r0 && q < r1);
                        if (used && sf_is_tag(uv[k][e][0])) {
                            const int c = cb + k;
                            const int64_t woff = c < ncd ? cd_woff[c] : FD[child_idx[fd.child_begin + c]].woff + FD[child_idx[fd.child_begin + c]].p;
                            uv[k][e][0] = sf_tag_wait(work + woff + tid + 256 * e, uv[k][e][0], err);
                        }
                    }
                    if (q >= c0 && q < c1) {
#pragma unroll
                        for (int cc = 0; cc < K; cc++)
                            if (cc < nk) wc[cc * wld + q - c0] += uv[k][e][cc];
                    } else if (c0 == 0 && q >= p && q >= r0 && q < r1) {
#pragma unroll
                        for (int cc = 0; cc < K; cc++)
                            if (cc < nk) wsl[cc * 128 + q - r0] += uv[k][e][cc];
                    }
                }
                if (cm[k] < 0) {
                    const int c = cb + k;
                    int64_t woff, relo;
                    if (c < ncd) {
                        woff = cd_woff[c], relo = cd_rel[c];
                    } else {
                        const FrontDesc cd = FD[child_idx[fd.child_begin + c]];
                        woff = cd.woff + cd.p, relo = cd.rowptr;
                    }
                    for (int i = tid + 256 * NE; i < -cm[k]; i += 256) {
                        const int q = rel[relo + i];
                        double tv[K];
                        ld_cols<K>(tv, work, wstr, woff + i, nk);
                        if constexpr (TAG) {
                            if ((q >= c0 && q < c1) || (c0 == 0 && q >= p && q >= r0 && q < r1)) tv[0] = sf_tag_wait(work + woff + i, tv[0], err);
                        }
                        if (q >= c0 && q < c1) {
#pragma unroll
                            for (int cc = 0; cc < K; cc++)
                                if (cc < nk) wc[cc * wld + q - c0] += tv[cc];
                        } else if (c0 == 0 && q >= p && q >= r0 && q < r1) {
#pragma unroll
                            for (int cc = 0; cc < K; cc++)
                                if (cc < nk) wsl[cc * 128 + q - r0] += tv[cc];
                        }
                    }
                }
                __syncthreads();
            }
        }
    }
}

// The same sums for the blocked instances (K > 1) when the children are many and short -- the lower levels of a 3D factor: six to ten
// children of 30 - 100 rows each, where sf_children's two children per round trip leave most of the 256 threads without an entry and
// the rounds (10 - 17 us of a ~27 us task, profiles/r05_solve_trace_200cube_sym_16col.txt) are what the task spends its time on.  Here a
// round takes as many consecutive children as fit on the 256 threads (at most eight), one row of one child per thread, everything
// fetched in one round trip; the adds then run child by child in ascending order as before: the same bits as sf_children.
// Requires the children's descriptors in LDS (nch <= 64) and no child of more than 256 rows.
template <int K>
__device__ __forceinline__ void sf_children_packed(int tid, int nch, const int64_t *cd_woff, const int64_t *cd_rel, const int32_t *cd_m,
                                                   const int32_t *__restrict__ rel, const double *work, double *wc, int wld, double *wsl, int c0,
                                                   int c1, int p, int r0, int r1, int nk, int64_t wstr) {
    int cb = 0;
    while (cb < nch) { // (workgroup-uniform)
        int ce = cb, tot = 0, mine = -1, myi = 0;
        while (ce < nch && ce - cb < 8 && tot + cd_m[ce] <= 256) {
            const int m = cd_m[ce];
            if (mine < 0 && tid < tot + m) mine = ce, myi = tid - tot;
            tot += m;
            ce++;
        }
        // (clamped addresses: the loads are unconditional -- see ld_cols; a thread without an entry re-reads entry 0 of the round's first child)
        const int ch = mine >= 0 ? mine : cb;
        const int qq = rel[cd_rel[ch] + myi];
        const int q = mine >= 0 ? qq : -1;
        double uv[K];
        ld_cols<K>(uv, work, wstr, cd_woff[ch] + myi, nk);
        for (int k = cb; k < ce; k++) {
            if (mine == k) {
                if (q >= c0 && q < c1) {
#pragma unroll
                    for (int cc = 0; cc < K; cc++)
                        if (cc < nk) wc[cc * wld + q - c0] += uv[cc];
                } else if (c0 == 0 && q >= p && q >= r0 && q < r1) {
#pragma unroll
                    for (int cc = 0; cc < K; cc++)
                        if (cc < nk) wsl[cc * 128 + q - r0] += uv[cc];
                }
            }
            __syncthreads();
        }
        cb = ce;
    }
}

// pairwise sum over the G <= 32 column groups of row rr in a fixed order
__device__ __forceinline__ double sf_group_sum(const double *red, int rows, int rr, int G) {
    double tsum[32];
#pragma unroll
    for (int q = 0; q < 32; q++) tsum[q] = (q < G) ? red[q * rows + rr] : 0.0;
#pragma unroll
    for (int wdt = 1; wdt < 32; wdt <<= 1)
#pragma unroll
        for (int q = 0; q + wdt < 32; q += 2 * wdt) tsum[q] += tsum[q + wdt];
    return tsum[0];
}

// ---- blocked instances (K > 1): the slab's dot products on v_mfma_f64_16x16x4_f64 ----
// A slab of `rows` = 16 nsub outputs against K <= 16 right-hand sides is nsub tiles  Y(16 x 16) += M(16 x 4) W(4 x 16)  per four
// positions of the contraction index: one load of M per lane and MFMA instead of K multiply-adds and K LDS reads per loaded entry
// (the scalar form made the blocked instances instruction-bound: 8 columns cost 3x one column).  Half of every MFMA works on the
// zero columns K..15; at 64 cycles per instruction the matrix pipe still takes 512 B of M per 64 cycles and SIMD, above what HBM
// delivers.  The summation order differs from the single-column kernels': blocked and single solves agree to rounding, not bit for bit.
//   TRANS = false:  output o, position k  ->  M[o + k ld]   (forward E, backward E': outputs run along memory)
//   TRANS = true:   output o, position k  ->  M[k + o ld]   (backward E^T of the L D L^T fronts: positions run along memory)
// Work split over the four wavefronts: nsub >= 4: wave w owns the tiles w, w + 4 for every position; nsub < 4: 4 / nsub waves share a
// tile and split the positions of each chunk into contiguous parts (partial tiles are added in a fixed order by sf_mma_finish).
// w: the chunk [c0, c1) of the K vectors in LDS, column c at w + c wld, position k at index k - c0.
template <bool TRANS, int DEPTH = 8>
__device__ __forceinline__ void sf_mma_chunk(f64x4 (&acc)[2], const double *__restrict__ M, int64_t ld, const double *w, int wld, int c0, int c1,
                                             int nk, int nsub, int nout, int wave, int lane) {
    const int o = lane & 15, kk = lane >> 4;
    const int ksplit = nsub >= 4 ? 1 : 4 / nsub;
    const int part = nsub >= 4 ? 0 : wave / nsub;
    int ka = c0, kb = c1;
    if (ksplit > 1) {
        const int len = ((c1 - c0 + ksplit - 1) / ksplit + 3) & ~3;
        ka = c0 + part * len;
        kb = ka + len < c1 ? ka + len : c1;
    }
    const double wmask = o < nk ? 1.0 : 0.0; // (columns >= nk of the W tile are zero)
    const double *wl = w + (o < nk ? o : 0) * wld - c0;
#pragma unroll
    for (int tq = 0; tq < 2; tq++) {
        const int tile = nsub >= 4 ? wave + 4 * tq : wave % nsub;
        if (tile >= nsub || (nsub < 4 && tq > 0)) continue; // (wave-uniform)
        const int oo = 16 * tile + o;
        const int oc = oo < nout ? oo : nout - 1; // outputs past the slab's end: clamped address, result discarded
        const double *Mo = TRANS ? M + (int64_t)oc * ld : M + oc;
        const int64_t ks = TRANS ? 1 : ld;
        // FOUR partial tiles per chunk, the MFMAs dealt to them in turn, added pairwise at the end of the chunk: one accumulator summed
        // every position of a dot product one after the other -- hundreds to thousands of dependent additions where the single-column
        // kernels add ~25 terms per thread and then reduce a tree -- and the blocked first solve came out with 5 - 10 x the componentwise
        // backward error of the single-column one (C2, random right-hand sides: 2.6e-14 against 2.9e-15, tools/omega_probe.py): above the
        // 64 eps below which ONE refinement step is taken without a second look, so every block paid a third pass pair.  The four chains
        // are independent instructions for the matrix pipe as well.
        f64x4 part[4] = {{0.0, 0.0, 0.0, 0.0}, {0.0, 0.0, 0.0, 0.0}, {0.0, 0.0, 0.0, 0.0}, {0.0, 0.0, 0.0, 0.0}};
        int k = ka;
        for (; k + 4 * DEPTH <= kb; k += 4 * DEPTH) { // DEPTH loads of M in flight per lane
            double a[DEPTH], b[DEPTH];
#pragma unroll
            for (int u = 0; u < DEPTH; u++) a[u] = Mo[(int64_t)(k + 4 * u + kk) * ks];
#pragma unroll
            for (int u = 0; u < DEPTH; u++) b[u] = wl[k + 4 * u + kk] * wmask;
#pragma unroll
            for (int u = 0; u < DEPTH; u++) part[u & 3] = mfma_f64_16x16x4(a[u], b[u], part[u & 3]);
        }
        for (; k < kb; k += 4) { // (fewer than DEPTH positions: one chain)
            const int kq = k + kk;
            const bool in = kq < kb;
            const double a = Mo[(int64_t)(in ? kq : kb - 1) * ks];
            const double b = in ? wl[kq] * wmask : 0.0;
            part[0] = mfma_f64_16x16x4(a, b, part[0]);
        }
#pragma unroll
        for (int g = 0; g < 4; g++) acc[tq][g] += (part[0][g] + part[1][g]) + (part[2][g] + part[3][g]);
    }
}

// The tiles go to LDS (mt: 8 tiles x 256 doubles; tile slot = tile + nsub * part), then thread rr < rows adds the parts of its row in a
// fixed order: result of output rr, column c.  Call sf_mma_store, __syncthreads(), then sf_mma_sum.
__device__ __forceinline__ void sf_mma_store(const f64x4 (&acc)[2], double *mt, int nsub, int wave, int lane) {
#pragma unroll
    for (int tq = 0; tq < 2; tq++) {
        const int tile = nsub >= 4 ? wave + 4 * tq : wave % nsub;
        if (tile >= nsub || (nsub < 4 && tq > 0)) continue;
        const int slot = nsub >= 4 ? tile : tile + nsub * (wave / nsub);
#pragma unroll
        for (int g = 0; g < 4; g++) mt[slot * 256 + ((lane >> 4) + 4 * g) * 16 + (lane & 15)] = acc[tq][g]; // [output][column]
    }
}
__device__ __forceinline__ double sf_mma_sum(const double *mt, int nsub, int rr, int c) {
    const int tile = rr >> 4, o = rr & 15;
    if (nsub >= 4) return mt[tile * 256 + o * 16 + c];
    double s = mt[tile * 256 + o * 16 + c];
    for (int part = 1; part < 4 / nsub; part++) s += mt[(tile + nsub * part) * 256 + o * 16 + c];
    return s;
}

// Forward pass, one launch per band of levels.  sync[SF_SYNC_HEADER + s] = completed tasks of front s (zeroed before
// every pass); *err is sticky: set when a wait timed out.
// STG (K = 1 only): the instance that runs ABOVE the wave-subtrees (kernels_solve_tree.hpp).  The upper levels of the tree are a chain
// of dependent hand-offs with a few megabytes of E per level: what a slab does AFTER its children are complete must be short.  Every
// thread parks the first `stage` entries of its share of its row of E (positions g, g + G, ... of row r) in dynamic LDS -- stage x 256
// doubles, slot [u][tid]: private to the thread, no barrier, conflict-free -- BEFORE it waits; after the wait the dot product reads
// them back in the same order and with the same two accumulators as sf_dot (bit-identical sums).  Registers cannot hold them: the
// kernel also runs thousands of small fronts, whose occupancy pays for every VGPR (measured in round 2: 108 -> 152 VGPRs, slower).
// TAG (K = 1 only): data-tagged hand-offs (see sf_tag_wait) -- no completion counters, no drains; the task list then holds no
// ASSEMBLE tasks (their intermediate result would sit where the parent looks for the final one).
#ifndef HIPMF_PACKED_GATHER
#define HIPMF_PACKED_GATHER 1 // (0: A/B builds without sf_children_packed)
#endif
#ifndef HIPMF_SF_FWD_WGS
#define HIPMF_SF_FWD_WGS 3 // workgroups per compute unit the single-column forward instances above the wave-subtrees are compiled for (A/B builds)
#endif
template <bool SMALL_ONLY, int K, bool STG = false, bool TAG = false, bool PLAIN = false>
__global__ void __launch_bounds__(256, (K == 1 && !SMALL_ONLY) ? HIPMF_SF_FWD_WGS : 3) k_fwd_fused(const SfTask *__restrict__ tasks, const FrontDesc *__restrict__ FD,
                                                   const double *__restrict__ pool, const int32_t *__restrict__ lperm,
                                                   const int32_t *__restrict__ child_idx, const int32_t *__restrict__ rel,
                                                   const int32_t *__restrict__ need, int *sync, int *err, double *work, double *x, int nk,
                                                   int64_t xstr, int64_t wstr, unsigned long long *trace, int stage,
                                                   const int32_t *__restrict__ rep_idx, int *rep, SfGroups GR) {
    static_assert(!STG || (K == 1 && !SMALL_ONLY), "the staged instance carries one right-hand side");
    static_assert(!TAG || (K == 1 && !SMALL_ONLY), "the tagged instance carries one right-hand side");
    HIPMF_DYN_SHARED(double, els); // STG: stage x 256 doubles
    int bid, grp;
    if (!sf_group_of<K>(GR, bid, grp, nk)) return;
    if constexpr (K > 1) {
        if (grp > 0) { // (workgroup-uniform) this group's columns, workspaces and completion counters
            x += (int64_t)grp * K * xstr, work += (int64_t)grp * K * wstr, sync += (int64_t)grp * GR.sync_stride;
            trace = nullptr;
        }
    }
    constexpr int CHK = K > 8 ? SF_CHUNK / 4 : (K > 4 ? SF_CHUNK / 2 : SF_CHUNK); // chunk of w1 per right-hand side
    // One LDS buffer, three uses that never overlap in time: a workgroup either runs four small fronts (wv: K x 64 doubles per wave)
    // or one slab of a big front (wc: the chunk of the K vectors; mt: the slab's MFMA tiles, written after the last chunk is consumed).
    // As separate arrays they added up to 75 KB in the blocked instance: two workgroups per CU.
    constexpr int LDS_D = SMALL_ONLY ? 4 * K * 64 : (K * CHK > 4 * K * 64 ? K * CHK : 4 * K * 64);
    static_assert(SMALL_ONLY || K == 1 || K * CHK >= 8 * 256, "the MFMA tiles share the chunk buffer");
    __shared__ double lds[LDS_D];
    double(*wv)[K][64] = reinterpret_cast<double(*)[K][64]>(lds);
    double *wc = lds, *mt = lds;
    __shared__ double wsl[SMALL_ONLY ? 1 : K * 128];
    __shared__ double red[256];
    __shared__ int64_t cd_woff[SMALL_ONLY ? 1 : 64], cd_rel[SMALL_ONLY ? 1 : 64];
    __shared__ int32_t cd_m[SMALL_ONLY ? 1 : 64];
    __shared__ int32_t cm_max_s;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    int *done = sync + SF_SYNC_HEADER;
    const SfTask t = tasks[bid];
    if (SMALL_ONLY || t.kind == 0) {
        // (wave-uniform: the front's descriptor then lives in scalar registers and the panel loads use scalar bases)
        const int s = wave_uniform(wave == 0 ? t.a : (wave == 1 ? t.b : (wave == 2 ? t.c : t.d)));
        static_assert(!PLAIN || SMALL_ONLY, "plain loads / stores: the per-level launches of the all-small band only");
        if (s >= 0) sf_fwd_small<K, TAG, PLAIN>(s, lane, wv[wave], FD, pool, lperm, child_idx, rel, need, done, err, work, x, nk, xstr, wstr);
        return;
    }
    if (SMALL_ONLY) return; // (never reached: the small-only instance is launched on all-small bands)
    if constexpr (K == 1 && !SMALL_ONLY) {
        if (t.kind == 2) { // wave fronts: one big front of few rows and pivots per wavefront
            const int s = wave_uniform(wave == 0 ? t.a : (wave == 1 ? t.b : (wave == 2 ? t.c : t.d)));
            if (s >= 0) sf_fwd_wave<TAG>(s, lane, lds + SF_WF_ROWS * wave, FD, pool, child_idx, rel, need, done, err, work, x);
            return;
        }
    }
    if (!TAG && t.kind == 1) {
        // ---- assemble rows [q0, q1) of the big front t.a: w = b (pivot rows) + the children's updates, once for all slabs.  Every
        //      slab used to gather all children itself: hundreds of redundant gathers on fronts of thousands of rows.  The pivot part
        //      goes back into x (the slabs read their w1 from there), the rest into the front's work vector (the slabs add E w1). ----
        const FrontDesc fd = FD[t.a];
        const int p = fd.p;
        const int q0 = t.b, q1 = t.c, nq = q1 - q0; // nq <= CHK
        double *W = work + fd.woff;
        const int nch = fd.child_end - fd.child_begin;
        for (int i = tid; i < nq; i += 256) {
            double xv[K];
            ld_cols<K>(xv, x, xstr, fd.first + (q0 + i < p ? q0 + i : 0), nk);
#pragma unroll
            for (int c = 0; c < K; c++)
                if (c < nk) wc[c * CHK + i] = (q0 + i < p) ? xv[c] : 0.0;
        }
        for (int cb = 0; cb < nch; cb += 256) { // wait for the children (one per thread)
            if (cb + tid < nch) {
                const int ch = child_idx[fd.child_begin + cb + tid];
                sf_wait(done + ch, need[ch], err);
            }
        }
        __syncthreads();
        __shared__ int32_t a_lo[64], a_hi[64], a_m[64];
        __shared__ int64_t a_woff[64], a_rel[64];
        for (int cb = 0; cb < nch; cb += 64) {
            // the entries of up to 64 children that fall into [q0, q1): relative indices ascend, two binary searches per child
            if (tid < 64 && cb + tid < nch) {
                const FrontDesc cd = FD[child_idx[fd.child_begin + cb + tid]];
                const int32_t *rl = rel + cd.rowptr;
                a_lo[tid] = lower_bound_i32(rl, cd.m, q0);
                a_hi[tid] = lower_bound_i32(rl, cd.m, q1);
                a_woff[tid] = cd.woff + cd.p;
                a_rel[tid] = cd.rowptr;
                a_m[tid] = cd.m;
            }
            __syncthreads();
            const int nb = nch - cb < 64 ? nch - cb : 64;
            for (int k = 0; k < nb; k++) { // children in ascending order: the order fixes the floating-point sums
                const int lo = a_lo[k], hi = a_hi[k];
                const int64_t woff = a_woff[k], relo = a_rel[k];
                for (int i = lo + tid; i < hi; i += 256) {
                    const int q = rel[relo + i] - q0;
                    double tv[K];
                    ld_cols<K>(tv, work, wstr, woff + i, nk);
#pragma unroll
                    for (int c = 0; c < K; c++)
                        if (c < nk) wc[c * CHK + q] += tv[c];
                }
                __syncthreads();
            }
        }
        for (int i = tid; i < nq; i += 256) {
#pragma unroll
            for (int c = 0; c < K; c++)
                if (c < nk) {
                    if (q0 + i < p) st_agent(x + c * xstr + fd.first + q0 + i, wc[c * CHK + i]);
                    else st_agent(W + c * wstr + q0 + i, wc[c * CHK + i]);
                }
        }
        drain_stores();
        __syncthreads();
        if (tid == 0) flag_add(done + t.a, 1);
        return;
    }
    // ---- slab [r0, r1) of the big front t.a:  [y1; -delta] = E w1,  work[r] = y1[r] (r < p) or w2[r] + (E w1)[r] ----
    unsigned long long tr0 = 0, tr1 = 0, tr2 = 0, tr_g = 0, tr_d = 0;
    if (trace && tid == 0) tr0 = dev_clock();
    const FrontDesc fd = FD[t.a];
    const int p = fd.p, f = fd.p + fd.m;
    const int64_t ld = fd.ld;
    const double *E = pool + fd.eoff;
    double *W = work + fd.woff;
    const int r0 = t.b, r1 = t.c, sh = t.kind;
    const int rr = tid & ((1 << sh) - 1), g = tid >> sh, G = 256 >> sh;
    const int r = r0 + rr;
    for (int i = tid; i < K * 128; i += 256) wsl[i] = 0.0;
    // rows of inv(L11) P are zero right of their own 32-column block
    int jmax = p;
    if (r1 <= p && !(fd.flags & FD_DENSE_TOP)) jmax = ((r1 - 1) / NB + 1) * NB < p ? ((r1 - 1) / NB + 1) * NB : p; // (k_front leaves a full block)
    // (K == 1) the first eight entries of this lane's row of E are on their way while the workgroup waits for the children
    constexpr int NPRE = 8;
    double e_pre[K == 1 ? NPRE : 1];
    const int c1_first = jmax < CHK ? jmax : CHK;
    const bool use_pre = K == 1 && !STG && r < r1 && (c1_first - g + G - 1) / G >= NPRE; // at least NPRE positions in the first chunk
    if (K == 1 && use_pre) {
#pragma unroll
        for (int u = 0; u < NPRE; u++) e_pre[K == 1 ? u : 0] = E[r + (int64_t)(g + u * G) * ld];
    }
    if (STG) {
        // (clamped addresses: unconditional loads, eight in flight; entries past the end of the row are never read back)
        const double *Er = E + (r < r1 ? r : r1 - 1);
        for (int u0 = 0; u0 < stage; u0 += 8) {
            double t[8];
#pragma unroll
            for (int q = 0; q < 8; q++) {
                const int j = g + (u0 + q) * G;
                t[q] = Er[(int64_t)(j < jmax ? j : jmax - 1) * ld];
            }
#pragma unroll
            for (int q = 0; q < 8; q++) els[(u0 + q) * 256 + tid] = t[q];
        }
    }
    // ---- before the wait: everything that does not depend on the children ----
    //  * the children's descriptors, one child per lane of wave 0, parked in LDS for all waves
    //  * b1 = the first chunk of x (set before the launch; the children write `work`, not x)
    const int nasm = t.d; // > 0: the front's vector is assembled by its own tasks (kind 1); this slab waits for them, not for the children
    const int nch = nasm > 0 ? 0 : fd.child_end - fd.child_begin;
    const int ncd = nch < 64 ? nch : 64; // children with a parked descriptor
    if (nasm > 0) {
        if (tid == 0) sf_wait(done + t.a, nasm, err);
        __syncthreads();
    }
    for (int i = tid; i < (jmax < CHK ? jmax : CHK); i += 256) {
        double xv[K];
        ld_cols<K>(xv, x, xstr, fd.first + i, nk);
#pragma unroll
        for (int c = 0; c < K; c++)
            if (c < nk) wc[c * CHK + i] = xv[c];
    }
    if (nasm > 0) { // (workgroup-uniform)
        // the assembled update part of the slab's own rows (what the children sweep leaves in wsl otherwise)
        const bool mine = r0 + (tid & 127) < r1 && r0 + (tid & 127) >= p && tid < 128;
        if constexpr (K == 1) {
            if (mine) wsl[tid] = ld_agent(W + r0 + tid);
        } else {
            double av[K];
            ld_cols<K>(av, W, wstr, r0 + (mine ? tid : 0), nk);
            if (mine) {
#pragma unroll
                for (int c = 0; c < K; c++)
                    if (c < nk) wsl[c * 128 + tid] = av[c];
            }
        }
    }
    if (wave == 0) {
        int mym = 0;
        if (lane < ncd) {
            const int ch = child_idx[fd.child_begin + lane];
            const FrontDesc cd = FD[ch];
            cd_woff[lane] = cd.woff + cd.p;
            cd_rel[lane] = cd.rowptr;
            cd_m[lane] = cd.m;
            mym = cd.m;
        }
        const int mx = (int)wave_max_u32((unsigned)mym);
        if (lane == 0) cm_max_s = nch > 64 ? 0x7fffffff : mx;
        if constexpr (!TAG) {
            if (lane < ncd) {
                const int ch = child_idx[fd.child_begin + lane];
                sf_wait_front(ch, need, done, STG ? rep_idx : nullptr, rep, err);
            }
            for (int c0 = 64; c0 < nch; c0 += 64)
                if (c0 + lane < nch) {
                    const int ch = child_idx[fd.child_begin + c0 + lane];
                    sf_wait_front(ch, need, done, STG ? rep_idx : nullptr, rep, err);
                }
        }
    }
    __syncthreads();
    if (trace && tid == 0) tr1 = dev_clock();
    // ---- after the wait ----
    const int cm_max = cm_max_s;
    double acc0[1] = {0.0}, acc1[1] = {0.0}; // (K = 1: the scalar dot products)
    f64x4 macc[2] = {{0.0, 0.0, 0.0, 0.0}, {0.0, 0.0, 0.0, 0.0}};
    const int nsub = (1 << sh) >> 4; // 16-output tiles of the slab
    for (int c0 = 0; c0 < jmax; c0 += CHK) {
        const int c1 = c0 + CHK < jmax ? c0 + CHK : jmax;
        // w1[c0, c1) = b1 + the children's updates to these pivot rows (children in ascending order)
        if (c0 > 0) {
            for (int i = c0 + tid; i < c1; i += 256) {
                double xv[K];
                ld_cols<K>(xv, x, xstr, fd.first + i, nk);
#pragma unroll
                for (int c = 0; c < K; c++)
                    if (c < nk) wc[c * CHK + i - c0] = xv[c];
            }
            __syncthreads();
        }
        if (K > 1 && HIPMF_PACKED_GATHER && nch > (K <= 8 ? 4 : 2) && nch <= 64 && cm_max <= 256) {
            if constexpr (K > 1) sf_children_packed<K>(tid, nch, cd_woff, cd_rel, cd_m, rel, work, wc, CHK, wsl, c0, c1, p, r0, r1, nk, wstr);
        } else if (cm_max <= 256)
            sf_children<(K == 1 ? 8 : (K <= 8 ? 4 : 2)), 1, K, TAG>(tid, nch, ncd, cd_woff, cd_rel, cd_m, fd, FD, child_idx, rel, work, wc, CHK, wsl, c0, c1, p, r0, r1, nk, wstr, err);
        else // (top-level instance: six entries per thread and child in one round trip -- 1 536 rows)
            sf_children<2, (STG ? 6 : (K == 1 ? 4 : (K <= 8 ? 2 : 1))), K, TAG>(tid, nch, ncd, cd_woff, cd_rel, cd_m, fd, FD, child_idx, rel, work, wc, CHK, wsl, c0, c1, p, r0, r1, nk, wstr, err);
        if (nch == 0) __syncthreads();
        if (trace && tid == 0 && c0 == 0) tr_g = dev_clock();
        // the group's columns of this chunk: g, g + G, ... continue across chunks (CHK is a multiple of every G)
        if (STG && c0 == 0) {
            if (r < r1) {
                // this thread's positions of the chunk: g, g + G, ...; whole groups of eight alternate between the two accumulators,
                // the last partial group goes into acc0 (sf_dot's order); the first `stage` of them come back from LDS
                const int n0 = g < c1 ? (c1 - g + G - 1) / G : 0, nfull = n0 / 8 * 8, nst = stage < n0 ? stage : n0;
                int sq = 0;
                for (; sq + 8 <= nst && sq + 8 <= nfull; sq += 8) {
                    double e[8];
#pragma unroll
                    for (int q = 0; q < 8; q++) e[q] = els[(sq + q) * 256 + tid];
#pragma unroll
                    for (int q = 0; q < 8; q += 2) {
                        acc0[0] += e[q] * wc[g + (sq + q) * G];
                        acc1[0] += e[q + 1] * wc[g + (sq + q + 1) * G];
                    }
                }
                if (sq + 8 <= nfull) sf_dot<1, true>(acc0, acc1, E + r, ld, wc, CHK, 0, g + sq * G, c1, G);
                else
                    for (; sq < n0; sq++) {
                        const double ev = sq < stage ? els[sq * 256 + tid] : E[r + (int64_t)(g + sq * G) * ld];
                        acc0[0] += ev * wc[g + sq * G];
                    }
            }
        } else if (K == 1 && c0 == 0 && use_pre) {
            // consume the prefetched entries exactly as sf_dot would (even positions into acc0, odd ones into acc1), then go on
#pragma unroll
            for (int u = 0; u < NPRE; u += 2) {
                acc0[0] += e_pre[K == 1 ? u : 0] * wc[g + u * G];
                acc1[0] += e_pre[K == 1 ? u + 1 : 0] * wc[g + (u + 1) * G];
            }
            sf_dot<1, true>(acc0, acc1, E + r, ld, wc, CHK, c0, c0 + g + NPRE * G, c1, G);
        } else if (K > 1) {
            sf_mma_chunk<false>(macc, E + r0, ld, wc, CHK, c0, c1, nk, nsub, r1 - r0, wave, lane);
        } else if (r < r1)
            sf_dot<1, true>(acc0, acc1, E + r, ld, wc, CHK, c0, c0 + g, c1, G);
        __syncthreads();
    }
    if (trace && tid == 0) tr_d = dev_clock();
    if (K > 1) {
        sf_mma_store(macc, mt, nsub, wave, lane);
        __syncthreads();
        if (tid < (1 << sh) && r0 + tid < r1) {
            const int ro = r0 + tid;
#pragma unroll
            for (int c = 0; c < K; c++)
                if (c < nk) {
                    const double tot = sf_mma_sum(mt, nsub, tid, c);
                    st_agent(W + c * wstr + ro, (ro < p) ? tot : wsl[c * 128 + tid] + tot);
                }
        }
    } else {
        red[g * (1 << sh) + rr] = acc0[0] + acc1[0];
        __syncthreads();
        if (g == 0 && r < r1) {
            const double tot = sf_group_sum(red, 1 << sh, rr, G);
            st_agent(W + r, (r < p) ? tot : wsl[rr] + tot);
        }
    }
    if (trace && tid == 0) tr2 = dev_clock();
    if constexpr (!TAG) {
        drain_stores();
        __syncthreads();
        if (tid == 0) sf_publish_front(t.a, need, done, STG ? rep_idx : nullptr, rep);
    }
    if (trace && tid == 0) {
        unsigned long long *tr = trace + 8 * (size_t)bid;
        tr[0] = tr0, tr[1] = tr1, tr[2] = tr2, tr[3] = dev_clock(), tr[4] = tr_g, tr[5] = tr_d;
    }
}

// ---- backward step of one WAVE FRONT by one wavefront (round 5) ----
// The mirror image of sf_fwd_wave for the LU fronts:  x1 = E' [y1; x2]  with p <= SF_WF_PIV outputs and f <= SF_WF_ROWS positions.  Lane
// (i = lane mod 32, h = lane / 32) owns output i and the positions h, h + 2, ...; the first SF_WB_PRE of its entries of row i of E' are
// requested before anything is waited for; y1 (forward launch) and the ancestors' entries x2 (the tagged shadow xt under TAG, else x after
// the parent's completion count) go to the wave's LDS copy of the front's vector; two accumulators per lane, the halves added by one
// shuffle.  Same result as the slab tasks to rounding (another summation order).
constexpr int SF_WB_PRE = 16;
template <bool TAG>
__device__ __forceinline__ void sf_bwd_wave(int s, int lane, double *w, const FrontDesc *__restrict__ FD, const double *__restrict__ pool,
                                            const int32_t *__restrict__ rows, const int32_t *__restrict__ need, int *done, int *err,
                                            const double *work, double *x, double *xt) {
    const FrontDesc fd = FD[s];
    const int p = fd.p, f = fd.p + fd.m;
    const int64_t ld = fd.ldp;
    const double *Ep = pool + fd.epoff;
    const double *W = work + fd.woff; // y1: written by the forward launch
    const int32_t *rws = rows + fd.rowptr;
    const int i = lane & 31, h = lane >> 5;
    const double *Ei = Ep + (i < p ? i : p - 1);
    double e[SF_WB_PRE];
#pragma unroll
    for (int c = 0; c < SF_WB_PRE; c++) {
        const int j = h + 2 * c;
        e[c] = Ei[(int64_t)(j < f ? j : f - 1) * ld];
    }
    // the front's vector: y1 on the pivot positions, x2 (row numbers first) below; positions lane and lane + 64
    const int j0 = lane, j1 = lane + 64;
    const double y0 = W[j0 < p ? j0 : 0];
    const int row0 = (j0 >= p && j0 < f) ? rws[j0 - p] : -1, row1 = (j1 < f) ? rws[j1 - p] : -1; // (p <= 32 < 64 <= j1)
    if constexpr (!TAG) {
        if (fd.parent >= 0 && lane == 0) sf_wait(done + fd.parent, need[fd.parent], err);
        wave_sync();
    }
    double v0, v1;
    if constexpr (TAG) {
        v0 = ld_agent(xt + (row0 >= 0 ? row0 : 0)), v1 = ld_agent(xt + (row1 >= 0 ? row1 : 0));
        if (row0 >= 0) v0 = sf_tag_wait(xt + row0, v0, err);
        if (row1 >= 0) v1 = sf_tag_wait(xt + row1, v1, err);
    } else {
        v0 = ld_agent(x + (row0 >= 0 ? row0 : 0)), v1 = ld_agent(x + (row1 >= 0 ? row1 : 0));
    }
    w[j0] = j0 < p ? y0 : (row0 >= 0 ? v0 : 0.0);
    w[j1] = row1 >= 0 ? v1 : 0.0;
    wave_sync();
    double a = 0.0, b = 0.0;
#pragma unroll
    for (int c = 0; c < SF_WB_PRE; c += 2) {
        if (2 * c < f) { // (wave-uniform bound; positions past f multiply a clamped entry by the zero the vector holds there)
            a += e[c] * w[h + 2 * c];
            b += e[c + 1] * w[h + 2 * c + 2];
        }
    }
    for (int c0 = SF_WB_PRE; 2 * c0 < f; c0 += SF_WB_PRE) { // (wave-uniform)
#pragma unroll
        for (int c = 0; c < SF_WB_PRE; c++) {
            const int j = h + 2 * (c0 + c);
            e[c] = Ei[(int64_t)(j < f ? j : f - 1) * ld];
        }
#pragma unroll
        for (int c = 0; c < SF_WB_PRE; c += 2) {
            const int ja = h + 2 * (c0 + c), jb = ja + 2;
            a += e[c] * (ja < SF_WF_ROWS ? w[ja] : 0.0);
            b += e[c + 1] * (jb < SF_WF_ROWS ? w[jb] : 0.0);
        }
    }
    double tot = a + b;
    tot += __shfl_xor(tot, 32);
    if (lane < p) { // (h = 0: lanes 0 .. p - 1)
        st_agent(x + fd.first + lane, tot);
        if constexpr (TAG) st_agent(xt + fd.first + lane, tot);
    }
    if constexpr (!TAG) {
        drain_stores();
        if (lane == 0) flag_add(done + s, 1);
    }
}

// Backward pass, one launch per band of levels (tasks ordered root first).
// SYM: instance for factors whose big fronts are L D L^T (x1 = E^T [D^{-1} y1; x2], transposed GEMV); the LU instance carries none of it.
// TAG (K = 1 only): data-tagged hand-offs -- a front's solved pivot entries also go to the tagged shadow `xt` of x (all tag words before
// the launch), and that is where the fronts below read their x2 from, re-loading what is not there yet (see sf_tag_wait).
template <bool SMALL_ONLY, int K, bool SYM, bool STG = false, bool TAG = false, bool PLAIN = false>
__global__ void __launch_bounds__(256, 3) k_bwd_fused(const SfTask *__restrict__ tasks, const FrontDesc *__restrict__ FD,
                                                   const double *__restrict__ pool, const int32_t *__restrict__ rows,
                                                   const int32_t *__restrict__ need, int *sync, int *err, const double *work, double *x, int nk,
                                                   int64_t xstr, int64_t wstr, unsigned long long *trace, const double *__restrict__ diag,
                                                   int stage, const int32_t *__restrict__ rep_idx, int *rep, double *xt, double *split_scr,
                                                   int *split_cnt, SfGroups GR) {
    static_assert(!STG || (K == 1 && !SMALL_ONLY), "the staged instance carries one right-hand side");
    static_assert(!TAG || (K == 1 && !SMALL_ONLY), "the tagged instance carries one right-hand side");
    HIPMF_DYN_SHARED(double, els); // STG: stage x 256 doubles (see k_fwd_fused)
    int bid, grp;
    if (!sf_group_of<K>(GR, bid, grp, nk)) return;
    if constexpr (K > 1) {
        if (grp > 0) { // (workgroup-uniform; see k_fwd_fused)
            x += (int64_t)grp * K * xstr, work += (int64_t)grp * K * wstr, sync += (int64_t)grp * GR.sync_stride;
            if (split_scr) split_scr += (int64_t)grp * GR.split_stride * 256, split_cnt += (int64_t)grp * GR.split_stride;
            trace = nullptr;
        }
    }
    constexpr int CHK = K > 8 ? SF_CHUNK / 4 : (K > 4 ? SF_CHUNK / 2 : SF_CHUNK);
    // (one LDS buffer for the small fronts' vectors, the chunk of a big front's vectors and its MFMA tiles: see k_fwd_fused)
    constexpr int LDS_D = SMALL_ONLY ? 4 * K * 64 : (K * CHK > 4 * K * 64 ? K * CHK : 4 * K * 64);
    static_assert(SMALL_ONLY || K == 1 || K * CHK >= 8 * 256, "the MFMA tiles share the chunk buffer");
    __shared__ double lds[LDS_D];
    double(*wv)[K][64] = reinterpret_cast<double(*)[K][64]>(lds);
    double *wc = lds, *mt = lds;
    __shared__ double red[256];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    int *done = sync + SF_SYNC_HEADER;
    const SfTask t = tasks[bid];
    if (SMALL_ONLY || t.kind == 0) {
        const int s = wave_uniform(wave == 0 ? t.a : (wave == 1 ? t.b : (wave == 2 ? t.c : t.d)));
        static_assert(!PLAIN || SMALL_ONLY, "plain loads / stores: the per-level launches of the all-small band only");
        if (s >= 0) sf_bwd_small<K, TAG, PLAIN>(s, lane, wv[wave], FD, pool, rows, need, done, err, x, nk, xstr, xt);
        return;
    }
    if (SMALL_ONLY) return;
    if constexpr (K == 1 && !SMALL_ONLY && !SYM) {
        if (t.kind == 2) { // wave fronts: one big front of few rows and pivots per wavefront
            const int s = wave_uniform(wave == 0 ? t.a : (wave == 1 ? t.b : (wave == 2 ? t.c : t.d)));
            if (s >= 0) sf_bwd_wave<TAG>(s, lane, lds + SF_WF_ROWS * wave, FD, pool, rows, need, done, err, work, x, xt);
            return;
        }
    }
    // ---- pivot rows [r0, r1) of the big front t.a:  x1 = E' [y1; x2] ----
    unsigned long long tr0 = 0, tr1 = 0, tr2 = 0, tr_g = 0, tr_d = 0;
    if (trace && tid == 0) tr0 = dev_clock();
    const FrontDesc fd = FD[t.a];
    const int p = fd.p, f = fd.p + fd.m;
    const bool sym = SYM && (fd.flags & FD_SYM) != 0; // L D L^T front: x1 = E^T [D^{-1} y1; x2]
    const int64_t ld = fd.ldp;
    const double *Ep = sym ? pool + fd.eoff : pool + fd.epoff;
    const double *W = work + fd.woff; // y1: written by the forward launch
    const int32_t *rws = rows + fd.rowptr;
    const int r0 = t.b, r1 = t.c, sh = t.kind;
    const int rr = tid & ((1 << sh) - 1), g = tid >> sh, G = 256 >> sh;
    const int i = r0 + rr;
    // columns of inv(U11) left of the slab's first 32-column block are zero
    const int jmin0 = (r0 / NB) * NB;
    // SPLIT dot products (blocked instances, levels of few tasks: numeric.cpp).  The top levels of a 3D factor hold a few hundred slabs
    // of 16 rows whose dot products run over tens of thousands of positions: one workgroup per compute unit, each with ~32 KB in flight,
    // leaves most of the memory system idle (144^3, 16 columns: the top eleven levels were half of the backward pass).  A slab's
    // positions [jmin0, f) are dealt to Q consecutive tasks in contiguous ranges (multiples of the chunk); each leaves its partial tile
    // sums in a scratch, the LAST one to arrive (a counter per slab; it resets the counter for the next pass) adds the Q partial sums
    // in the order of the parts -- the result does not depend on who arrives last -- and finishes the slab like an unsplit task.
    const int Q = K > 1 ? ((t.part >> 8) & 0xff) : 0, qpart = t.part & 0xff;
    int jmin = jmin0, jend = f;
    if (K > 1 && Q > 1) {
        const int len = (((f - jmin0 + Q - 1) / Q + CHK - 1) / CHK) * CHK;
        jmin = jmin0 + qpart * len < f ? jmin0 + qpart * len : f;
        jend = jmin + len < f ? jmin + len : f;
    }
    // ---- before the wait: y1 (forward launch) and the row numbers of x2 for the first chunk ----
    const int e1 = jmin + CHK < jend ? jmin + CHK : jend;
    int xrow[4];
#pragma unroll
    for (int k = 0; k < 4; k++) {
        const int j = jmin + tid + 256 * k;
        xrow[k] = -1;
        if (j < e1) {
            if (j < p) {
                const double dj = sym ? diag[fd.first + j] : 1.0;
#pragma unroll
                for (int c = 0; c < K; c++)
                    if (c < nk) wc[c * CHK + j - jmin] = sym ? W[c * wstr + j] / dj : W[c * wstr + j];
            } else {
                xrow[k] = rws[j - p];
            }
        }
    }
    if (STG && !sym) {
        // the first `stage` entries of this thread's share of its row of E' (positions jmin + g, jmin + g + G, ...), parked in LDS
        const double *Er = Ep + (i < r1 ? i : r1 - 1);
        for (int u0 = 0; u0 < stage; u0 += 8) {
            double t[8];
#pragma unroll
            for (int q = 0; q < 8; q++) {
                const int j = jmin + g + (u0 + q) * G;
                t[q] = Er[(int64_t)(j < f ? j : f - 1) * ld];
            }
#pragma unroll
            for (int q = 0; q < 8; q++) els[(u0 + q) * 256 + tid] = t[q];
        }
    }
    if constexpr (!TAG) {
        if (fd.parent >= 0 && tid == 0) sf_wait_front(fd.parent, need, done, STG ? rep_idx : nullptr, rep, err);
    }
    __syncthreads();
    if (trace && tid == 0) tr1 = dev_clock();
    // ---- after the wait ----
    if constexpr (TAG) {
        // all (up to four) words of x2 requested at once; the ones that still hold the tag are re-loaded until they are there
        double xv[4];
#pragma unroll
        for (int k = 0; k < 4; k++) xv[k] = ld_agent(xt + (xrow[k] >= 0 ? xrow[k] : 0));
#pragma unroll
        for (int k = 0; k < 4; k++)
            if (xrow[k] >= 0) wc[tid + 256 * k] = sf_tag_wait(xt + xrow[k], xv[k], err);
    } else if constexpr (K == 1) {
#pragma unroll
        for (int k = 0; k < 4; k++)
            if (xrow[k] >= 0) wc[tid + 256 * k] = ld_agent(x + xrow[k]);
    } else if constexpr (K <= 4) {
        // (all loads first, unconditional -- see ld_cols: up to four rows of x2 per thread and column)
        double xv[4][K];
#pragma unroll
        for (int k = 0; k < 4; k++) ld_cols<K>(xv[k], x, xstr, xrow[k] >= 0 ? xrow[k] : 0, nk);
#pragma unroll
        for (int k = 0; k < 4; k++)
            if (xrow[k] >= 0) {
#pragma unroll
                for (int c = 0; c < K; c++)
                    if (c < nk) wc[c * CHK + tid + 256 * k] = xv[k][c];
            }
    } else {
        // (the K columns of one row together; the wider blocks cannot hold 4 K values in registers)
#pragma unroll
        for (int k = 0; k < 4; k++) {
            if (jmin + 256 * k >= e1) break; // (workgroup-uniform: no thread has a row in this group)
            double xv[K];
            ld_cols<K>(xv, x, xstr, xrow[k] >= 0 ? xrow[k] : 0, nk);
            if (xrow[k] >= 0) {
#pragma unroll
                for (int c = 0; c < K; c++)
                    if (c < nk) wc[c * CHK + tid + 256 * k] = xv[c];
            }
        }
    }
    __syncthreads();
    if (trace && tid == 0) tr_g = dev_clock();
    // (K = 1: scalar dot products; K > 1: MFMA tiles, see sf_mma_chunk)
    double acc0[1] = {0.0}, acc1[1] = {0.0};
    double sacc[(SYM && K == 1) ? SF_SYMC : 1][1];
#pragma unroll
    for (int q = 0; q < ((SYM && K == 1) ? SF_SYMC : 1); q++) sacc[q][0] = 0.0;
    f64x4 macc[2] = {{0.0, 0.0, 0.0, 0.0}, {0.0, 0.0, 0.0, 0.0}};
    const int nsub = (1 << sh) >> 4; // 16-output tiles of the slab
    for (int c0 = jmin; c0 < jend; c0 += CHK) {
        const int c1 = c0 + CHK < jend ? c0 + CHK : jend;
        if (c0 > jmin) {
            for (int j = c0 + tid; j < c1; j += 256) {
                const int row = (j < p) ? 0 : rws[j - p];
                const double dj = (sym && j < p) ? diag[fd.first + j] : 1.0;
                double xv[K];
                if constexpr (TAG) xv[0] = (j < p) ? 0.0 : sf_tag_load(xt + row, err);
                else ld_cols<K>(xv, x, xstr, row, nk);
#pragma unroll
                for (int c = 0; c < K; c++)
                    if (c < nk) wc[c * CHK + j - c0] = (j < p) ? (sym ? W[c * wstr + j] / dj : W[c * wstr + j]) : xv[c];
            }
            __syncthreads();
        }
        if (K > 1) {
            // (sixteen loads in flight: the backward instances have the registers for it)
            if (sym) sf_mma_chunk<true, 16>(macc, Ep + (int64_t)r0 * fd.ld, fd.ld, wc, CHK, c0, c1, nk, nsub, r1 - r0, wave, lane);
            else sf_mma_chunk<false, 16>(macc, Ep + r0, ld, wc, CHK, c0, c1, nk, nsub, r1 - r0, wave, lane);
        } else if (sym) {
            // transposed GEMV: wave w owns the columns r0 + w, r0 + w + 4, ... of E (<= SF_SYMC of them: slabs of 16 rows),
            // lanes run down the column (contiguous), two positions of every column in flight per lane
            const double *wk = wc - c0;
            int j = c0 + lane;
            for (; j + 64 < c1; j += 128) {
                double e[SF_SYMC][2];
#pragma unroll
                for (int q = 0; q < SF_SYMC; q++) {
                    const int col = r0 + wave + 4 * q;
                    const double *Ec = Ep + (int64_t)(col < r1 ? col : r0) * fd.ld;
                    e[q][0] = Ec[j], e[q][1] = Ec[j + 64];
                }
#pragma unroll
                for (int q = 0; q < SF_SYMC; q++) {
                    sacc[(SYM && K == 1) ? q : 0][0] += e[q][0] * wk[j];
                    sacc[(SYM && K == 1) ? q : 0][0] += e[q][1] * wk[j + 64];
                }
            }
            if (j < c1) {
                double e[SF_SYMC];
#pragma unroll
                for (int q = 0; q < SF_SYMC; q++) {
                    const int col = r0 + wave + 4 * q;
                    e[q] = Ep[(int64_t)(col < r1 ? col : r0) * fd.ld + j];
                }
#pragma unroll
                for (int q = 0; q < SF_SYMC; q++) sacc[(SYM && K == 1) ? q : 0][0] += e[q] * wk[j];
            }
        } else if (STG && c0 == jmin) {
            if (i < r1) { // (same order of additions as sf_dot: see k_fwd_fused)
                const int n0 = c0 + g < c1 ? (c1 - c0 - g + G - 1) / G : 0, nfull = n0 / 8 * 8, nst = stage < n0 ? stage : n0;
                int sq = 0;
                for (; sq + 8 <= nst && sq + 8 <= nfull; sq += 8) {
                    double e[8];
#pragma unroll
                    for (int q = 0; q < 8; q++) e[q] = els[(sq + q) * 256 + tid];
#pragma unroll
                    for (int q = 0; q < 8; q += 2) {
                        acc0[0] += e[q] * wc[g + (sq + q) * G];
                        acc1[0] += e[q + 1] * wc[g + (sq + q + 1) * G];
                    }
                }
                if (sq + 8 <= nfull) sf_dot<1>(acc0, acc1, Ep + i, ld, wc, CHK, c0, c0 + g + sq * G, c1, G);
                else
                    for (; sq < n0; sq++) {
                        const double ev = sq < stage ? els[sq * 256 + tid] : Ep[i + (int64_t)(c0 + g + sq * G) * ld];
                        acc0[0] += ev * wc[g + sq * G];
                    }
            }
        } else if (i < r1)
            sf_dot<1>(acc0, acc1, Ep + i, ld, wc, CHK, c0, c0 + g, c1, G);
        __syncthreads();
    }
    if (trace && tid == 0) tr_d = dev_clock();
    bool finish = true; // (workgroup-uniform) this task completes the slab
    if (K > 1) {
        sf_mma_store(macc, mt, nsub, wave, lane);
        __syncthreads();
        if (Q > 1) {
            // part qpart of the slab: nsub units of 16 rows x 16 columns
            double *mine = split_scr + ((int64_t)t.d + (int64_t)qpart * nsub) * 256;
            if (tid < (1 << sh)) {
#pragma unroll
                for (int c = 0; c < K; c++) st_agent(mine + tid * 16 + c, sf_mma_sum(mt, nsub, tid, c));
            }
            drain_stores();
            __syncthreads();
            if (tid == 0) red[0] = (double)flag_add(split_cnt + t.d, 1);
            __syncthreads();
            finish = (int)red[0] == Q - 1;
            if (finish && tid == 0) flag_store(split_cnt + t.d, 0);
        }
        if (finish && tid < (1 << sh) && r0 + tid < r1) {
#pragma unroll
            for (int c = 0; c < K; c++)
                if (c < nk) {
                    double tot;
                    if (Q > 1) {
                        tot = 0.0;
                        for (int qq = 0; qq < Q; qq++) tot += ld_agent(split_scr + ((int64_t)t.d + (int64_t)qq * nsub) * 256 + tid * 16 + c);
                    } else
                        tot = sf_mma_sum(mt, nsub, tid, c);
                    st_agent(x + c * xstr + fd.first + r0 + tid, tot);
                }
        }
    } else if (sym) {
        // the 64 partial sums of a column are added in a fixed (butterfly) order
#pragma unroll
        for (int q = 0; q < SF_SYMC; q++) {
            const int col = r0 + wave + 4 * q;
            double v = sacc[(SYM && K == 1) ? q : 0][0];
            for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off);
            if (lane == 0 && col < r1) {
                st_agent(x + fd.first + col, v);
                if constexpr (TAG) st_agent(xt + fd.first + col, v);
            }
        }
    } else {
        red[g * (1 << sh) + rr] = acc0[0] + acc1[0];
        __syncthreads();
        if (g == 0 && i < r1) {
            const double xi = sf_group_sum(red, 1 << sh, rr, G);
            st_agent(x + fd.first + i, xi);
            if constexpr (TAG) st_agent(xt + fd.first + i, xi);
        }
    }
    if (trace && tid == 0) tr2 = dev_clock();
    if constexpr (!TAG) {
        if (finish) { // (a part that is not the last one of its slab has nothing to publish)
            drain_stores();
            __syncthreads();
            if (tid == 0) sf_publish_front(t.a, need, done, STG ? rep_idx : nullptr, rep);
        }
    }
    if (trace && tid == 0) {
        unsigned long long *tr = trace + 8 * (size_t)bid;
        tr[0] = tr0, tr[1] = tr1, tr[2] = tr2, tr[3] = dev_clock(), tr[4] = tr_g, tr[5] = tr_d;
    }
}

} // namespace hipmf
