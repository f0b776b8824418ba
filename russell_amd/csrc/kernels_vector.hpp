// kernels_vector.hpp -- vector kernels of the solve driver and of iterative refinement (HBM-bound).
#pragma once
#include "kernels_common.hpp"

namespace hipmf {

// xp[i] = rs[rperm[i]] * b[rperm[i]]   (rperm: the row of A that is row i of the permuted system)
__global__ void k_perm_in(int32_t n, const int32_t *__restrict__ perm, const double *__restrict__ rs,
                          const double *__restrict__ b, double *__restrict__ xp) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) {
        double v = rs[perm[i]] * b[perm[i]];
        // (a NaN of the right-hand side enters the solves as THE quiet NaN: the data-tagged hand-offs of kernels_solve_fused.hpp use
        //  another NaN pattern as "not yet written", and NaN payloads propagate through the arithmetic)
        if (v != v) v = __longlong_as_double(0x7ff8000000000000LL);
        xp[i] = v;
    }
}

// out[perm[j]] = cs[perm[j]] * xp[j] (mode 0), += (mode 1), -= (mode 2); cs == nullptr: no column scaling
__global__ void k_perm_out(int32_t n, const int32_t *__restrict__ perm, const double *__restrict__ cs, const double *__restrict__ xp,
                           double *__restrict__ out, int32_t mode) {
    int j = blockIdx.x * blockDim.x + threadIdx.x;
    if (j < n) {
        const int q = perm[j];
        const double v = cs ? cs[q] * xp[j] : xp[j];
        if (mode == 1) out[q] += v;
        else if (mode == 2) out[q] -= v;
        else out[q] = v;
    }
}

// The same for a block of columns in ONE launch (blockIdx.y = column; column c of b / xp / out at base + c * stride): a block of 16
// right-hand sides used to cost 32 launches of 8 - 14 us around every pass.  Columns whose bit in `mask` is clear get zeros
// (k_perm_in_cols: finished columns of a refinement step ride along as zeros) or are left alone (k_perm_out_cols).
// Round 6: a thread carries PERM_CW columns of its row -- the permutation entry and the scaling factor are fetched once per row and
// chunk of columns instead of once per column (they were 12 of the 28 bytes a column entry moved: 64 columns of the 1M-DOF system
// 542 + 486 us per pass, profiles/r06_many_rhs_kernel_stats.txt), and the PERM_CW gathers of a thread are in flight together.
// blockIdx.y = chunk of PERM_CW columns.
constexpr int PERM_CW = 8;
__global__ void __launch_bounds__(256) k_perm_in_cols(int32_t n, const int32_t *__restrict__ perm, const double *__restrict__ rs, const double *__restrict__ b,
                                                      int64_t bstr, double *__restrict__ xp, int64_t xstr, uint64_t mask, int32_t ncols) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x, c0 = blockIdx.y * PERM_CW;
    if (i >= n) return;
    const uint32_t m = (uint32_t)((mask >> c0) & ((1u << PERM_CW) - 1u));
    const int q = perm[i];
    const double r = rs[q];
    double v[PERM_CW];
#pragma unroll
    for (int k = 0; k < PERM_CW; k++) v[k] = b[q + (int64_t)(c0 + k < ncols ? c0 + k : c0) * bstr]; // (clamped column: unconditional loads)
#pragma unroll
    for (int k = 0; k < PERM_CW; k++)
        if (c0 + k < ncols) xp[i + (int64_t)(c0 + k) * xstr] = ((m >> k) & 1u) ? r * v[k] : 0.0;
}
__global__ void __launch_bounds__(256) k_perm_out_cols(int32_t n, const int32_t *__restrict__ perm, const double *__restrict__ cs, const double *__restrict__ xp,
                                                       int64_t xstr, double *__restrict__ out, int64_t ostr, int32_t mode, uint64_t mask, int32_t ncols) {
    const int j = blockIdx.x * blockDim.x + threadIdx.x, c0 = blockIdx.y * PERM_CW;
    if (j >= n) return;
    const uint32_t m = (uint32_t)((mask >> c0) & ((1u << PERM_CW) - 1u));
    if (m == 0) return;
    const int q = perm[j];
    const double sc = cs ? cs[q] : 1.0;
    double v[PERM_CW], o[PERM_CW];
#pragma unroll
    for (int k = 0; k < PERM_CW; k++) v[k] = xp[j + (int64_t)(c0 + k < ncols ? c0 + k : c0) * xstr];
    if (mode != 0) {
#pragma unroll
        for (int k = 0; k < PERM_CW; k++) o[k] = out[q + (int64_t)(c0 + k < ncols ? c0 + k : c0) * ostr];
    }
#pragma unroll
    for (int k = 0; k < PERM_CW; k++)
        if (c0 + k < ncols && ((m >> k) & 1u)) {
            const double t = cs ? sc * v[k] : v[k];
            double *dst = out + q + (int64_t)(c0 + k) * ostr;
            if (mode == 1) *dst = o[k] + t;
            else if (mode == 2) *dst = o[k] - t;
            else *dst = t;
        }
}

// CSR SpMV in "stream" form, shared by y = alpha A x (mat_vec_mul, csr_matrix.rs:709-729) and by the residual of the iterative
// refinement.  The rows are cut at initialize into blocks of at most SPMV_CAP stored entries (row_blk, host); a workgroup
//   1. streams its block's entries: lanes read CONSECUTIVE entries of vals / ci (12 B per entry in whole cache lines, four loads
//      in flight per lane, no dependence on the row pointers), gather x and park the products in LDS;
//   2. reduces the rows from LDS: L lanes per row (L = the power of two that fills the workgroup, 1 for stencil matrices, 8-16 for
//      FE matrices with ~40 entries per row), entries in order, lanes combined by a fixed butterfly -- the sums are reproducible;
//      the mirrored entries of symmetric-lower storage (tptr / tidx / arow lists) are added here by the row's first lane.
// A row longer than SPMV_CAP is a block of its own and is summed in strides by the whole workgroup.
// RESID: r = b - A x and, fused, the two norms the refinement needs: nrm[0] = max_i |r_i|, nrm[1] = omega = max_i |r_i| / den_i
// with den_i = (|A| |x| + |b|)_i (ordered bits of non-negative doubles, atomicMax; nrm is zeroed before the launch; see RES_SLOTS).
// omega, the componentwise backward error, decides as in UMFPACK's / LAPACK's refinement whether another step can still help.
// The maxima of the workgroups are combined with atomicMax on RES_SLOTS separate cache lines (workgroup b uses slot b mod
// RES_SLOTS; one word takes only ~90 atomics per microsecond); the host takes the maximum over the slots.
// Layout per column: slot s at nrm[s * RES_SLOT_WORDS] (|r|) and nrm[s * RES_SLOT_WORDS + 1] (omega).
constexpr int SPMV_CAP = 1024;
constexpr int RES_SLOTS = 64, RES_SLOT_WORDS = 16, RES_NORM_WORDS = RES_SLOTS * RES_SLOT_WORDS;
struct SpmvLds {
    double prod[SPMV_CAP];
    double aprod[SPMV_CAP];
    double red[256], red2[256];
};
// one column: the workgroup's row block against x (the entries of the block stay in L1 / L2 between the columns of a block call)
template <bool RESID>
__device__ __forceinline__ void spmv_block(SpmvLds &sh, const int32_t *__restrict__ row_blk, const int32_t *__restrict__ rp,
                                           const int32_t *__restrict__ ci, const double *__restrict__ vals,
                                           const int32_t *__restrict__ tptr, const int32_t *__restrict__ tidx,
                                           const int32_t *__restrict__ arow, double alpha, const double *__restrict__ x,
                                           const double *__restrict__ b, double *__restrict__ y, unsigned long long *nrm) {
    double(&prod)[SPMV_CAP] = sh.prod;
    double(&aprod)[SPMV_CAP] = sh.aprod;
    double(&red)[256] = sh.red;
    double(&red2)[256] = sh.red2;
    const int tid = threadIdx.x;
    const int r0 = row_blk[blockIdx.x], r1 = row_blk[blockIdx.x + 1];
    const int e0 = rp[r0], e1 = rp[r1];
    double a_max = 0.0, q_max = 0.0;
    if (e1 - e0 > SPMV_CAP) {
        // one long row (r1 == r0 + 1): strided partial sums, fixed-order tree over the workgroup
        double acc = 0.0, d = 0.0;
        for (int e = e0 + tid; e < e1; e += 256) {
            const double t = vals[e] * x[ci[e]];
            acc += t;
            d += fabs(t);
        }
        if (tptr)
            for (int q = tptr[r0] + tid; q < tptr[r0 + 1]; q += 256) {
                const double t = vals[tidx[q]] * x[arow[tidx[q]]];
                acc += t;
                d += fabs(t);
            }
        red[tid] = acc;
        if (RESID) red2[tid] = d;
        __syncthreads();
        for (int s = 128; s > 0; s >>= 1) {
            if (tid < s) {
                red[tid] += red[tid + s];
                if (RESID) red2[tid] += red2[tid + s];
            }
            __syncthreads();
        }
        if (tid == 0) {
            if (RESID) {
                const double ri = b[r0] - red[0], di = red2[0] + fabs(b[r0]);
                y[r0] = ri;
                a_max = fabs(ri);
                q_max = di > 0.0 ? a_max / di : (a_max > 0.0 ? 1.0 : 0.0);
            } else {
                y[r0] = alpha * red[0];
            }
        }
    } else {
        const int ne = e1 - e0;
        for (int k = tid; k < ne; k += 4 * 256) {
            double v[4];
            int c[4];
#pragma unroll
            for (int u = 0; u < 4; u++) {
                const int kk = k + 256 * u;
                v[u] = kk < ne ? vals[e0 + kk] : 0.0;
                c[u] = kk < ne ? ci[e0 + kk] : 0;
            }
            double xv[4];
#pragma unroll
            for (int u = 0; u < 4; u++) xv[u] = (k + 256 * u < ne) ? x[c[u]] : 0.0;
#pragma unroll
            for (int u = 0; u < 4; u++) {
                const int kk = k + 256 * u;
                if (kk < ne) {
                    const double t = v[u] * xv[u];
                    prod[kk] = t;
                    if (RESID) aprod[kk] = fabs(t);
                }
            }
        }
        __syncthreads();
        const int nr = r1 - r0;
        int lsh = 0; // log2(lanes per row)
        while (lsh < 6 && (nr << (lsh + 1)) <= 256) lsh++;
        const int L = 1 << lsh, sub = tid & (L - 1);
        for (int rr = tid >> lsh; rr < ((nr + (256 >> lsh) - 1) / (256 >> lsh)) * (256 >> lsh); rr += 256 >> lsh) {
            const int i = r0 + rr;
            double acc = 0.0, d = 0.0;
            if (rr < nr) {
                for (int e = rp[i] - e0 + sub; e < rp[i + 1] - e0; e += L) {
                    acc += prod[e];
                    if (RESID) d += aprod[e];
                }
                if (tptr && sub == 0)
                    for (int q = tptr[i]; q < tptr[i + 1]; q++) {
                        const double t = vals[tidx[q]] * x[arow[tidx[q]]];
                        acc += t;
                        d += fabs(t);
                    }
            }
            for (int o = L >> 1; o > 0; o >>= 1) { // (every lane of the wavefront takes part: the trip count above is uniform)
                acc += __shfl_xor(acc, o);
                if (RESID) d += __shfl_xor(d, o);
            }
            if (rr < nr && sub == 0) {
                if (RESID) {
                    const double ri = b[i] - acc, di = d + fabs(b[i]);
                    y[i] = ri;
                    const double ai = fabs(ri);
                    const double qi = di > 0.0 ? ai / di : (ai > 0.0 ? 1.0 : 0.0);
                    if (ai > a_max) a_max = ai;
                    if (qi > q_max) q_max = qi;
                } else {
                    y[i] = alpha * acc;
                }
            }
        }
    }
    if (RESID) {
        // a NaN never wins a maximum: a NaN residual ends the refinement through the "no progress" test
        __syncthreads();
        red[tid] = a_max > 0.0 ? a_max : 0.0;
        red2[tid] = q_max > 0.0 ? q_max : 0.0;
        __syncthreads();
        for (int s = 128; s > 0; s >>= 1) {
            if (tid < s) {
                if (red[tid + s] > red[tid]) red[tid] = red[tid + s];
                if (red2[tid + s] > red2[tid]) red2[tid] = red2[tid + s];
            }
            __syncthreads();
        }
        if (tid == 0) {
            unsigned long long *slot = nrm + (size_t)(blockIdx.x & (RES_SLOTS - 1)) * RES_SLOT_WORDS;
            atomicMax(slot, (unsigned long long)__double_as_longlong(red[0]));
            atomicMax(slot + 1, (unsigned long long)__double_as_longlong(red2[0]));
        }
    }
}

template <bool RESID>
__global__ void __launch_bounds__(256) k_spmv_stream(const int32_t *__restrict__ row_blk, const int32_t *__restrict__ rp,
                                                     const int32_t *__restrict__ ci, const double *__restrict__ vals,
                                                     const int32_t *__restrict__ tptr, const int32_t *__restrict__ tidx,
                                                     const int32_t *__restrict__ arow, double alpha, const double *__restrict__ x,
                                                     const double *__restrict__ b, double *__restrict__ y, unsigned long long *nrm) {
    __shared__ SpmvLds sh;
    spmv_block<RESID>(sh, row_blk, rp, ci, vals, tptr, tidx, arow, alpha, x, b, y, nrm);
}

// Residuals of a block of columns in one launch: r_c = b_c - A x_c for the columns whose bit in `mask` is set (column c of x / b at
// base + c * stride, of r at r + c * rstr, norms at nrm + c * RES_NORM_WORDS).  The workgroup keeps its row block and walks the columns:
// vals / ci come from HBM once per block call instead of once per column (16 columns: 16 launches of 28 us before).
__global__ void __launch_bounds__(256) k_residual_cols(const int32_t *__restrict__ row_blk, const int32_t *__restrict__ rp,
                                                       const int32_t *__restrict__ ci, const double *__restrict__ vals,
                                                       const int32_t *__restrict__ tptr, const int32_t *__restrict__ tidx,
                                                       const int32_t *__restrict__ arow, const double *__restrict__ x, int64_t xstr,
                                                       const double *__restrict__ b, int64_t bstr, double *__restrict__ r, int64_t rstr,
                                                       unsigned long long *nrm, int32_t ncols, uint64_t mask) {
    __shared__ SpmvLds sh;
    for (int c = 0; c < ncols; c++) {
        if (!((mask >> c) & 1ull)) continue;
        spmv_block<true>(sh, row_blk, rp, ci, vals, tptr, tidx, arow, 1.0, x + c * xstr, b + c * bstr, r + c * rstr, nrm + (size_t)c * RES_NORM_WORDS);
        __syncthreads();
    }
}

// min and max of |d_i| over the pivots (the cheap reciprocal-condition estimate min |u_ii| / max |u_ii|, UMFPACK_RCOND's definition):
// out[0] = bits of the minimum, out[1] = bits of the maximum (non-negative doubles order like their bit patterns);
// the caller sets out[0] = +inf bits, out[1] = 0
__global__ void __launch_bounds__(256) k_diag_minmax(int32_t n, const double *__restrict__ d, unsigned long long *out) {
    __shared__ double smin[256], smax[256];
    double mn = INFINITY, mx = 0.0;
    for (int i = blockIdx.x * 256 + threadIdx.x; i < n; i += gridDim.x * 256) {
        const double a = fabs(d[i]);
        mn = a < mn ? a : mn;
        mx = a > mx ? a : mx;
    }
    smin[threadIdx.x] = mn, smax[threadIdx.x] = mx;
    __syncthreads();
    for (int s = 128; s > 0; s >>= 1) {
        if ((int)threadIdx.x < s) {
            if (smin[threadIdx.x + s] < smin[threadIdx.x]) smin[threadIdx.x] = smin[threadIdx.x + s];
            if (smax[threadIdx.x + s] > smax[threadIdx.x]) smax[threadIdx.x] = smax[threadIdx.x + s];
        }
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        atomicMin(out, (unsigned long long)__double_as_longlong(smin[0]));
        atomicMax(out + 1, (unsigned long long)__double_as_longlong(smax[0]));
    }
}

} // namespace hipmf
