// kernels_vector.hpp -- vector kernels of the solve driver and of iterative refinement (HBM-bound).
#pragma once
#include "kernels_common.hpp"

namespace hipmf {

// xp[i] = rs[rperm[i]] * b[rperm[i]]   (rperm: the row of A that is row i of the permuted system)
__global__ void k_perm_in(int32_t n, const int32_t *__restrict__ perm, const double *__restrict__ rs,
                          const double *__restrict__ b, double *__restrict__ xp) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) xp[i] = rs[perm[i]] * b[perm[i]];
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

// r = b - A x and, fused, the two norms the refinement needs: nrm[0] = max_i |r_i|, nrm[1] = omega = max_i |r_i| / den_i
// with den_i = (|A| |x| + |b|)_i (ordered bits of non-negative doubles, atomicMax; nrm is zeroed before the launch; see RES_SLOTS).
// CSR; for symmetric-lower storage the mirrored entries come from tptr/tidx/arow.  omega, the componentwise backward
// error, decides as in UMFPACK's / LAPACK's refinement whether another step can still help.
// RES_LANES = 8 lanes share a row (stencil matrices have 5-7 entries per row): consecutive lanes read consecutive entries of
// vals / ci, i.e. the 12 B per entry stream in whole cache lines (one thread per row reads them with a stride of a row).
constexpr int RES_LANES = 8;
// The maxima of the workgroups are combined with atomicMax on RES_SLOTS separate cache lines (workgroup b uses slot b mod
// RES_SLOTS; one word takes only ~90 atomics per microsecond, which would bound a launch of 31 000 workgroups); the host takes
// the maximum over the slots.  Layout per column: slot s at nrm[s * RES_SLOT_WORDS] (|r|) and nrm[s * RES_SLOT_WORDS + 1] (omega).
constexpr int RES_SLOTS = 64, RES_SLOT_WORDS = 16, RES_NORM_WORDS = RES_SLOTS * RES_SLOT_WORDS;
__global__ void __launch_bounds__(256) k_residual(int32_t n, const int32_t *__restrict__ rp, const int32_t *__restrict__ ci,
                                                  const double *__restrict__ vals, const int32_t *__restrict__ tptr,
                                                  const int32_t *__restrict__ tidx, const int32_t *__restrict__ arow,
                                                  const double *__restrict__ x, const double *__restrict__ b, double *__restrict__ r,
                                                  unsigned long long *nrm) {
    __shared__ double red[256 / RES_LANES], red2[256 / RES_LANES];
    const int sub = threadIdx.x & (RES_LANES - 1), grp = threadIdx.x / RES_LANES;
    double a = 0.0, q = 0.0;
    {
        const int i = blockIdx.x * (256 / RES_LANES) + grp;
        double acc = 0.0, d = 0.0;
        if (i < n) {
            for (int p = rp[i] + sub; p < rp[i + 1]; p += RES_LANES) {
                double t = vals[p] * x[ci[p]];
                acc -= t;
                d += fabs(t);
            }
            if (tptr)
                for (int k = tptr[i] + sub; k < tptr[i + 1]; k += RES_LANES) {
                    double t = vals[tidx[k]] * x[arow[tidx[k]]];
                    acc -= t;
                    d += fabs(t);
                }
        }
        // fixed-order tree over the lanes of a row (the same order in every run: the refinement stays deterministic)
#pragma unroll
        for (int o = RES_LANES / 2; o > 0; o >>= 1) {
            acc += __shfl_xor(acc, o);
            d += __shfl_xor(d, o);
        }
        if (i < n && sub == 0) {
            acc += b[i];
            d += fabs(b[i]);
            r[i] = acc;
            const double ai = fabs(acc);
            const double qi = (d > 0.0) ? ai / d : (ai > 0.0 ? 1.0 : 0.0);
            if (ai > a) a = ai;
            if (qi > q) q = qi;
        }
    }
    // a NaN never wins a maximum: a NaN residual ends the refinement through the "no progress" test
    if (sub == 0) {
        red[grp] = a > 0.0 ? a : 0.0;
        red2[grp] = q > 0.0 ? q : 0.0;
    }
    __syncthreads();
    for (int s = 256 / RES_LANES / 2; s > 0; s >>= 1) {
        if ((int)threadIdx.x < s) {
            if (red[threadIdx.x + s] > red[threadIdx.x]) red[threadIdx.x] = red[threadIdx.x + s];
            if (red2[threadIdx.x + s] > red2[threadIdx.x]) red2[threadIdx.x] = red2[threadIdx.x + s];
        }
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        unsigned long long *slot = nrm + (size_t)(blockIdx.x & (RES_SLOTS - 1)) * RES_SLOT_WORDS;
        atomicMax(slot, (unsigned long long)__double_as_longlong(red[0]));
        atomicMax(slot + 1, (unsigned long long)__double_as_longlong(red2[0]));
    }
}

// y = alpha * A x  (CSR SpMV, the mat_vec_mul of csr_matrix.rs:709-729), one thread per row
__global__ void k_spmv(int32_t n, const int32_t *__restrict__ rp, const int32_t *__restrict__ ci,
                       const double *__restrict__ vals, const int32_t *__restrict__ tptr, const int32_t *__restrict__ tidx,
                       const int32_t *__restrict__ arow, double alpha, const double *__restrict__ x, double *__restrict__ y) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    double acc = 0.0;
    for (int p = rp[i]; p < rp[i + 1]; p++) acc += vals[p] * x[ci[p]];
    if (tptr)
        for (int q = tptr[i]; q < tptr[i + 1]; q++) acc += vals[tidx[q]] * x[arow[tidx[q]]];
    y[i] = alpha * acc;
}

} // namespace hipmf
