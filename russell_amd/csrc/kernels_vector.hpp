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

// r = b - A x and den_i = (|A| |x| + |b|)_i  (CSR; for symmetric-lower storage the mirrored entries come
// from tptr/tidx/arow).  den feeds the componentwise backward error omega = max_i |r_i| / den_i that
// decides, as in UMFPACK's / LAPACK's refinement, whether another step can still help.
__global__ void k_residual(int32_t n, const int32_t *__restrict__ rp, const int32_t *__restrict__ ci,
                           const double *__restrict__ vals, const int32_t *__restrict__ tptr, const int32_t *__restrict__ tidx,
                           const int32_t *__restrict__ arow, const double *__restrict__ x, const double *__restrict__ b,
                           double *__restrict__ r, double *__restrict__ den) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    double acc = b[i], d = fabs(b[i]);
    for (int p = rp[i]; p < rp[i + 1]; p++) {
        double t = vals[p] * x[ci[p]];
        acc -= t;
        d += fabs(t);
    }
    if (tptr)
        for (int q = tptr[i]; q < tptr[i + 1]; q++) {
            double t = vals[tidx[q]] * x[arow[tidx[q]]];
            acc -= t;
            d += fabs(t);
        }
    r[i] = acc;
    den[i] = d;
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

// out[0] = max_i |v_i| ; out[1] = max_i |v_i| / den_i  (ordered bits; den may be NULL)
__global__ void k_norms(int32_t n, const double *__restrict__ v, const double *__restrict__ den, unsigned long long *out) {
    __shared__ double red[256], red2[256];
    double m = 0.0, w = 0.0;
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
        double a = fabs(v[i]);
        m = a > m ? a : m; // a NaN never wins: a NaN residual ends the refinement through the "no progress" test
        if (den) {
            double q = (den[i] > 0.0) ? a / den[i] : (a > 0.0 ? 1.0 : 0.0);
            w = q > w ? q : w;
        }
    }
    red[threadIdx.x] = m;
    red2[threadIdx.x] = w;
    __syncthreads();
    for (int s = blockDim.x / 2; s > 0; s >>= 1) {
        if ((int)threadIdx.x < s) {
            if (red[threadIdx.x + s] > red[threadIdx.x]) red[threadIdx.x] = red[threadIdx.x + s];
            if (red2[threadIdx.x + s] > red2[threadIdx.x]) red2[threadIdx.x] = red2[threadIdx.x + s];
        }
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        atomicMax(out, (unsigned long long)__double_as_longlong(red[0]));
        atomicMax(out + 1, (unsigned long long)__double_as_longlong(red2[0]));
    }
}

// diagonal of U in pivot order (for the determinant / rcond estimate)
__global__ void k_diag_gather(int32_t nsuper, const FrontDesc *__restrict__ FD, const double *__restrict__ pool,
                              double *__restrict__ du) {
    int s = blockIdx.x;
    if (s >= nsuper) return;
    FrontDesc fd = FD[s];
    const int64_t ld = fd.ld;
    for (int i = threadIdx.x; i < fd.p; i += blockDim.x) du[fd.first + i] = pool[fd.off + i + i * ld];
}

} // namespace hipmf
