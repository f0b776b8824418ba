// kernels.hpp -- hand-written HIP kernels (gfx950 / CDNA4) of the multifrontal LU backend.
//
// Data layout in HBM
//   pool      every supernode s owns an f x f column-major frontal matrix (f = p + m, p pivot
//             columns, m off-diagonal rows) at pool + off[s].  After factorisation the first p
//             columns hold L11\U11 and L21 (one contiguous f x p panel), rows 0..p of the
//             remaining columns hold U12, and the trailing m x m block is the contribution block
//             that the parent front consumes (extend-add).
//   lperm     n int32: for pivot row r of front s, the front-local row that was moved there by
//             partial pivoting (restricted to the pivot block / 32-row diagonal tile).
//   work      one f-vector per front for the multifrontal forward/backward substitutions.
//
// Kernel families (the roofline that bounds each is stated in DESIGN.md):
//   k_row_scale, k_absmax, k_scatter, k_extend_add          assembly           HBM-bound
//   k_small_factor                                           fronts f <= 64    LDS / latency-bound
//   k_diag, k_panel, k_update (v_mfma_f64_16x16x4_f64)       tiled big fronts   MFMA / HBM
//   k_fwd, k_bwd                                             level-set SpTRSV   HBM-bound
//   k_residual (CSR SpMV), k_perm_in/out, k_axpy, k_norminf  refinement         HBM-bound
#pragma once
#include <hipmf_device_rt.h>

#include <cstdint>

namespace hipmf {

constexpr int NB = 32;       // pivot-block width of the tiled path
constexpr int PANEL_T = 128; // rows (L) / columns (U) handled by one panel workgroup
constexpr int UPD_T = 64;    // trailing-update tile edge (one 256-thread workgroup, 4 waves of 32x32)
constexpr int SMALL_F = 64;  // fronts with f <= SMALL_F are factorised by one wavefront in LDS
constexpr int LS_LD = 80;    // LDS leading dimensions of the update kernel (bank-conflict free, see k_update)
constexpr int US_LD = 34;

struct FrontDesc {
    int64_t off;    // offset of the front in the pool (doubles)
    int64_t rowptr; // offset of the row structure / relative indices
    int64_t woff;   // offset of the f-vector in the solve workspace
    int32_t p, m;   // pivots, off-diagonal rows
    int32_t first;  // first permuted column
    int32_t child_begin, child_end;
    int32_t parent;
};

struct EaTask {
    int32_t s, c0, c1; // parent front, parent-column range [c0, c1)
};

// device-side counters written by the factorisation kernels
struct FactorInfo {
    int32_t n_perturbed;  // pivots replaced by +-eps (cf. CUDSS_DATA_NPIVOTS, interface_cudss.cu:466-475)
    int32_t n_zero_pivot; // exactly-zero pivots met (singular in the UMFPACK sense, solver_umfpack.rs:492)
    int32_t pad0, pad1;
};

__device__ __forceinline__ int lower_bound_i32(const int32_t *a, int n, int v) {
    int lo = 0, hi = n;
    while (lo < hi) {
        int mid = (lo + hi) >> 1;
        if (a[mid] < v) lo = mid + 1;
        else hi = mid;
    }
    return lo;
}

// upper_bound on a prefix array: largest a with pfx[a] <= g
__device__ __forceinline__ int find_slot(const int32_t *pfx, int n, int g) {
    int lo = 0, hi = n;
    while (hi - lo > 1) {
        int mid = (lo + hi) >> 1;
        if (pfx[mid] <= g) lo = mid;
        else hi = mid;
    }
    return lo;
}

// ------------------------------------------------------------------------------------------------
// Assembly
// ------------------------------------------------------------------------------------------------

// rs[i] = 1 / sum_j |a_ij| (mode 1, UMFPACK_SCALE_SUM), 1 / max_j |a_ij| (mode 2), 1 (mode 0).
// tptr/tidx list, for every row i, the positions of the stored entries (r, i), r != i, that the
// symmetric-lower storage mirrors into row i (empty for general storage).
__global__ void k_row_scale(int32_t n, const int32_t *__restrict__ rp, const double *__restrict__ vals,
                            const int32_t *__restrict__ tptr, const int32_t *__restrict__ tidx, int32_t mode,
                            double *__restrict__ rs) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    double acc = 0.0;
    if (mode != 0) {
        for (int p = rp[i]; p < rp[i + 1]; p++) {
            double a = fabs(vals[p]);
            acc = (mode == 1) ? acc + a : (a > acc ? a : acc);
        }
        if (tptr)
            for (int q = tptr[i]; q < tptr[i + 1]; q++) {
                double a = fabs(vals[tidx[q]]);
                acc = (mode == 1) ? acc + a : (a > acc ? a : acc);
            }
    }
    rs[i] = (mode == 0 || acc == 0.0) ? 1.0 : 1.0 / acc;
}

// max |rs[row] * a| over the stored entries -> *out (as ordered bits of a non-negative double)
__global__ void k_absmax(int64_t nnz, const double *__restrict__ vals, const int32_t *__restrict__ arow,
                         const double *__restrict__ rs, unsigned long long *out) {
    __shared__ double red[256];
    double m = 0.0;
    for (int64_t k = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; k < nnz; k += (int64_t)gridDim.x * blockDim.x) {
        double a = fabs(vals[k] * rs[arow[k]]);
        m = a > m ? a : m;
    }
    red[threadIdx.x] = m;
    __syncthreads();
    for (int s = blockDim.x / 2; s > 0; s >>= 1) {
        if ((int)threadIdx.x < s && red[threadIdx.x + s] > red[threadIdx.x]) red[threadIdx.x] = red[threadIdx.x + s];
        __syncthreads();
    }
    if (threadIdx.x == 0) atomicMax(out, (unsigned long long)__double_as_longlong(red[0]));
}

// pool[amap[k]] = rs[row(k)] * a_k  (the pool is zero-filled first; every position is hit once)
__global__ void k_scatter(int64_t nnz, const double *__restrict__ vals, const int32_t *__restrict__ arow,
                          const int64_t *__restrict__ amap, const int64_t *__restrict__ amap2,
                          const double *__restrict__ rs, const int32_t *__restrict__ acol, double *__restrict__ pool) {
    for (int64_t k = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; k < nnz; k += (int64_t)gridDim.x * blockDim.x) {
        double v = vals[k];
        pool[amap[k]] = v * rs[arow[k]];
        if (amap2) {
            int64_t q = amap2[k];
            if (q >= 0) pool[q] = v * rs[acol[k]]; // mirrored entry lives in row acol[k]
        }
    }
}

// extend-add: every task adds the children's contribution blocks into a column range of the
// parent front.  Children are visited in ascending order and a parent column belongs to exactly
// one task, so the floating-point summation order is fixed (bit-reproducible factors).
__global__ void k_extend_add(const EaTask *__restrict__ tasks, const FrontDesc *__restrict__ FD,
                             const int32_t *__restrict__ child_idx, const int32_t *__restrict__ rel,
                             double *__restrict__ pool) {
    EaTask t = tasks[blockIdx.x];
    FrontDesc fd = FD[t.s];
    const int64_t f = (int64_t)fd.p + fd.m;
    double *F = pool + fd.off;
    for (int ci = fd.child_begin; ci < fd.child_end; ci++) {
        FrontDesc cd = FD[child_idx[ci]];
        const int mc = cd.m;
        if (mc == 0) continue;
        const int64_t fc = (int64_t)cd.p + cd.m;
        const double *CB = pool + cd.off + cd.p + (int64_t)cd.p * fc;
        const int32_t *relc = rel + cd.rowptr;
        int jlo = lower_bound_i32(relc, mc, t.c0), jhi = lower_bound_i32(relc, mc, t.c1);
        int64_t total = (int64_t)(jhi - jlo) * mc;
        for (int64_t e = threadIdx.x; e < total; e += blockDim.x) {
            int j = jlo + (int)(e / mc), i = (int)(e % mc);
            F[relc[i] + (int64_t)relc[j] * f] += CB[i + (int64_t)j * fc];
        }
        __syncthreads();
    }
}

// ------------------------------------------------------------------------------------------------
// Dense partial factorisation of a front:   P F = [L11 0; L21 I] [U11 U12; 0 S]
// ------------------------------------------------------------------------------------------------

// wave-wide arg-max of (value, index); ties resolved towards the smaller index (deterministic)
__device__ __forceinline__ void wave_argmax(double &v, int &i) {
    for (int off = 32; off > 0; off >>= 1) {
        double ov = __shfl_xor(v, off);
        int oi = __shfl_xor(i, off);
        if (ov > v || (ov == v && oi < i)) {
            v = ov;
            i = oi;
        }
    }
}

// One wavefront factorises one small front (f <= SMALL_F) held entirely in LDS.
// Partial pivoting searches the whole remaining pivot block (rows c..p-1).
__global__ void __launch_bounds__(64) k_small_factor(const int32_t *__restrict__ list, const FrontDesc *__restrict__ FD,
                                                     double *__restrict__ pool, int32_t *__restrict__ lperm,
                                                     const unsigned long long *__restrict__ anorm_bits, double pivot_eps,
                                                     FactorInfo *info, int32_t ld) {
    HIPMF_DYN_SHARED(double, sm);
    __shared__ int32_t lp[SMALL_F];
    const int tid = threadIdx.x;
    FrontDesc fd = FD[list[blockIdx.x]];
    const int p = fd.p, f = fd.p + fd.m;
    double *F = pool + fd.off;
    const double eps = pivot_eps * __longlong_as_double((long long)*anorm_bits);
    for (int e = tid; e < f * f; e += 64) sm[(e % f) + (e / f) * ld] = F[e];
    if (tid < p) lp[tid] = tid;
    __syncthreads();
    for (int c = 0; c < p; c++) {
        double v = -1.0;
        int idx = c + tid;
        if (idx < p) v = fabs(sm[idx + c * ld]);
        else idx = 1 << 30;
        wave_argmax(v, idx);
        const int piv = idx;
        if (piv != c) {
            if (tid < f) {
                double a = sm[c + tid * ld];
                sm[c + tid * ld] = sm[piv + tid * ld];
                sm[piv + tid * ld] = a;
            }
            if (tid == 0) {
                int a = lp[c];
                lp[c] = lp[piv];
                lp[piv] = a;
            }
            __syncthreads();
        }
        double d = sm[c + c * ld];
        if (fabs(d) < eps || d == 0.0) {
            // static pivoting: replace a tiny pivot by +-eps (wave-uniform branch: d is one LDS word)
            double dn = (d < 0.0) ? -eps : eps;
            if (dn == 0.0) dn = 1.0; // eps == 0 requested and an exact zero: keep the factors finite
            __syncthreads();
            if (tid == 0) {
                if (d == 0.0) atomicAdd(&info->n_zero_pivot, 1);
                atomicAdd(&info->n_perturbed, 1);
                sm[c + c * ld] = dn;
            }
            __syncthreads();
            d = dn;
        }
        const double inv = 1.0 / d;
        const int w = f - c - 1;
        if (tid < w) sm[(c + 1 + tid) + c * ld] *= inv;
        __syncthreads();
        for (int e = tid; e < w * w; e += 64) {
            int r = c + 1 + e % w, cc = c + 1 + e / w;
            sm[r + cc * ld] -= sm[r + c * ld] * sm[c + cc * ld];
        }
        __syncthreads();
    }
    for (int e = tid; e < f * f; e += 64) F[e] = sm[(e % f) + (e / f) * ld];
    if (tid < p) lperm[fd.first + tid] = lp[tid];
}

// Tiled path, step k0: factorise the nb x nb diagonal tile (pivoting inside the tile).
__global__ void __launch_bounds__(64) k_diag(const int32_t *__restrict__ list, const FrontDesc *__restrict__ FD, int32_t k0,
                                             double *__restrict__ pool, int32_t *__restrict__ lperm,
                                             const unsigned long long *__restrict__ anorm_bits, double pivot_eps, FactorInfo *info) {
    __shared__ double T[NB][NB + 1];
    __shared__ int32_t lp[NB];
    const int tid = threadIdx.x;
    FrontDesc fd = FD[list[blockIdx.x]];
    const int64_t f = (int64_t)fd.p + fd.m;
    const int nb = (fd.p - k0) < NB ? (fd.p - k0) : NB;
    double *F = pool + fd.off;
    const double eps = pivot_eps * __longlong_as_double((long long)*anorm_bits);
    for (int e = tid; e < nb * nb; e += 64) T[e % nb][e / nb] = F[(k0 + e % nb) + (int64_t)(k0 + e / nb) * f];
    if (tid < nb) lp[tid] = tid;
    __syncthreads();
    for (int c = 0; c < nb; c++) {
        double v = -1.0;
        int idx = c + tid;
        if (idx < nb) v = fabs(T[idx][c]);
        else idx = 1 << 30;
        wave_argmax(v, idx);
        const int piv = idx;
        if (piv != c) {
            if (tid < nb) {
                double a = T[c][tid];
                T[c][tid] = T[piv][tid];
                T[piv][tid] = a;
            }
            if (tid == 0) {
                int a = lp[c];
                lp[c] = lp[piv];
                lp[piv] = a;
            }
            __syncthreads();
        }
        double d = T[c][c];
        if (fabs(d) < eps || d == 0.0) {
            double dn = (d < 0.0) ? -eps : eps;
            if (dn == 0.0) dn = 1.0;
            __syncthreads();
            if (tid == 0) {
                if (d == 0.0) atomicAdd(&info->n_zero_pivot, 1);
                atomicAdd(&info->n_perturbed, 1);
                T[c][c] = dn;
            }
            __syncthreads();
            d = dn;
        }
        const double inv = 1.0 / d;
        const int w = nb - c - 1;
        if (tid < w) T[c + 1 + tid][c] *= inv;
        __syncthreads();
        for (int e = tid; e < w * w; e += 64) {
            int r = c + 1 + e % w, cc = c + 1 + e / w;
            T[r][cc] -= T[r][c] * T[c][cc];
        }
        __syncthreads();
    }
    for (int e = tid; e < nb * nb; e += 64) F[(k0 + e % nb) + (int64_t)(k0 + e / nb) * f] = T[e % nb][e / nb];
    if (tid < nb) lperm[fd.first + k0 + tid] = k0 + lp[tid];
}

// Tiled path, step k0: triangular solves against the diagonal tile.
//   L tiles  (rows below the tile):   L_ik = A_ik * U_kk^{-1}
//   U tiles  (columns right of it):   U_kj = L_kk^{-1} * (P A_kj)
//   left tiles (columns < k0):        rows of block k permuted only (LAPACK-style row interchange)
// One thread owns one row (L) / one column (U) of the tile and keeps it in registers.
__global__ void __launch_bounds__(PANEL_T) k_panel(const int32_t *__restrict__ pfx, int32_t nactive, const int32_t *__restrict__ list,
                                                   const FrontDesc *__restrict__ FD, int32_t k0, double *__restrict__ pool,
                                                   const int32_t *__restrict__ lperm) {
    __shared__ double D[NB][NB + 1];
    __shared__ double T[NB][PANEL_T + 1];
    __shared__ int32_t lp[NB];
    const int tid = threadIdx.x;
    const int slot = find_slot(pfx, nactive, blockIdx.x);
    const int t = blockIdx.x - pfx[slot];
    FrontDesc fd = FD[list[slot]];
    const int64_t f = (int64_t)fd.p + fd.m;
    const int nb = (fd.p - k0) < NB ? (fd.p - k0) : NB;
    const int below = (int)f - (k0 + nb);
    const int nT = (below + PANEL_T - 1) / PANEL_T;
    double *F = pool + fd.off;
    for (int e = tid; e < nb * nb; e += PANEL_T) D[e % nb][e / nb] = F[(k0 + e % nb) + (int64_t)(k0 + e / nb) * f];
    if (tid < nb) lp[tid] = lperm[fd.first + k0 + tid];
    __syncthreads();
    if (t < nT) {
        // ---- L tile ----
        const int r0 = k0 + nb + t * PANEL_T;
        const int h = ((int)f - r0) < PANEL_T ? ((int)f - r0) : PANEL_T;
        for (int e = tid; e < h * nb; e += PANEL_T) T[e / h][e % h] = F[(r0 + e % h) + (int64_t)(k0 + e / h) * f];
        __syncthreads();
        if (tid < h) {
            double x[NB];
#pragma unroll
            for (int c = 0; c < NB; c++) x[c] = (c < nb) ? T[c][tid] : 0.0;
#pragma unroll
            for (int c = 0; c < NB; c++) {
                if (c < nb) {
                    double v = x[c];
#pragma unroll
                    for (int k = 0; k < c; k++) v -= x[k] * D[k][c];
                    x[c] = v / D[c][c];
                }
            }
#pragma unroll
            for (int c = 0; c < NB; c++)
                if (c < nb) T[c][tid] = x[c];
        }
        __syncthreads();
        for (int e = tid; e < h * nb; e += PANEL_T) F[(r0 + e % h) + (int64_t)(k0 + e / h) * f] = T[e / h][e % h];
    } else {
        const bool left = t >= 2 * nT;
        const int c0 = left ? (t - 2 * nT) * PANEL_T : k0 + nb + (t - nT) * PANEL_T;
        const int cend = left ? k0 : (int)f;
        const int w = (cend - c0) < PANEL_T ? (cend - c0) : PANEL_T;
        // load with the row interchange applied: new row r <- old row lp[r]
        for (int e = tid; e < w * nb; e += PANEL_T) T[e % nb][e / nb] = F[lp[e % nb] + (int64_t)(c0 + e / nb) * f];
        __syncthreads();
        if (!left && tid < w) {
            double x[NB];
#pragma unroll
            for (int r = 0; r < NB; r++) x[r] = (r < nb) ? T[r][tid] : 0.0;
#pragma unroll
            for (int r = 1; r < NB; r++) {
                if (r < nb) {
                    double v = x[r];
#pragma unroll
                    for (int k = 0; k < r; k++) v -= D[r][k] * x[k];
                    x[r] = v;
                }
            }
#pragma unroll
            for (int r = 0; r < NB; r++)
                if (r < nb) T[r][tid] = x[r];
        }
        __syncthreads();
        for (int e = tid; e < w * nb; e += PANEL_T) F[(k0 + e % nb) + (int64_t)(c0 + e / nb) * f] = T[e % nb][e / nb];
    }
}

// Tiled path, step k0: trailing update  A22 -= L21 * U12  on v_mfma_f64_16x16x4_f64.
// A 256-thread workgroup owns a 64 x 64 tile; each of its 4 waves owns 32 x 32 = 2 x 2 MFMA tiles.
// The product is formed transposed (D = U^T L^T) so that a result register of 16 adjacent lanes
// maps to 16 consecutive rows of one column: stores are 128-byte contiguous segments.
// LDS layouts: Ls[kk][r] (ld 80) and Us[c][kk] (ld 34) make the fragment reads of ds_read_b64
// conflict-free (banks = (dword address) mod 64) and both global->LDS copies conflict-free too.
__global__ void __launch_bounds__(256) k_update(const int32_t *__restrict__ pfx, int32_t nactive, const int32_t *__restrict__ list,
                                                const FrontDesc *__restrict__ FD, int32_t k0, double *__restrict__ pool) {
    __shared__ double Ls[NB * LS_LD];
    __shared__ double Us[UPD_T * US_LD];
    const int tid = threadIdx.x;
    const int slot = find_slot(pfx, nactive, blockIdx.x);
    const int t = blockIdx.x - pfx[slot];
    FrontDesc fd = FD[list[slot]];
    const int64_t f = (int64_t)fd.p + fd.m;
    const int nb = (fd.p - k0) < NB ? (fd.p - k0) : NB;
    const int base = k0 + nb;
    const int below = (int)f - base;
    const int nt = (below + UPD_T - 1) / UPD_T;
    const int r0 = base + (t % nt) * UPD_T, c0 = base + (t / nt) * UPD_T;
    double *F = pool + fd.off;
    for (int e = tid; e < NB * UPD_T; e += 256) {
        int r = e % UPD_T, kk = e / UPD_T;
        Ls[kk * LS_LD + r] = (r0 + r < f && kk < nb) ? F[(r0 + r) + (int64_t)(k0 + kk) * f] : 0.0;
    }
    for (int e = tid; e < NB * UPD_T; e += 256) {
        int kk = e % NB, c = e / NB;
        Us[c * US_LD + kk] = (c0 + c < f && kk < nb) ? F[(k0 + kk) + (int64_t)(c0 + c) * f] : 0.0;
    }
    __syncthreads();
    const int lane = tid & 63, wave = tid >> 6;
    const int wr = (wave & 1) * 32, wc = (wave >> 1) * 32;
    const int l15 = lane & 15, l4 = lane >> 4;
    f64x4 acc[2][2];
#pragma unroll
    for (int a = 0; a < 2; a++)
#pragma unroll
        for (int b = 0; b < 2; b++) acc[a][b] = f64x4{0.0, 0.0, 0.0, 0.0};
#pragma unroll
    for (int kk0 = 0; kk0 < NB; kk0 += 4) {
        double ua[2], lb[2];
#pragma unroll
        for (int a = 0; a < 2; a++) ua[a] = Us[(wc + a * 16 + l15) * US_LD + kk0 + l4];
#pragma unroll
        for (int b = 0; b < 2; b++) lb[b] = Ls[(kk0 + l4) * LS_LD + wr + b * 16 + l15];
#pragma unroll
        for (int a = 0; a < 2; a++)
#pragma unroll
            for (int b = 0; b < 2; b++) acc[a][b] = mfma_f64_16x16x4(ua[a], lb[b], acc[a][b]);
    }
#pragma unroll
    for (int a = 0; a < 2; a++)
#pragma unroll
        for (int b = 0; b < 2; b++)
#pragma unroll
            for (int g = 0; g < 4; g++) {
                int r = r0 + wr + b * 16 + l15;
                int c = c0 + wc + a * 16 + l4 + 4 * g;
                if (r < f && c < f) F[r + (int64_t)c * f] -= acc[a][b][g];
            }
}

// ------------------------------------------------------------------------------------------------
// Level-set sparse triangular solves (multifrontal form).  One workgroup per supernode.
// ------------------------------------------------------------------------------------------------

// forward:  w = [b1 ; 0] + sum_children u_c ;  y1 = L11^{-1} P w1 ;  u = w2 - L21 y1  (kept in work)
__global__ void k_fwd(const int32_t *__restrict__ list, const FrontDesc *__restrict__ FD, const double *__restrict__ pool,
                      const int32_t *__restrict__ lperm, const int32_t *__restrict__ child_idx, const int32_t *__restrict__ rel,
                      double *__restrict__ work, double *__restrict__ x) {
    __shared__ double xb[NB];
    const int tid = threadIdx.x, nt = blockDim.x;
    FrontDesc fd = FD[list[blockIdx.x]];
    const int p = fd.p, f = fd.p + fd.m;
    const int64_t ld = f;
    const double *F = pool + fd.off;
    double *W = work + fd.woff;
    double *xs = x + fd.first;
    for (int i = tid; i < f; i += nt) W[i] = (i < p) ? xs[i] : 0.0;
    __syncthreads();
    for (int ci = fd.child_begin; ci < fd.child_end; ci++) {
        FrontDesc cd = FD[child_idx[ci]];
        const double *uc = work + cd.woff + cd.p;
        const int32_t *relc = rel + cd.rowptr;
        for (int i = tid; i < cd.m; i += nt) W[relc[i]] += uc[i];
        __syncthreads();
    }
    // row interchanges of the pivot block: xs[r] = W[lperm[r]]
    for (int i = tid; i < p; i += nt) xs[i] = W[lperm[fd.first + i]];
    __syncthreads();
    for (int j0 = 0; j0 < p; j0 += NB) {
        const int jb = (p - j0) < NB ? (p - j0) : NB;
        if (tid < 64) {
            double v = (tid < jb) ? xs[j0 + tid] : 0.0;
            for (int j = 0; j < jb; j++) {
                double vj = __shfl(v, j);
                if (tid > j && tid < jb) v -= F[(j0 + tid) + (int64_t)(j0 + j) * ld] * vj;
            }
            if (tid < jb) {
                xb[tid] = v;
                xs[j0 + tid] = v;
            }
        }
        __syncthreads();
        for (int i = j0 + jb + tid; i < f; i += nt) {
            double acc = 0.0;
            for (int j = 0; j < jb; j++) acc += F[i + (int64_t)(j0 + j) * ld] * xb[j];
            if (i < p) xs[i] -= acc;
            else W[i] -= acc;
        }
        __syncthreads();
    }
}

// backward:  x1 = U11^{-1} (y1 - U12 x2),  x2 gathered from the ancestors' solved entries
__global__ void k_bwd(const int32_t *__restrict__ list, const FrontDesc *__restrict__ FD, const double *__restrict__ pool,
                      const int32_t *__restrict__ rows, double *__restrict__ work, double *__restrict__ x) {
    __shared__ double xb[NB];
    const int tid = threadIdx.x, nt = blockDim.x;
    FrontDesc fd = FD[list[blockIdx.x]];
    const int p = fd.p, m = fd.m, f = fd.p + fd.m;
    const int64_t ld = f;
    const double *F = pool + fd.off;
    double *W = work + fd.woff;
    double *xs = x + fd.first;
    const int32_t *rws = rows + fd.rowptr;
    for (int i = tid; i < m; i += nt) W[p + i] = x[rws[i]];
    __syncthreads();
    for (int i = tid; i < p; i += nt) {
        double acc = 0.0;
        for (int j = 0; j < m; j++) acc += F[i + (int64_t)(p + j) * ld] * W[p + j];
        xs[i] -= acc;
    }
    __syncthreads();
    for (int j0 = ((p - 1) / NB) * NB; j0 >= 0; j0 -= NB) {
        const int jb = (p - j0) < NB ? (p - j0) : NB;
        if (tid < 64) {
            double v = (tid < jb) ? xs[j0 + tid] : 0.0;
            for (int j = jb - 1; j >= 0; j--) {
                if (tid == j) v /= F[(j0 + j) + (int64_t)(j0 + j) * ld];
                double vj = __shfl(v, j);
                if (tid < j) v -= F[(j0 + tid) + (int64_t)(j0 + j) * ld] * vj;
            }
            if (tid < jb) {
                xb[tid] = v;
                xs[j0 + tid] = v;
            }
        }
        __syncthreads();
        for (int i = tid; i < j0; i += nt) {
            double acc = 0.0;
            for (int j = 0; j < jb; j++) acc += F[i + (int64_t)(j0 + j) * ld] * xb[j];
            xs[i] -= acc;
        }
        __syncthreads();
    }
}

// ------------------------------------------------------------------------------------------------
// Vector kernels for the solve driver and iterative refinement
// ------------------------------------------------------------------------------------------------

// xp[i] = rs[perm[i]] * b[perm[i]]
__global__ void k_perm_in(int32_t n, const int32_t *__restrict__ perm, const double *__restrict__ rs,
                          const double *__restrict__ b, double *__restrict__ xp) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) xp[i] = rs[perm[i]] * b[perm[i]];
}

// out[perm[j]] (+)= xp[j]
__global__ void k_perm_out(int32_t n, const int32_t *__restrict__ perm, const double *__restrict__ xp,
                           double *__restrict__ out, int32_t accumulate) {
    int j = blockIdx.x * blockDim.x + threadIdx.x;
    if (j < n) {
        if (accumulate == 1) out[perm[j]] += xp[j];
        else if (accumulate == 2) out[perm[j]] -= xp[j];
        else out[perm[j]] = xp[j];
    }
}

// r = b - A x   (CSR; for symmetric-lower storage the mirrored entries come from tptr/tidx/trow)
__global__ void k_residual(int32_t n, const int32_t *__restrict__ rp, const int32_t *__restrict__ ci,
                           const double *__restrict__ vals, const int32_t *__restrict__ tptr, const int32_t *__restrict__ tidx,
                           const int32_t *__restrict__ arow, const double *__restrict__ x, const double *__restrict__ b,
                           double *__restrict__ r) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    double acc = b[i];
    for (int p = rp[i]; p < rp[i + 1]; p++) acc -= vals[p] * x[ci[p]];
    if (tptr)
        for (int q = tptr[i]; q < tptr[i + 1]; q++) acc -= vals[tidx[q]] * x[arow[tidx[q]]];
    r[i] = acc;
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

// *out = max_i |v_i| (ordered bits)
__global__ void k_norminf(int32_t n, const double *__restrict__ v, unsigned long long *out) {
    __shared__ double red[256];
    double m = 0.0;
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
        double a = fabs(v[i]);
        m = a > m ? a : m; // NaN never wins: a NaN residual leaves the previous iterate in place
    }
    red[threadIdx.x] = m;
    __syncthreads();
    for (int s = blockDim.x / 2; s > 0; s >>= 1) {
        if ((int)threadIdx.x < s && red[threadIdx.x + s] > red[threadIdx.x]) red[threadIdx.x] = red[threadIdx.x + s];
        __syncthreads();
    }
    if (threadIdx.x == 0) atomicMax(out, (unsigned long long)__double_as_longlong(red[0]));
}

__global__ void k_axpy(int32_t n, double alpha, const double *__restrict__ x, double *__restrict__ y) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) y[i] += alpha * x[i];
}

// diagonal of U in pivot order (for the determinant / rcond estimate)
__global__ void k_diag_gather(int32_t nsuper, const FrontDesc *__restrict__ FD, const double *__restrict__ pool,
                              double *__restrict__ du) {
    int s = blockIdx.x;
    if (s >= nsuper) return;
    FrontDesc fd = FD[s];
    const int64_t f = (int64_t)fd.p + fd.m;
    for (int i = threadIdx.x; i < fd.p; i += blockDim.x) du[fd.first + i] = pool[fd.off + i + i * f];
}

} // namespace hipmf
