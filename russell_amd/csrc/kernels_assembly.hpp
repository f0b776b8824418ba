// kernels_assembly.hpp -- row scaling, scatter of A into the fronts, extend-add.  All HBM-bound.
#pragma once
#include "kernels_common.hpp"

namespace hipmf {

// rs[i] = 1 / sum_j |a_ij| (mode 1, UMFPACK_SCALE_SUM), 1 / max_j |a_ij| (mode 2), 1 (mode 0).
// tptr/tidx list, for every row i, the positions of the stored entries (r, i), r != i, that the
// symmetric-lower storage mirrors into row i (empty for general storage).
__global__ void k_row_scale(int32_t n, const int32_t *__restrict__ rp, const double *__restrict__ vals,
                            const int32_t *__restrict__ tptr, const int32_t *__restrict__ tidx, int32_t mode, int32_t symmetric,
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
    // symmetric mode: S A S with s = 1 / sqrt(norm) (the same vector scales rows and columns, the fronts stay symmetric)
    rs[i] = (mode == 0 || acc == 0.0) ? 1.0 : (symmetric ? 1.0 / sqrt(acc) : 1.0 / acc);
}

// Value refresh of a fixed structure (the Radau5 / Newton repeat-factorise pattern): CSR entry j is the sum of
// the caller's entries in[seg_idx[q]], q in [seg_ptr[j], seg_ptr[j + 1]), added in that (fixed) order: the caller's COO
// triplets with their duplicates, without a host-side COO -> CSR conversion per factorisation.  A negative index ~k subtracts in[k].
__global__ void k_gather_values(int64_t nnz, const int32_t *__restrict__ seg_ptr, const int32_t *__restrict__ seg_idx,
                                const double *__restrict__ in, double *__restrict__ out) {
    for (int64_t j = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; j < nnz; j += (int64_t)gridDim.x * blockDim.x) {
        double acc = 0.0;
        for (int32_t q = seg_ptr[j]; q < seg_ptr[j + 1]; q++) {
            const int32_t k = seg_idx[q]; // k < 0: the entry ~k is SUBTRACTED (imaginary parts of the real-equivalent form of a complex matrix)
            acc += k < 0 ? -in[~k] : in[k];
        }
        out[j] = acc;
    }
}

// Expansion of a symmetric-lower value array to the general storage the handle was analysed with (interface_hipmf.cpp: a symmetric
// matrix with a weak diagonal is factorised as a general one, with the matching): out[k] = in[emap[k]].
__global__ void k_expand_values(int64_t nnz, const int32_t *__restrict__ emap, const double *__restrict__ in, double *__restrict__ out) {
    for (int64_t j = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; j < nnz; j += (int64_t)gridDim.x * blockDim.x) out[j] = in[emap[j]];
}

// Scaled values vs[k] = rs[row(k)] * a_k * cs[col(k)] (and, for symmetric-lower storage, the mirrored entry's
// vs2[k] = rs[col(k)] * a_k) for the assembly kernels, and max |vs| -> *out (as ordered bits of a non-negative double).
__global__ void k_absmax(int64_t nnz, const double *__restrict__ vals, const int32_t *__restrict__ arow, const int32_t *__restrict__ acol,
                         const double *__restrict__ rs, const double *__restrict__ cs, double *__restrict__ vs, double *__restrict__ vs2,
                         unsigned long long *out, FactorInfo *info) {
    __shared__ double red[256];
    double m = 0.0;
    bool bad = false;
    for (int64_t k = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; k < nnz; k += (int64_t)gridDim.x * blockDim.x) {
        const double v = vals[k];
        double sv = v * rs[arow[k]];
        if (cs) sv *= cs[acol[k]];
        vs[k] = sv;
        if (vs2) vs2[k] = cs ? v * rs[acol[k]] * cs[arow[k]] : v * rs[acol[k]];
        const double a = fabs(sv);
        bad |= !(a <= 1.7976931348623157e308); // NaN or Inf
        m = a > m ? a : m;
    }
    if (bad) atomicAdd(&info->n_nonfinite, 1);
    red[threadIdx.x] = m;
    __syncthreads();
    for (int s = blockDim.x / 2; s > 0; s >>= 1) {
        if ((int)threadIdx.x < s && red[threadIdx.x + s] > red[threadIdx.x]) red[threadIdx.x] = red[threadIdx.x + s];
        __syncthreads();
    }
    if (threadIdx.x == 0) atomicMax(out, (unsigned long long)__double_as_longlong(red[0]));
}

// Is the static pivot order still sound for THESE values?  Row r of A is pivot row dcol[r] of the (matched) system: its entry in
// column dcol[r] is the diagonal of the matrix that is factorised.  A row whose scaled diagonal is missing, zero or below
// `threshold` times the row's largest scaled entry counts as weak -- the criterion initialize uses to decide on the maximum-product
// matching (matching.cpp, diagonal_is_weak).  One thread per row; dcol == nullptr: identity.
__global__ void k_diag_check(int32_t n, const int32_t *__restrict__ rp, const int32_t *__restrict__ ci, const double *__restrict__ vs,
                             const int32_t *__restrict__ dcol, double threshold, FactorInfo *info, int32_t pairs) {
    const int r = blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= n) return;
    const int dc = dcol ? dcol[r] : r;
    double dg = 0.0, mx = 0.0;
    for (int p = rp[r]; p < rp[r + 1]; p++) {
        const double a = fabs(vs[p]);
        mx = a > mx ? a : mx;
        // (pairs -- the real-equivalent form of a complex matrix: the larger part of the complex diagonal entry)
        if (ci[p] == dc || (pairs && ci[p] == (dc ^ 1))) dg = a > dg ? a : dg;
    }
    if (!(dg > threshold * mx) || dg == 0.0) atomicAdd(&info->n_weak_diag, 1);
}

// pool[at[e]] = vs[k_e] (scaled values, k_absmax; k_e < 0: the mirrored copy vs2[~k_e] of a symmetric-lower entry) for the entries
// of one level's tiled fronts (their working blocks are zero-filled first; every position is hit once)
__global__ void k_scatter(int64_t cnt, const int32_t *__restrict__ sc_k, const int64_t *__restrict__ sc_at, const double *__restrict__ vs,
                          const double *__restrict__ vs2, double *__restrict__ pool) {
    for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < cnt; e += (int64_t)gridDim.x * blockDim.x) {
        const int32_t k = sc_k[e];
        pool[sc_at[e]] = k < 0 ? vs2[~k] : vs[k];
    }
}

// zero-fill of the big fronts' blocks (the small fronts are written whole by k_small_factor): one workgroup per chunk
struct ZeroTask {
    int64_t off; // pool offset (doubles)
    int32_t len; // doubles
    int32_t pad;
};
__global__ void __launch_bounds__(256) k_zero(const ZeroTask *__restrict__ tasks, double *__restrict__ pool) {
    const ZeroTask t = tasks[blockIdx.x];
    double *p = pool + t.off;
    for (int i = threadIdx.x; i < t.len; i += 256) p[i] = 0.0;
}

// identity blocks of the augmented big fronts: E(i, i) = 1 and E'(i, i) = 1 (the panels are zero-filled first)
__global__ void k_set_identity(const int32_t *__restrict__ list, const FrontDesc *__restrict__ FD, double *__restrict__ pool) {
    FrontDesc fd = FD[list[blockIdx.x]];
    const int64_t ld = fd.ld, ps = fd.ldp;
    double *E = pool + fd.eoff;
    double *Ep = fd.epoff >= 0 ? pool + fd.epoff : nullptr;
    for (int i = threadIdx.x; i < fd.p; i += blockDim.x) {
        E[i + i * ld] = 1.0;
        if (Ep) Ep[i + i * ps] = 1.0;
    }
}

// Symmetric mode: a tiled (L D L^T) front whose PARENT is a small front hands over a full contribution block: its lower triangle
// is mirrored into the upper one (the small fronts are factorised as general matrices).  One workgroup per such front; m <= SMALL_F.
__global__ void k_mirror_cb(const int32_t *__restrict__ list, const FrontDesc *__restrict__ FD, double *__restrict__ pool) {
    const FrontDesc fd = FD[list[blockIdx.x]];
    const int m = fd.m;
    const int64_t ld = fd.ld;
    double *CB = pool + fd.off + fd.p + (int64_t)fd.p * ld;
    for (int e = threadIdx.x; e < m * m; e += blockDim.x) {
        const int i = e % m, j = e / m;
        if (i > j) CB[j + (int64_t)i * ld] = CB[i + (int64_t)j * ld];
    }
}

// extend-add: every task adds the children's contribution blocks into a (column range x row range)
// tile of the parent front.  Children are visited in ascending order and a parent entry belongs to
// exactly one task, so the floating-point summation order is fixed (bit-reproducible factors).
// The sub-ranges of every child's (sorted) relative-index list that fall into the tile are precomputed on
// the host (EaRange, one per task and child that actually hits the tile, self-contained): no dependent
// descriptor loads or binary searches on the device, and tiles no child touches have no task at all.
template <bool SYM>
__global__ void k_extend_add(const EaTask *__restrict__ tasks, const EaRange *__restrict__ ranges, const int32_t *__restrict__ rel,
                             double *__restrict__ pool) {
    const EaTask t = tasks[blockIdx.x];
    const int64_t ld = t.ld;
    double *F = pool + t.f_off;
    const int tid = threadIdx.x;
    EaRange rg = ranges[t.piece_begin];
    for (int pc = t.piece_begin; pc < t.piece_end; pc++) {
        const EaRange nxt = ranges[pc + 1 < t.piece_end ? pc + 1 : pc]; // the next piece's descriptor travels with this one's data
        const int64_t ldc = rg.ldc;
        const double *CB = pool + rg.cb_off;
        const int32_t *relc = rel + rg.rel_off;
        const int jlo = rg.jlo, jhi = rg.jhi, ilo = rg.ilo, ihi = rg.ihi;
        // lane = row, sh = log2(rows per pass) chosen by the height of the piece; 256 >> sh column groups; eight
        // columns are in flight per thread before the first store (within one child the targets are distinct:
        // rel is strictly increasing)
        const int ni = ihi - ilo;
        const int sh = ni <= 16 ? 4 : (ni <= 32 ? 5 : 6);
        const int tx = tid & ((1 << sh) - 1), ty = tid >> sh, ng = (int)blockDim.x >> sh;
        for (int i = ilo + tx; i < ihi; i += (1 << sh)) {
            const int ri = relc[i];
            for (int j0 = jlo + ty; j0 < jhi; j0 += 8 * ng) {
                double cb[8], fo[8];
                int64_t at[8];
#pragma unroll
                for (int q = 0; q < 8; q++) {
                    const int j = j0 + q * ng;
                    // (sym: rel is increasing, so child entry (i, j), i >= j, lands on or below the parent's diagonal; the child's
                    //  own block is valid there whether it was factorised as LU or as L D L^T)
                    at[q] = (j < jhi && (!SYM || j <= i)) ? ri + (int64_t)relc[j] * ld : -1;
                }
#pragma unroll
                for (int q = 0; q < 8; q++) {
                    const int j = j0 + q * ng;
                    if (at[q] >= 0) {
                        cb[q] = CB[i + (int64_t)j * ldc];
                        fo[q] = F[at[q]];
                    }
                }
#pragma unroll
                for (int q = 0; q < 8; q++)
                    if (at[q] >= 0) F[at[q]] = fo[q] + cb[q];
            }
        }
        __syncthreads(); // the next child may hit the same parent entries from other threads
        rg = nxt;
    }
}

} // namespace hipmf
