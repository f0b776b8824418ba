// fdm_device.cpp -- the coefficient matrices of russell_pde's finite-difference Laplacian assembled in HBM.
//
// Restates, for the device, Fdm2d::get_matrices_sps (/root/reference/russell_pde/src/fdm_2d.rs:603-649): the "system partitioning
// strategy" matrices  K-bar (unknown x unknown)  and  K-check (unknown x prescribed)  as COO triplets IN THE REFERENCE'S ORDER --
// unknown nodes ascending, per node the molecule CUR, LEF, RIG, BOT, TOP (fdm_2d.rs:6-10) -- with
//   * the molecule  [2 (kx/dx^2 + ky/dy^2), -kx/dx^2, -kx/dx^2, -ky/dy^2, -ky/dy^2]                     (fdm_2d.rs:376-386)
//   * ghost nodes mirrored onto the interior (zero-flux boundaries: repeated column indices, kept as duplicates), or wrapped
//     when the direction is periodic                                                                   (fdm_2d.rs:944-979)
//   * rows of boundary nodes halved per non-periodic direction they are on the boundary of               (fdm_2d.rs:626-631)
//   * the Helmholtz coefficient added on the diagonal entry                                              (fdm_2d.rs:622-624)
//   * entries above / below the diagonal skipped for lower / upper triangular storage                    (fdm_2d.rs:636)
//   * the local numbering iu / ip of EquationHandler::recompute: ranks among the unknown / prescribed nodes
//                                                                                  (russell_pde/src/equation_handler.rs:153-190)
// nz > 1 gives the 7-point analogue (molecule entries 5, 6 = the neighbours below / above in z, same mirroring and halving rules);
// the reference has no Fdm3d -- SURVEY.md section 8f ranks it as the natural extension for BASELINE config 4.
//
// Structure (indices) and values are separate calls: a PDE solver that changes kx / ky / alpha between solves only re-runs
// k_fdm_values and hands the value array to solver_hipmf_factorize_mapped_device -- nothing crosses PCIe.
#include <hipmf_device_rt.h>

#include <cstdint>
#include <cstdio>
#include <new>

#include "../../include/russell_hipmf.h"

namespace {

constexpr int FDM_T = 256;

struct FdmGrid {
    int32_t nx, ny, nz;
    int32_t px, py, pz; // periodic along x / y / z
    int32_t sym;        // 0: all entries, 1: lower triangle (m >= n), 2: upper triangle (m <= n)
};

// column index of molecule entry b of row m (b: 0 CUR, 1 LEF, 2 RIG, 3 BOT, 4 TOP, 5 z-, 6 z+)
__device__ __forceinline__ int64_t fdm_neighbour(const FdmGrid &g, int64_t m, int i, int j, int k, int b) {
    const int64_t nxy = (int64_t)g.nx * g.ny;
    switch (b) {
    case 1: return g.px ? (i != 0 ? m - 1 : m + (g.nx - 1)) : (i != 0 ? m - 1 : m + 1);
    case 2: return g.px ? (i != g.nx - 1 ? m + 1 : m - (g.nx - 1)) : (i != g.nx - 1 ? m + 1 : m - 1);
    case 3: return g.py ? (j != 0 ? m - g.nx : m + (int64_t)(g.ny - 1) * g.nx) : (j != 0 ? m - g.nx : m + g.nx);
    case 4: return g.py ? (j != g.ny - 1 ? m + g.nx : m - (int64_t)(g.ny - 1) * g.nx) : (j != g.ny - 1 ? m + g.nx : m - g.nx);
    case 5: return g.pz ? (k != 0 ? m - nxy : m + (int64_t)(g.nz - 1) * nxy) : (k != 0 ? m - nxy : m + nxy);
    case 6: return g.pz ? (k != g.nz - 1 ? m + nxy : m - (int64_t)(g.nz - 1) * nxy) : (k != g.nz - 1 ? m + nxy : m - nxy);
    default: return m;
    }
}

__device__ __forceinline__ bool fdm_skip(const FdmGrid &g, int64_t m, int64_t n) { return (g.sym == 1 && m < n) || (g.sym == 2 && m > n); }

// per node: [is unknown, is prescribed, entries in K-bar, entries in K-check]; pass 1 sums them per block of FDM_T nodes
__device__ __forceinline__ void fdm_counts(const FdmGrid &g, const uint8_t *presc, int64_t m, int64_t ntot, int32_t c[4]) {
    c[0] = c[1] = c[2] = c[3] = 0;
    if (m >= ntot) return;
    if (presc && presc[m]) {
        c[1] = 1;
        return;
    }
    c[0] = 1;
    const int i = (int)(m % g.nx), j = (int)((m / g.nx) % g.ny), k = (int)(m / ((int64_t)g.nx * g.ny));
    const int nb = g.nz > 1 ? 7 : 5;
    for (int b = 0; b < nb; b++) {
        const int64_t n = fdm_neighbour(g, m, i, j, k, b);
        if (presc && presc[n]) c[3]++;
        else if (!fdm_skip(g, m, n)) c[2]++;
    }
}

__global__ void __launch_bounds__(FDM_T) k_fdm_block_sums(FdmGrid g, const uint8_t *__restrict__ presc, int64_t ntot, int64_t *__restrict__ bsum) {
    __shared__ int32_t red[4][FDM_T];
    const int tid = threadIdx.x;
    int32_t c[4];
    fdm_counts(g, presc, (int64_t)blockIdx.x * FDM_T + tid, ntot, c);
    for (int q = 0; q < 4; q++) red[q][tid] = c[q];
    __syncthreads();
    for (int off = FDM_T / 2; off > 0; off >>= 1) {
        if (tid < off)
            for (int q = 0; q < 4; q++) red[q][tid] += red[q][tid + off];
        __syncthreads();
    }
    if (tid < 4) bsum[(int64_t)blockIdx.x * 4 + tid] = red[tid][0];
}

// exclusive scan of the block sums (one workgroup; nblocks <= a few hundred thousand) and the four totals
__global__ void __launch_bounds__(1024) k_fdm_scan_blocks(int64_t nblocks, int64_t *__restrict__ bsum, int64_t *__restrict__ totals) {
    __shared__ int64_t part[4][1024];
    const int tid = threadIdx.x;
    const int64_t per = (nblocks + 1023) / 1024, b0 = (int64_t)tid * per, b1 = b0 + per < nblocks ? b0 + per : nblocks;
    int64_t s[4] = {0, 0, 0, 0};
    for (int64_t b = b0; b < b1; b++)
        for (int q = 0; q < 4; q++) s[q] += bsum[b * 4 + q];
    for (int q = 0; q < 4; q++) part[q][tid] = s[q];
    __syncthreads();
    if (tid < 4) { // (1024 partial sums per quantity: a serial scan by one lane each is microseconds)
        int64_t run = 0;
        for (int t = 0; t < 1024; t++) {
            const int64_t v = part[tid][t];
            part[tid][t] = run;
            run += v;
        }
        totals[tid] = run;
    }
    __syncthreads();
    int64_t run[4];
    for (int q = 0; q < 4; q++) run[q] = part[q][tid];
    for (int64_t b = b0; b < b1; b++)
        for (int q = 0; q < 4; q++) {
            const int64_t v = bsum[b * 4 + q];
            bsum[b * 4 + q] = run[q];
            run[q] += v;
        }
}

// per node: local number (iu or ip) and the offsets of its first entries in K-bar / K-check
__global__ void __launch_bounds__(FDM_T) k_fdm_node_offsets(FdmGrid g, const uint8_t *__restrict__ presc, int64_t ntot, const int64_t *__restrict__ bsum,
                                                           int32_t *__restrict__ local, int64_t *__restrict__ off_bar, int64_t *__restrict__ off_chk) {
    __shared__ int32_t sc[4][FDM_T];
    const int tid = threadIdx.x;
    const int64_t m = (int64_t)blockIdx.x * FDM_T + tid;
    int32_t c[4];
    fdm_counts(g, presc, m, ntot, c);
    for (int q = 0; q < 4; q++) sc[q][tid] = c[q];
    __syncthreads();
    for (int off = 1; off < FDM_T; off <<= 1) { // inclusive Hillis-Steele scan of the four counts
        int32_t v[4];
        for (int q = 0; q < 4; q++) v[q] = tid >= off ? sc[q][tid - off] : 0;
        __syncthreads();
        for (int q = 0; q < 4; q++) sc[q][tid] += v[q];
        __syncthreads();
    }
    if (m >= ntot) return;
    const int64_t *bs = bsum + (int64_t)blockIdx.x * 4;
    local[m] = (int32_t)(c[0] ? bs[0] + sc[0][tid] - 1 : bs[1] + sc[1][tid] - 1);
    off_bar[m] = bs[2] + sc[2][tid] - c[2];
    off_chk[m] = bs[3] + sc[3][tid] - c[3];
}

// the triplets of the unknown node m: indices (VALUES = false) or values (VALUES = true)
template <bool VALUES>
__global__ void __launch_bounds__(FDM_T) k_fdm_fill(FdmGrid g, const uint8_t *__restrict__ presc, int64_t ntot, const int32_t *__restrict__ local,
                                                   const int64_t *__restrict__ off_bar, const int64_t *__restrict__ off_chk,
                                                   int32_t *__restrict__ bar_i, int32_t *__restrict__ bar_j, int32_t *__restrict__ chk_i, int32_t *__restrict__ chk_j,
                                                   double *__restrict__ bar_v, double *__restrict__ chk_v, double mol0, double molx, double moly, double molz, double alpha) {
    const int64_t m = (int64_t)blockIdx.x * FDM_T + threadIdx.x;
    if (m >= ntot || (presc && presc[m])) return;
    const int i = (int)(m % g.nx), j = (int)((m / g.nx) % g.ny), k = (int)(m / ((int64_t)g.nx * g.ny));
    const int nb = g.nz > 1 ? 7 : 5;
    const int32_t iu = local[m];
    int64_t ob = off_bar[m], oc = off_chk[m];
    // a row of a boundary node is halved once per non-periodic direction whose boundary it lies on
    double scale = 1.0;
    if (!g.px && (i == 0 || i == g.nx - 1)) scale *= 0.5;
    if (!g.py && (j == 0 || j == g.ny - 1)) scale *= 0.5;
    if (g.nz > 1 && !g.pz && (k == 0 || k == g.nz - 1)) scale *= 0.5;
    for (int b = 0; b < nb; b++) {
        const int64_t n = fdm_neighbour(g, m, i, j, k, b);
        double val = 0.0;
        if (VALUES) {
            val = b == 0 ? mol0 : (b <= 2 ? molx : (b <= 4 ? moly : molz));
            if (m == n) val += alpha;
            val *= scale; // (the reference divides by 2 once or twice: exact in binary either way)
        }
        if (presc && presc[n]) {
            if (VALUES) chk_v[oc] = val;
            else chk_i[oc] = iu, chk_j[oc] = local[n];
            oc++;
        } else if (!fdm_skip(g, m, n)) {
            if (VALUES) bar_v[ob] = val;
            else bar_i[ob] = iu, bar_j[ob] = local[n];
            ob++;
        }
    }
}

// Lagrange-multiplier form (Fdm2d::get_matrices_lmm, fdm_2d.rs:672-748): M = [K C^T; C 0].  The molecule of EVERY node at off_all[m]
// (the offsets of a numbering without prescribed nodes), then per prescribed node the entries of C (row neq + ip, column m) and / or
// C^T at  nnz(K) + ip * per_presc  (lower storage keeps C, upper C^T, general storage both, C first).
template <bool VALUES>
__global__ void __launch_bounds__(FDM_T) k_fdm_lmm_fill(FdmGrid g, const uint8_t *__restrict__ presc, int64_t ntot, const int32_t *__restrict__ local,
                                                       const int64_t *__restrict__ off_all, int64_t nnz_k, int32_t *__restrict__ mm_i, int32_t *__restrict__ mm_j,
                                                       double *__restrict__ mm_v, double mol0, double molx, double moly, double molz, double alpha) {
    const int64_t m = (int64_t)blockIdx.x * FDM_T + threadIdx.x;
    if (m >= ntot) return;
    const int i = (int)(m % g.nx), j = (int)((m / g.nx) % g.ny), k = (int)(m / ((int64_t)g.nx * g.ny));
    const int nb = g.nz > 1 ? 7 : 5;
    int64_t o = off_all[m];
    double scale = 1.0;
    if (!g.px && (i == 0 || i == g.nx - 1)) scale *= 0.5;
    if (!g.py && (j == 0 || j == g.ny - 1)) scale *= 0.5;
    if (g.nz > 1 && !g.pz && (k == 0 || k == g.nz - 1)) scale *= 0.5;
    for (int b = 0; b < nb; b++) {
        const int64_t n = fdm_neighbour(g, m, i, j, k, b);
        if (fdm_skip(g, m, n)) continue;
        if (VALUES) {
            double val = b == 0 ? mol0 : (b <= 2 ? molx : (b <= 4 ? moly : molz));
            if (m == n) val += alpha;
            mm_v[o] = val * scale;
        } else
            mm_i[o] = (int32_t)m, mm_j[o] = (int32_t)n;
        o++;
    }
    if (presc && presc[m]) {
        const int64_t ip = local[m];
        int64_t q = nnz_k + ip * (g.sym == 0 ? 2 : 1);
        if (g.sym != 2) { // C
            if (VALUES) mm_v[q] = 1.0;
            else mm_i[q] = (int32_t)(ntot + ip), mm_j[q] = (int32_t)m;
            q++;
        }
        if (g.sym != 1) { // C^T
            if (VALUES) mm_v[q] = 1.0;
            else mm_i[q] = (int32_t)m, mm_j[q] = (int32_t)(ntot + ip);
        }
    }
}

struct FdmHandle {
    int64_t *d_off_all = nullptr; // Lagrange-multiplier form: offsets of every node's molecule (built at the first hipmf_fdm_lmm_* call)
    int64_t nnz_k_all = 0;
    FdmGrid g;
    int64_t ntot = 0, totals[4] = {0, 0, 0, 0}; // nu, np, nnz(K-bar), nnz(K-check)
    uint8_t *d_presc = nullptr;
    int32_t *d_local = nullptr;
    int64_t *d_off_bar = nullptr, *d_off_chk = nullptr;
    int device = 0;
};

// a handle lives on the device that was current when it was created; every entry point switches to it and restores the caller's
struct FdmDeviceScope {
    int saved = -1;
    explicit FdmDeviceScope(int dev) {
        if (hipGetDevice(&saved) != hipSuccess) saved = -1;
        if (saved != dev) (void)hipSetDevice(dev);
        else saved = -1;
    }
    ~FdmDeviceScope() {
        if (saved >= 0) (void)hipSetDevice(saved);
    }
};

void fdm_free(FdmHandle *h) {
    if (!h) return;
    FdmDeviceScope scope(h->device);
    if (h->d_presc) (void)hipFree(h->d_presc);
    if (h->d_local) (void)hipFree(h->d_local);
    if (h->d_off_bar) (void)hipFree(h->d_off_bar);
    if (h->d_off_chk) (void)hipFree(h->d_off_chk);
    if (h->d_off_all) (void)hipFree(h->d_off_all);
    delete h;
}

// offsets of the molecules of ALL nodes (the counting kernels run without the prescribed mask); 0 or a status code
int32_t fdm_ensure_lmm(FdmHandle *h) {
    if (h->d_off_all) return 0;
    if ((int64_t)h->ntot + h->totals[1] > 0x7fffffffLL) return 803; // rows neq + ip are int32 in the triplets
    const int64_t nblocks = (h->ntot + FDM_T - 1) / FDM_T;
    int64_t *d_bsum = nullptr, *d_tot = nullptr, *d_chk = nullptr, *d_all = nullptr;
    int32_t *d_loc = nullptr;
    bool ok = hipMalloc((void **)&d_bsum, sizeof(int64_t) * 4 * (size_t)nblocks) == hipSuccess;
    ok = ok && hipMalloc((void **)&d_tot, sizeof(int64_t) * 4) == hipSuccess;
    ok = ok && hipMalloc((void **)&d_loc, sizeof(int32_t) * (size_t)h->ntot) == hipSuccess;
    ok = ok && hipMalloc((void **)&d_chk, sizeof(int64_t) * (size_t)h->ntot) == hipSuccess;
    ok = ok && hipMalloc((void **)&d_all, sizeof(int64_t) * (size_t)h->ntot) == hipSuccess;
    int64_t tot[4] = {0, 0, 0, 0};
    if (ok) {
        hipLaunchKernelGGL(k_fdm_block_sums, dim3((unsigned)nblocks), dim3(FDM_T), 0, 0, h->g, (const uint8_t *)nullptr, h->ntot, d_bsum);
        hipLaunchKernelGGL(k_fdm_scan_blocks, dim3(1), dim3(1024), 0, 0, nblocks, d_bsum, d_tot);
        hipLaunchKernelGGL(k_fdm_node_offsets, dim3((unsigned)nblocks), dim3(FDM_T), 0, 0, h->g, (const uint8_t *)nullptr, h->ntot, d_bsum, d_loc, d_all, d_chk);
        ok = hipMemcpy(tot, d_tot, sizeof(int64_t) * 4, hipMemcpyDeviceToHost) == hipSuccess && hipGetLastError() == hipSuccess;
    }
    if (d_bsum) (void)hipFree(d_bsum);
    if (d_tot) (void)hipFree(d_tot);
    if (d_loc) (void)hipFree(d_loc);
    if (d_chk) (void)hipFree(d_chk);
    if (!ok) {
        if (d_all) (void)hipFree(d_all);
        return 200000; // ERROR_MALLOC's neighbour on the device side would be ERROR_HIP_MALLOC; the shared code says "memory"
    }
    h->d_off_all = d_all;
    h->nnz_k_all = tot[2];
    return 0;
}

} // namespace

extern "C" {

void *hipmf_fdm_new(int32_t nx, int32_t ny, int32_t nz, int32_t periodic_x, int32_t periodic_y, int32_t periodic_z, int32_t sym,
                    const uint8_t *prescribed) {
    // (a periodic direction needs three points for the wrapped neighbours to be distinct from the node, as in the reference's grids)
    if (nx < 2 || ny < 2 || nz < 1 || sym < 0 || sym > 2) return nullptr;
    if ((int64_t)nx * ny * nz > 0x7fffffffLL) return nullptr; // node numbers are int32 in the triplets
    FdmHandle *h = new (std::nothrow) FdmHandle;
    if (!h) return nullptr;
    h->g = FdmGrid{nx, ny, nz, periodic_x ? 1 : 0, periodic_y ? 1 : 0, (nz > 1 && periodic_z) ? 1 : 0, sym};
    h->ntot = (int64_t)nx * ny * nz;
    if (hipGetDevice(&h->device) != hipSuccess) {
        delete h;
        return nullptr;
    }
    const int64_t nblocks = (h->ntot + FDM_T - 1) / FDM_T;
    int64_t *d_bsum = nullptr, *d_tot = nullptr;
    bool ok = true;
    if (prescribed) {
        ok = ok && hipMalloc((void **)&h->d_presc, (size_t)h->ntot) == hipSuccess;
        ok = ok && hipMemcpy(h->d_presc, prescribed, (size_t)h->ntot, hipMemcpyHostToDevice) == hipSuccess;
    }
    ok = ok && hipMalloc((void **)&h->d_local, sizeof(int32_t) * (size_t)h->ntot) == hipSuccess;
    ok = ok && hipMalloc((void **)&h->d_off_bar, sizeof(int64_t) * (size_t)h->ntot) == hipSuccess;
    ok = ok && hipMalloc((void **)&h->d_off_chk, sizeof(int64_t) * (size_t)h->ntot) == hipSuccess;
    ok = ok && hipMalloc((void **)&d_bsum, sizeof(int64_t) * 4 * (size_t)nblocks) == hipSuccess;
    ok = ok && hipMalloc((void **)&d_tot, sizeof(int64_t) * 4) == hipSuccess;
    if (ok) {
        hipLaunchKernelGGL(k_fdm_block_sums, dim3((unsigned)nblocks), dim3(FDM_T), 0, 0, h->g, h->d_presc, h->ntot, d_bsum);
        hipLaunchKernelGGL(k_fdm_scan_blocks, dim3(1), dim3(1024), 0, 0, nblocks, d_bsum, d_tot);
        hipLaunchKernelGGL(k_fdm_node_offsets, dim3((unsigned)nblocks), dim3(FDM_T), 0, 0, h->g, h->d_presc, h->ntot, d_bsum, h->d_local, h->d_off_bar,
                           h->d_off_chk);
        ok = hipMemcpy(h->totals, d_tot, sizeof(int64_t) * 4, hipMemcpyDeviceToHost) == hipSuccess && hipGetLastError() == hipSuccess;
    }
    if (d_bsum) (void)hipFree(d_bsum);
    if (d_tot) (void)hipFree(d_tot);
    if (!ok) {
        fdm_free(h);
        return nullptr;
    }
    return h;
}

void hipmf_fdm_drop(void *handle) { fdm_free((FdmHandle *)handle); }

int32_t hipmf_fdm_dims(const void *handle, int64_t *nu, int64_t *np, int64_t *nnz_bar, int64_t *nnz_check) {
    const FdmHandle *h = (const FdmHandle *)handle;
    if (!h || !nu || !np || !nnz_bar || !nnz_check) return 100000; // ERROR_NULL_POINTER (constants.h:6)
    *nu = h->totals[0], *np = h->totals[1], *nnz_bar = h->totals[2], *nnz_check = h->totals[3];
    return 0;
}

int32_t hipmf_fdm_structure_device(const void *handle, int32_t *d_bar_i, int32_t *d_bar_j, int32_t *d_check_i, int32_t *d_check_j) {
    const FdmHandle *h = (const FdmHandle *)handle;
    if (!h || !d_bar_i || !d_bar_j) return 100000;
    if (h->totals[3] > 0 && (!d_check_i || !d_check_j)) return 100000;
    FdmDeviceScope scope(h->device);
    const int64_t nblocks = (h->ntot + FDM_T - 1) / FDM_T;
    hipLaunchKernelGGL(k_fdm_fill<false>, dim3((unsigned)nblocks), dim3(FDM_T), 0, 0, h->g, h->d_presc, h->ntot, h->d_local, h->d_off_bar, h->d_off_chk, d_bar_i,
                       d_bar_j, d_check_i, d_check_j, (double *)nullptr, (double *)nullptr, 0.0, 0.0, 0.0, 0.0, 0.0);
    if (hipDeviceSynchronize() != hipSuccess || hipGetLastError() != hipSuccess) return 350; // ERROR_HIP_LAUNCH
    return 0;
}

int32_t hipmf_fdm_values_device(const void *handle, double dx, double dy, double dz, double kx, double ky, double kz, double alpha, double *d_bar_values,
                                double *d_check_values) {
    const FdmHandle *h = (const FdmHandle *)handle;
    if (!h || !d_bar_values) return 100000;
    if (h->totals[3] > 0 && !d_check_values) return 100000;
    if (!(dx > 0.0) || !(dy > 0.0) || (h->g.nz > 1 && !(dz > 0.0))) return 803; // ERROR_HIPMF_INVALID_VALUE
    FdmDeviceScope scope(h->device);
    const double bx = kx / (dx * dx), by = ky / (dy * dy), bz = h->g.nz > 1 ? kz / (dz * dz) : 0.0;
    const int64_t nblocks = (h->ntot + FDM_T - 1) / FDM_T;
    hipLaunchKernelGGL(k_fdm_fill<true>, dim3((unsigned)nblocks), dim3(FDM_T), 0, 0, h->g, h->d_presc, h->ntot, h->d_local, h->d_off_bar, h->d_off_chk,
                       (int32_t *)nullptr, (int32_t *)nullptr, (int32_t *)nullptr, (int32_t *)nullptr, d_bar_values, d_check_values, 2.0 * (bx + by + bz), -bx, -by, -bz,
                       alpha);
    if (hipDeviceSynchronize() != hipSuccess || hipGetLastError() != hipSuccess) return 350;
    return 0;
}

// ---- Lagrange-multiplier form (Fdm2d::get_matrices_lmm, fdm_2d.rs:672-748) -----------------------------------------------------------
int32_t hipmf_fdm_lmm_dims(void *handle, int64_t *neq, int64_t *nlag, int64_t *nnz) {
    FdmHandle *h = (FdmHandle *)handle;
    if (!h || !neq || !nlag || !nnz) return 100000;
    FdmDeviceScope scope(h->device);
    const int32_t rc = fdm_ensure_lmm(h);
    if (rc != 0) return rc;
    *neq = h->ntot, *nlag = h->totals[1];
    *nnz = h->nnz_k_all + h->totals[1] * (h->g.sym == 0 ? 2 : 1);
    return 0;
}

int32_t hipmf_fdm_lmm_structure_device(void *handle, int32_t *d_i, int32_t *d_j) {
    FdmHandle *h = (FdmHandle *)handle;
    if (!h || !d_i || !d_j) return 100000;
    FdmDeviceScope scope(h->device);
    const int32_t rc = fdm_ensure_lmm(h);
    if (rc != 0) return rc;
    const int64_t nblocks = (h->ntot + FDM_T - 1) / FDM_T;
    hipLaunchKernelGGL(k_fdm_lmm_fill<false>, dim3((unsigned)nblocks), dim3(FDM_T), 0, 0, h->g, h->d_presc, h->ntot, h->d_local, h->d_off_all, h->nnz_k_all, d_i, d_j,
                       (double *)nullptr, 0.0, 0.0, 0.0, 0.0, 0.0);
    if (hipDeviceSynchronize() != hipSuccess || hipGetLastError() != hipSuccess) return 350;
    return 0;
}

int32_t hipmf_fdm_lmm_values_device(void *handle, double dx, double dy, double dz, double kx, double ky, double kz, double alpha, double *d_values) {
    FdmHandle *h = (FdmHandle *)handle;
    if (!h || !d_values) return 100000;
    if (!(dx > 0.0) || !(dy > 0.0) || (h->g.nz > 1 && !(dz > 0.0))) return 803;
    FdmDeviceScope scope(h->device);
    const int32_t rc = fdm_ensure_lmm(h);
    if (rc != 0) return rc;
    const double bx = kx / (dx * dx), by = ky / (dy * dy), bz = h->g.nz > 1 ? kz / (dz * dz) : 0.0;
    const int64_t nblocks = (h->ntot + FDM_T - 1) / FDM_T;
    hipLaunchKernelGGL(k_fdm_lmm_fill<true>, dim3((unsigned)nblocks), dim3(FDM_T), 0, 0, h->g, h->d_presc, h->ntot, h->d_local, h->d_off_all, h->nnz_k_all,
                       (int32_t *)nullptr, (int32_t *)nullptr, d_values, 2.0 * (bx + by + bz), -bx, -by, -bz, alpha);
    if (hipDeviceSynchronize() != hipSuccess || hipGetLastError() != hipSuccess) return 350;
    return 0;
}

} // extern "C"
