// front_bench.hip -- k_front (kernels_factor_front.hpp) in isolation: synthetic fronts, time per launch, device-clock stamps of the
// phases of workgroup 0 (-DHIPMF_STAMPS), and a check of E / E' / the contribution block against a host elimination.
// build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -DHIPMF_STAMPS -I russell_amd/csrc/rt_hip -I russell_amd/csrc tools/microbench/front_bench.hip -o tools/microbench/front_bench
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <vector>

#include "kernels_factor_front.hpp"

using namespace hipmf;

#define CK(x)                                                                  \
    do {                                                                       \
        hipError_t e_ = (x);                                                   \
        if (e_ != hipSuccess) {                                                \
            printf("%s: %s\n", #x, hipGetErrorString(e_));                     \
            exit(1);                                                           \
        }                                                                      \
    } while (0)

template <int CM> static void launch(int n, size_t dyn, const FrontDesc *fd, double *pool, int32_t *lperm, unsigned long long *an, FactorInfo *info, double *diag) {
    CK(hipFuncSetAttribute(reinterpret_cast<const void *>(k_front<CM>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)(sizeof(double) * MID_LDS_DOUBLES)));
    hipLaunchKernelGGL(k_front<CM>, dim3(n), dim3(64 * MID_NW), dyn, 0, fd, pool, lperm, an, 1e-13, info, diag);
}

static bool g_lu = false; // argv[2] = 1: k_front_lu for the fronts with at most 32 pivots
static void run(int p, int m, int nf, bool check) {
    const int f = p + m, ld = f;
    const int64_t per = (int64_t)f * f + (int64_t)f * p + (int64_t)p * f; // F | E | E'
    std::vector<double> h((size_t)per * nf);
    std::vector<FrontDesc> fd((size_t)nf);
    srand(1234);
    for (int s = 0; s < nf; s++) {
        double *F = h.data() + (int64_t)s * per;
        for (int c = 0; c < f; c++)
            for (int r = 0; r < f; r++) F[r + (int64_t)c * ld] = (rand() / (double)RAND_MAX - 0.5) / f;
        for (int r = 0; r < f; r++) F[r + (int64_t)r * ld] += (s % 3 == 0 && r < p) ? 0.02 : 1.0; // (every third front: a weak diagonal in the pivot block -> interchanges)
        FrontDesc &d = fd[(size_t)s];
        d.off = (int64_t)s * per, d.eoff = d.off + (int64_t)f * f, d.epoff = d.eoff + (int64_t)f * p;
        d.p = p, d.m = m, d.first = s * p, d.ld = ld, d.flags = FD_BIG | FD_DENSE_TOP, d.ugroup = 2;
        d.rowptr = 0, d.woff = 0, d.child_begin = d.child_end = 0, d.parent = -1, d.ldp = p;
    }
    double *pool, *diag;
    int32_t *lperm;
    FrontDesc *dfd;
    unsigned long long *an;
    FactorInfo *info;
    CK(hipMalloc(&pool, sizeof(double) * h.size()));
    CK(hipMalloc(&diag, sizeof(double) * (size_t)nf * p));
    CK(hipMalloc(&lperm, sizeof(int32_t) * (size_t)nf * p));
    CK(hipMalloc(&dfd, sizeof(FrontDesc) * (size_t)nf));
    CK(hipMalloc(&an, 8));
    CK(hipMalloc(&info, sizeof(FactorInfo)));
    CK(hipMemset(info, 0, sizeof(FactorInfo)));
    double one = 1.0;
    CK(hipMemcpy(an, &one, 8, hipMemcpyHostToDevice));
    CK(hipMemcpy(dfd, fd.data(), sizeof(FrontDesc) * (size_t)nf, hipMemcpyHostToDevice));
    const bool lu = g_lu && p <= MIDL_P && midl_lds_doubles(p, m) <= MIDL_LDS_DOUBLES;
    const size_t dyn = sizeof(double) * (size_t)(lu ? midl_lds_doubles(p, m) : mid_lds_doubles(p, m));
    if (!lu && (mid_lds_doubles(p, m) > MID_LDS_DOUBLES || m > MID_MMAX)) {
        printf("p=%3d m=%3d: not eligible (%d doubles of LDS)\n", p, m, mid_lds_doubles(p, m));
        return;
    }
    const int cls = m <= 80 ? 0 : (m <= 128 ? 1 : 2);
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    float best = 1e30f;
    for (int rep = 0; rep < 4; rep++) {
        CK(hipMemcpy(pool, h.data(), sizeof(double) * h.size(), hipMemcpyHostToDevice));
        CK(hipEventRecord(e0, 0));
        if (lu) {
            CK(hipFuncSetAttribute(reinterpret_cast<const void *>(k_front_lu<false>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)(sizeof(double) * MIDL_LDS_DOUBLES)));
            hipLaunchKernelGGL(k_front_lu<false>, dim3(nf), dim3(64 * MIDL_NW), dyn, 0, dfd, pool, lperm, an, 1e-13, info, diag);
        } else if (cls == 0) launch<10>(nf, dyn, dfd, pool, lperm, an, info, diag);
        else if (cls == 1) launch<16>(nf, dyn, dfd, pool, lperm, an, info, diag);
        else launch<24>(nf, dyn, dfd, pool, lperm, an, info, diag);
        CK(hipEventRecord(e1, 0));
        CK(hipEventSynchronize(e1));
        float ms;
        CK(hipEventElapsedTime(&ms, e0, e1));
        if (rep > 0 && ms < best) best = ms;
    }
    CK(hipGetLastError());
    printf("%s p=%3d m=%3d f=%3d fronts=%5d class=%d  launch %8.1f us  (%.2f us per front and CU-slot, %.2f GFLOP/s)", lu ? "k_front_lu" : "k_front   ", p, m, f, nf, cls, best * 1e3,
           best * 1e3 / std::max(1.0, nf / 256.0), 2.0 * p * f * (double)f * nf / (best * 1e-3) * 1e-9);
#ifdef HIPMF_STAMPS
    {
        unsigned long long st[16];
        CK(hipMemcpyFromSymbol(st, HIP_SYMBOL(hipmf::hipmf_stamps), sizeof(st)));
        printf("  stamps(us):");
        for (int i = 1; i < 8; i++)
            if (st[i]) printf(" %.2f", (double)(st[i] - st[0]) / 100.0);
    }
#endif
    if (check) {
        std::vector<double> g(h.size());
        std::vector<double> dg((size_t)nf * p);
        CK(hipMemcpy(g.data(), pool, sizeof(double) * g.size(), hipMemcpyDeviceToHost));
        CK(hipMemcpy(dg.data(), diag, sizeof(double) * dg.size(), hipMemcpyDeviceToHost));
        double worst = 0.0;
        for (int s = 0; s < nf; s += std::max(1, nf / 7)) {
            const double *F0 = h.data() + (int64_t)s * per;
            const double *S = g.data() + (int64_t)s * per, *E = S + (int64_t)f * f, *Ep = E + (int64_t)f * p;
            // host: G = inv(F11) by Gauss-Jordan with partial pivoting (long double)
            std::vector<long double> A((size_t)p * 2 * p);
            for (int r = 0; r < p; r++)
                for (int c = 0; c < p; c++) A[(size_t)r * 2 * p + c] = F0[r + (int64_t)c * ld], A[(size_t)r * 2 * p + p + c] = r == c;
            for (int k = 0; k < p; k++) {
                int pv = k;
                for (int r = k; r < p; r++)
                    if (fabsl(A[(size_t)r * 2 * p + k]) > fabsl(A[(size_t)pv * 2 * p + k])) pv = r;
                for (int c = 0; c < 2 * p; c++) std::swap(A[(size_t)k * 2 * p + c], A[(size_t)pv * 2 * p + c]);
                const long double d = A[(size_t)k * 2 * p + k];
                for (int c = 0; c < 2 * p; c++) A[(size_t)k * 2 * p + c] /= d;
                for (int r = 0; r < p; r++)
                    if (r != k) {
                        const long double l = A[(size_t)r * 2 * p + k];
                        for (int c = 0; c < 2 * p; c++) A[(size_t)r * 2 * p + c] -= l * A[(size_t)k * 2 * p + c];
                    }
            }
            auto G = [&](int r, int c) { return A[(size_t)r * 2 * p + p + c]; };
            for (int r = 0; r < p; r++)
                for (int c = 0; c < p; c++) worst = std::max(worst, (double)fabsl(G(r, c) - E[r + (int64_t)c * ld]));
            for (int i = 0; i < m; i++)
                for (int c = 0; c < p; c++) {
                    long double w = 0;
                    for (int k = 0; k < p; k++) w += (long double)F0[(p + i) + (int64_t)k * ld] * G(k, c);
                    worst = std::max(worst, (double)fabsl(-w - E[(p + i) + (int64_t)c * ld]));
                }
            for (int r = 0; r < p; r++)
                for (int c = 0; c < f; c++) {
                    long double v = 0;
                    if (c < p) v = r == c;
                    else
                        for (int k = 0; k < p; k++) v -= G(r, k) * F0[k + (int64_t)c * ld];
                    worst = std::max(worst, (double)fabsl(v - Ep[r + (int64_t)c * p]));
                }
            for (int i = 0; i < m; i++)
                for (int c = 0; c < m; c++) {
                    long double v = F0[(p + i) + (int64_t)(p + c) * ld];
                    for (int k = 0; k < p; k++) v += (long double)E[(p + i) + (int64_t)k * ld] * F0[k + (int64_t)(p + c) * ld];
                    worst = std::max(worst, (double)fabsl(v - S[(p + i) + (int64_t)(p + c) * ld]));
                }
        }
        printf("  max |diff| vs host %.2e", worst);
    }
    printf("\n");
    hipFree(pool), hipFree(diag), hipFree(lperm), hipFree(dfd), hipFree(an), hipFree(info);
}

int main(int argc, char **argv) {
    const bool check = argc > 1 && atoi(argv[1]) != 0;
    g_lu = argc > 2 && atoi(argv[2]) != 0;
    const int cfg[][3] = {{2, 70, 64}, {5, 75, 256}, {16, 50, 1},  {16, 50, 256},  {16, 50, 1024}, {30, 70, 1},  {30, 70, 256}, {30, 70, 768}, {30, 70, 1536}, {40, 100, 1},
                          {40, 100, 512}, {32, 120, 1}, {32, 120, 512}, {48, 120, 512}, {32, 192, 1}, {32, 192, 256}};
    for (auto &c : cfg) run(c[0], c[1], c[2], check);
    return 0;
}
