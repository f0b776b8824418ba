// tile_bench.hip -- the 32 x 32 diagonal-tile kernels in isolation: in-register LU (tile_lu32), Gauss-Jordan inverse unrolled (tile_inv32)
// and as a loop (tile_inv32_rot), by one wavefront per workgroup, for 1 .. many workgroups; time per launch from events over a chain of
// launches, and the phases of workgroup 0 from the device clock (100 MHz).
// build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -I russell_amd/csrc/rt_hip -I russell_amd/csrc tools/microbench/tile_bench.hip -o tools/microbench/tile_bench
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <vector>

#include "kernels_factor_binv.hpp"

using namespace hipmf;

#define CK(x)                                                                  \
    do {                                                                       \
        hipError_t e_ = (x);                                                   \
        if (e_ != hipSuccess) {                                                \
            printf("%s: %s\n", #x, hipGetErrorString(e_));                     \
            exit(1);                                                           \
        }                                                                      \
    } while (0)

__global__ void __launch_bounds__(64) k_empty(unsigned long long *st) {
    if (blockIdx.x == 0 && threadIdx.x == 0) st[0] = dev_clock();
}

// MODE 0: tile_lu32, 1: tile_inv32 (unrolled), 2: tile_inv32_rot
template <int MODE> __global__ void __launch_bounds__(64) k_tile(const double *__restrict__ A, double *__restrict__ out, unsigned long long *st) {
    __shared__ int32_t rk[NB];
    const int tid = threadIdx.x;
    const double *F = A + (size_t)blockIdx.x * NB * NB;
    unsigned long long t0 = dev_clock();
    double a[NB];
#pragma unroll
    for (int c = 0; c < NB; c++) a[c] = tid < NB ? F[tid + c * NB] : 0.0;
    unsigned long long t1 = dev_clock();
    int step = 0, npert = 0, nzero = 0;
    double dval = 1.0;
    // (the clock is read after the last loaded value was used: a[31] feeds the first instruction below)
    if (MODE == 0) tile_lu32<true>(a, tid, 1e-13, 1e-13, step, npert, nzero);
    else if (MODE == 1) {
        double zr, zi;
        tile_inv32(a, tid, NB, 1e-13, 1e-13, step, dval, rk, npert, nzero, zr, zi);
    }
    else tile_inv32_rot(a, tid, 1e-13, 1e-13, step, dval, rk, npert, nzero);
    unsigned long long t2 = dev_clock();
    if (tid < NB) {
        double *o = out + (size_t)blockIdx.x * NB * NB;
#pragma unroll
        for (int c = 0; c < NB; c++) o[step + c * NB] = a[c];
    }
    unsigned long long t3 = dev_clock();
    if (blockIdx.x == 0 && tid == 0) st[0] = t0, st[1] = t1, st[2] = t2, st[3] = t3, st[4] = (unsigned long long)(npert + nzero + (int)dval);
}

template <typename L> static double chain_us(L launch, int reps) {
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    for (int i = 0; i < 5; i++) launch();
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(e0, 0));
    for (int i = 0; i < reps; i++) launch();
    CK(hipEventRecord(e1, 0));
    CK(hipEventSynchronize(e1));
    float ms = 0;
    CK(hipEventElapsedTime(&ms, e0, e1));
    return 1e3 * ms / reps;
}

int main() {
    const int maxwg = 4096;
    std::vector<double> h((size_t)maxwg * NB * NB);
    srand(7);
    for (int s = 0; s < maxwg; s++)
        for (int c = 0; c < NB; c++)
            for (int r = 0; r < NB; r++) h[(size_t)s * NB * NB + r + c * NB] = (rand() / (double)RAND_MAX - 0.5) + (r == c ? 4.0 : 0.0);
    double *A, *out;
    unsigned long long *st;
    CK(hipMalloc(&A, sizeof(double) * h.size()));
    CK(hipMalloc(&out, sizeof(double) * h.size()));
    CK(hipMalloc(&st, 64));
    CK(hipMemcpy(A, h.data(), sizeof(double) * h.size(), hipMemcpyHostToDevice));
    printf("empty launch chain: %.2f us per launch\n", chain_us([&] { hipLaunchKernelGGL(k_empty, dim3(1), dim3(64), 0, 0, st); }, 200));
    const char *names[3] = {"tile_lu32 (unrolled LU)", "tile_inv32 (unrolled Gauss-Jordan)", "tile_inv32_rot (loop Gauss-Jordan)"};
    for (int mode = 0; mode < 3; mode++)
        for (int nwg : {1, 64, 1024, 4096}) {
            auto launch = [&] {
                if (mode == 0) hipLaunchKernelGGL(k_tile<0>, dim3(nwg), dim3(64), 0, 0, A, out, st);
                else if (mode == 1) hipLaunchKernelGGL(k_tile<1>, dim3(nwg), dim3(64), 0, 0, A, out, st);
                else hipLaunchKernelGGL(k_tile<2>, dim3(nwg), dim3(64), 0, 0, A, out, st);
            };
            const double us = chain_us(launch, 100);
            unsigned long long hs[5];
            CK(hipMemcpy(hs, st, 40, hipMemcpyDeviceToHost));
            printf("%-36s %5d workgroups: %7.2f us per launch; workgroup 0: load %.2f us, tile %.2f us, store %.2f us\n", names[mode], nwg, us,
                   (hs[1] - hs[0]) * 0.01, (hs[2] - hs[1]) * 0.01, (hs[3] - hs[2]) * 0.01);
        }
    return 0;
}
