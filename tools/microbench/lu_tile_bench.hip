// Micro-benchmark of the register-resident 32x32 tile LU (tile_lu32): cycles for the first (cold
// instruction cache) and for later (warm) executions inside one wave.  Build and run on the GPU box:
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -I russell_amd/csrc/rt_hip -I russell_amd/csrc tools/microbench/lu_tile_bench.hip -o /tmp/lu_bench && /tmp/lu_bench
#include <hipmf_device_rt.h>

#include <cstdio>
#include <vector>

#include "kernels_common.hpp"
#include "kernels_factor.hpp"

using namespace hipmf;

__global__ void bench_lu(const double *in, double *out, long long *cycles, int reps) {
    __shared__ double T[NB][NB + 1];
    const int tid = threadIdx.x;
    for (int e = tid; e < NB * NB; e += 64) T[e % NB][e / NB] = in[e];
    __syncthreads();
    double acc = 0.0;
    for (int it = 0; it < reps; it++) {
        double a[NB];
#pragma unroll
        for (int c = 0; c < NB; c++) a[c] = (tid < NB) ? T[tid][c] + 1e-9 * it : 0.0;
        long long t0 = clock64();
        int step, npert, nzero;
        tile_lu32(a, tid, 1e-300, 1e-300, step, npert, nzero);
        long long t1 = clock64();
        if (tid == 0) cycles[it] = t1 - t0;
#pragma unroll
        for (int c = 0; c < NB; c++) acc += a[c];
        acc += step;
    }
    out[blockIdx.x * 64 + tid] = acc;
}

int main() {
    std::vector<double> h(NB * NB);
    for (int i = 0; i < NB * NB; i++) h[i] = ((i * 7919) % 1000) / 1000.0 + ((i % 33 == 0) ? 4.0 : 0.0);
    double *din, *dout;
    long long *dc;
    const int reps = 8;
    hipMalloc(&din, sizeof(double) * NB * NB);
    hipMalloc(&dout, sizeof(double) * 64 * 256);
    hipMalloc(&dc, sizeof(long long) * reps);
    hipMemcpy(din, h.data(), sizeof(double) * NB * NB, hipMemcpyHostToDevice);
    for (int blocks : {1, 64}) {
        for (int trial = 0; trial < 2; trial++) {
            hipLaunchKernelGGL(bench_lu, dim3(blocks), dim3(64), 0, 0, din, dout, dc, reps);
            hipDeviceSynchronize();
            long long hc[reps];
            hipMemcpy(hc, dc, sizeof(hc), hipMemcpyDeviceToHost);
            printf("blocks=%d trial=%d cycles per tile_lu32:", blocks, trial);
            for (int i = 0; i < reps; i++) printf(" %lld", hc[i]);
            printf("\n");
        }
    }
    hipEvent_t e0, e1;
    hipEventCreate(&e0), hipEventCreate(&e1);
    hipEventRecord(e0);
    for (int i = 0; i < 200; i++) hipLaunchKernelGGL(bench_lu, dim3(1), dim3(64), 0, 0, din, dout, dc, 1);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    printf("200 back-to-back single-LU launches: %.2f us each\n", ms * 1e3 / 200);
    return 0;
}
