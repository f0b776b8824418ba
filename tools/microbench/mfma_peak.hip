// Ceiling of v_mfma_f64_16x16x4_f64 on this GPU: every wave issues back-to-back MFMAs on NACC independent accumulators from
// registers (no memory traffic).  Build and run on the GPU box:
//   hipcc --offload-arch=gfx950 -O3 tools/microbench/mfma_peak.hip -o /tmp/mfma_peak && /tmp/mfma_peak
#include <hip/hip_runtime.h>

#include <cstdio>

typedef double f64x4 __attribute__((ext_vector_type(4)));

template <int NACC>
__global__ void __launch_bounds__(256) k_mfma(double *out, int iters, double a0, double b0) {
    f64x4 acc[NACC];
    for (int i = 0; i < NACC; i++) acc[i] = f64x4{0.0, 0.0, 0.0, 0.0};
    double a = a0 + threadIdx.x * 1e-9, b = b0 - threadIdx.x * 1e-9;
    for (int it = 0; it < iters; it++) {
#pragma unroll
        for (int i = 0; i < NACC; i++) acc[i] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[i], 0, 0, 0);
    }
    double s = 0.0;
    for (int i = 0; i < NACC; i++) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}

template <int NACC>
void run(int blocks, int iters, double *d) {
    hipEvent_t e0, e1;
    hipEventCreate(&e0), hipEventCreate(&e1);
    hipLaunchKernelGGL(k_mfma<NACC>, dim3(blocks), dim3(256), 0, 0, d, iters, 1.0, 1.0);
    hipEventRecord(e0);
    hipLaunchKernelGGL(k_mfma<NACC>, dim3(blocks), dim3(256), 0, 0, d, iters, 1.0, 1.0);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms = 0;
    hipEventElapsedTime(&ms, e0, e1);
    const double flops = (double)blocks * 4 /*waves*/ * iters * NACC * 2048.0;
    printf("blocks=%5d (%.0f per CU)  independent accumulators per wave=%d: %.2f ms, %.1f TFLOP/s\n", blocks, blocks / 256.0, NACC, ms,
           flops / (ms * 1e-3) / 1e12);
}

int main() {
    double *d;
    if (hipMalloc(&d, sizeof(double) * 256 * 4096) != hipSuccess) return 1;
    for (int blocks : {256, 512, 1024}) {
        run<1>(blocks, 20000, d);
        run<2>(blocks, 20000, d);
        run<4>(blocks, 20000, d);
        run<8>(blocks, 10000, d);
    }
    return 0;
}
