// Ceiling of the FP64 matrix pipe on this GPU, and why it is not the data-sheet number.  Every wave issues back-to-back MFMAs on NACC
// independent accumulators from registers (no memory traffic); the kernel also reads the shader-cycle counter (s_memtime) and the
// constant 100 MHz counter around the loop, which separates the two possible explanations of a low rate:
//   cycles per MFMA and SIMD  (64 for v_mfma_f64_16x16x4_f64 if the pipe runs at the advertised 16 passes x 4 cycles)
//   effective shader clock    (cycles / wall time; the chip clocks to its power budget under FP64 matrix load)
// Variants: v_mfma_f64_16x16x4_f64, v_mfma_f64_4x4x4_f64 (4 blocks), plain v_fma_f64 (the vector rate the data sheet equates
// with the matrix rate), s_setprio 3 around the loop.  Build and run on the GPU box:
//   hipcc --offload-arch=gfx950 -O3 tools/microbench/mfma_peak.hip -o /tmp/mfma_peak && /tmp/mfma_peak
#include <hip/hip_runtime.h>

#include <cstdio>
#include <vector>

typedef double f64x4 __attribute__((ext_vector_type(4)));

struct Stamp {
    unsigned long long cyc, wall;
};

template <int NACC, int KIND, bool PRIO>
__global__ void __launch_bounds__(256) k_probe(double *out, Stamp *st, int iters, double a0, double b0) {
    f64x4 acc[NACC];
    for (int i = 0; i < NACC; i++) acc[i] = f64x4{0.0, 0.0, 0.0, 0.0};
    double a = a0 + threadIdx.x * 1e-9, b = b0 - threadIdx.x * 1e-9;
    if (PRIO) __builtin_amdgcn_s_setprio(3);
    const unsigned long long c0 = __builtin_amdgcn_s_memtime(), w0 = wall_clock64();
    for (int it = 0; it < iters; it++) {
#pragma unroll
        for (int i = 0; i < NACC; i++) {
            if (KIND == 0) acc[i] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[i], 0, 0, 0);
            else if (KIND == 1) acc[i][0] = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, acc[i][0], 0, 0, 0);
            else {
#pragma unroll
                for (int q = 0; q < 4; q++) acc[i][q] = __builtin_fma(a, b, acc[i][q]);
            }
        }
    }
    const unsigned long long c1 = __builtin_amdgcn_s_memtime(), w1 = wall_clock64();
    double s = 0.0;
    for (int i = 0; i < NACC; i++) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
    out[blockIdx.x * 256 + threadIdx.x] = s;
    if ((threadIdx.x & 63) == 0) st[blockIdx.x * 4 + (threadIdx.x >> 6)] = {c1 - c0, w1 - w0};
}

template <int NACC, int KIND, bool PRIO>
void run(int blocks, int iters, double *d, Stamp *dst) {
    hipEvent_t e0, e1;
    hipEventCreate(&e0), hipEventCreate(&e1);
    hipLaunchKernelGGL((k_probe<NACC, KIND, PRIO>), dim3(blocks), dim3(256), 0, 0, d, dst, iters, 1.0, 1.0);
    hipEventRecord(e0);
    hipLaunchKernelGGL((k_probe<NACC, KIND, PRIO>), dim3(blocks), dim3(256), 0, 0, d, dst, iters, 1.0, 1.0);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms = 0;
    hipEventElapsedTime(&ms, e0, e1);
    std::vector<Stamp> st((size_t)blocks * 4);
    hipMemcpy(st.data(), dst, sizeof(Stamp) * st.size(), hipMemcpyDeviceToHost);
    double cyc = 0, wall = 0;
    for (const Stamp &s : st) cyc += (double)s.cyc, wall += (double)s.wall;
    cyc /= st.size(), wall /= st.size();
    // flops per instruction and wave: 16x16x4: 2048; 4x4x4 (4 blocks): 512; v_fma_f64 x 4: 4 x 128
    const double fl = KIND == 0 ? 2048.0 : 512.0;
    const double flops = (double)blocks * 4 * iters * NACC * fl;
    const double waves_per_simd = blocks / 256.0; // 4 waves per block on the 4 SIMDs of a CU
    const double ops = (double)iters * NACC * (KIND == 2 ? 4.0 : 1.0);
    printf("%-24s%s blocks/CU=%.0f acc=%d: %7.2f ms %6.1f TFLOP/s | per wave: %.1f cycles per op -> %.1f per SIMD op slot, clock %.2f GHz\n",
           KIND == 0 ? "v_mfma_f64_16x16x4_f64" : (KIND == 1 ? "v_mfma_f64_4x4x4_f64" : "v_fma_f64"), PRIO ? " prio3" : "      ", blocks / 256.0, NACC,
           ms, flops / (ms * 1e-3) / 1e12, cyc / ops, cyc / ops / waves_per_simd, cyc / (wall * 10.0));
}

int main() {
    double *d;
    Stamp *dst;
    if (hipMalloc(&d, sizeof(double) * 256 * 4096) != hipSuccess || hipMalloc(&dst, sizeof(Stamp) * 4 * 4096) != hipSuccess) return 1;
    for (int blocks : {256, 512, 1024}) {
        run<1, 0, false>(blocks, 20000, d, dst);
        run<4, 0, false>(blocks, 20000, d, dst);
        run<8, 0, false>(blocks, 10000, d, dst);
        run<16, 0, false>(blocks, 5000, d, dst);
        run<8, 0, true>(blocks, 10000, d, dst);
    }
    for (int blocks : {256, 1024}) {
        run<8, 1, false>(blocks, 20000, d, dst);
        run<8, 2, false>(blocks, 20000, d, dst);
    }
    return 0;
}
