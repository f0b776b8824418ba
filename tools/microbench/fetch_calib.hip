// fetch_calib.hip -- what rocprofv3's FETCH_SIZE reports for the access shapes of the triangular-solve kernels, against a known byte count.
// MI355X_MICROARCH.md: "on gfx950 FETCH_SIZE reports exactly 1/2 of the bytes of a wide coalesced streaming read (16 B/lane) ...
// other access widths are uncalibrated: calibrate on a known byte count in your own access pattern".  Every kernel below reads a
// 1 GiB buffer exactly once (the sums go to a dummy output); run under `rocprofv3 --pmc FETCH_SIZE --kernel-trace` and divide
// (tools/gpu/r04_calib.sh prints counter / true bytes per kernel).
//   k_wide16        16 B per lane, consecutive lanes consecutive (the guide's reference shape)
//   k_flat8         8 B per lane, a wavefront reads flat 512-byte pieces (the wave-subtree kernels' factor stream)
//   k_slab64        8 B per lane, 8 lanes = one 64-byte segment of a column, 8 columns per load: an 8-row slab of a column-major
//                   panel whose rows start on a 64-byte boundary (top-level slabs of E / E')
//   k_slab64_off32  the same, the slab's rows start 32 bytes into a 64-byte line (every segment straddles two 64-byte halves)
//   k_slab128       16-row slabs (128-byte segments, aligned)
// build: hipcc --offload-arch=gfx950 -O3 tools/microbench/fetch_calib.hip -o tools/microbench/fetch_calib
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>

#define CK(x)                                                   \
    do {                                                        \
        hipError_t e_ = (x);                                    \
        if (e_ != hipSuccess) {                                 \
            printf("%s: %s\n", #x, hipGetErrorString(e_));      \
            exit(1);                                            \
        }                                                       \
    } while (0)

typedef double f64x2 __attribute__((ext_vector_type(2)));

__global__ void __launch_bounds__(256) k_wide16(const f64x2 *__restrict__ p, size_t n16, double *out) {
    double s = 0.0;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n16; i += (size_t)gridDim.x * 256) {
        const f64x2 v = p[i];
        s += v.x + v.y;
    }
    if (s == 123.456) out[0] = s;
}

__global__ void __launch_bounds__(256) k_flat8(const double *__restrict__ p, size_t n8, double *out) {
    double s = 0.0;
    // four 512-byte pieces per wavefront in flight
    const size_t wave = ((size_t)blockIdx.x * 256 + threadIdx.x) >> 6, lane = threadIdx.x & 63, nw = ((size_t)gridDim.x * 256) >> 6;
    for (size_t b = wave * 256; b + 256 <= n8; b += nw * 256) {
        const double a0 = p[b + lane], a1 = p[b + 64 + lane], a2 = p[b + 128 + lane], a3 = p[b + 192 + lane];
        s += (a0 + a1) + (a2 + a3);
    }
    if (s == 123.456) out[0] = s;
}

// column-major panel with `rows` rows (ld = rows + pad) and `cols` columns: workgroup = one slab of SR rows across all columns;
// lane = (row of the slab, column group): a load instruction of a wavefront touches 64 / SR columns
template <int SR> __global__ void __launch_bounds__(256) k_slab(const double *__restrict__ p, int rows, int ld, int cols, int off, double *out) {
    constexpr int G = 256 / SR; // column groups per workgroup
    const int r = threadIdx.x % SR, g = threadIdx.x / SR;
    double s = 0.0;
    for (int slab = blockIdx.x; slab * SR < rows; slab += gridDim.x) {
        const double *q = p + off + (size_t)slab * SR + r;
        for (int c = g; c + 3 * G < cols; c += 4 * G) {
            const double a0 = q[(size_t)c * ld], a1 = q[(size_t)(c + G) * ld], a2 = q[(size_t)(c + 2 * G) * ld], a3 = q[(size_t)(c + 3 * G) * ld];
            s += (a0 + a1) + (a2 + a3);
        }
    }
    if (s == 123.456) out[0] = s;
}

int main() {
    const size_t bytes = 1ull << 30;
    double *buf, *out;
    CK(hipMalloc(&buf, bytes + 4096));
    CK(hipMalloc(&out, 64));
    CK(hipMemset(buf, 0, bytes + 4096));
    const int grid = 256 * 8;
    for (int rep = 0; rep < 2; rep++) {
        hipLaunchKernelGGL(k_wide16, dim3(grid), dim3(256), 0, 0, (const f64x2 *)buf, bytes / 16, out);
        hipLaunchKernelGGL(k_flat8, dim3(grid), dim3(256), 0, 0, buf, bytes / 8, out);
        // panel of 4096 rows x 32768 columns = 1 GiB (ld = rows: a column is 32 KB; every slab row starts on a 64-byte line)
        const int rows = 4096, cols = 32768;
        hipLaunchKernelGGL(k_slab<8>, dim3(rows / 8), dim3(256), 0, 0, buf, rows, rows, cols, 0, out);
        hipLaunchKernelGGL(k_slab<8>, dim3(rows / 8), dim3(256), 0, 0, buf, rows - 8, rows, cols, 4, out); // (32 bytes into the line; one slab less)
        hipLaunchKernelGGL(k_slab<16>, dim3(rows / 16), dim3(256), 0, 0, buf, rows, rows, cols, 0, out);
        CK(hipDeviceSynchronize());
    }
    printf("true bytes per launch: k_wide16 %zu, k_flat8 %zu, k_slab<8> %zu, k_slab<8>+32B %zu, k_slab<16> %zu\n", bytes, bytes, bytes,
           (size_t)(4096 - 8) * 32768 * 8, bytes);
    return 0;
}
