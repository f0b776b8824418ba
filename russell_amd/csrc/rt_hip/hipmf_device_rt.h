// Device runtime glue for gfx950 (the product build).  Kernels include <hipmf_device_rt.h>;
// the development-only CPU emulator supplies a header of the same name under tools/hipemu/.
#pragma once
#include <hip/hip_runtime.h>

typedef double f64x4 __attribute__((ext_vector_type(4)));

// v_mfma_f64_16x16x4_f64: D(16x16) = A(16x4) * B(4x16) + C.
//   A operand: lane l holds A[l & 15][l >> 4];   B operand: lane l holds B[l >> 4][l & 15]
//   C/D: lane l, register g holds D[(l >> 4) + 4 g][l & 15]
__device__ __forceinline__ f64x4 mfma_f64_16x16x4(double a, double b, f64x4 c) {
    return __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c, 0, 0, 0);
}

#define HIPMF_DYN_SHARED(T, name) extern __shared__ __attribute__((aligned(16))) T name[]

// broadcast of lane `src` (wave-uniform) to every lane: two v_readlane_b32, result lives in SGPRs
__device__ __forceinline__ double wave_bcast(double v, int src) {
    int lo = __double2loint(v), hi = __double2hiint(v);
    lo = __builtin_amdgcn_readlane(lo, src);
    hi = __builtin_amdgcn_readlane(hi, src);
    return __hiloint2double(hi, lo);
}
