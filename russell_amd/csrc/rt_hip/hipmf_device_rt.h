// Device runtime glue for gfx950 (the product build).  Kernels include <hipmf_device_rt.h>;
// the development-only CPU emulator supplies a header of the same name under tools/hipemu/.
#pragma once
#include <hip/hip_runtime.h>

#define HIPMF_HAVE_RCCL 1 // the product build binds RCCL (lazily, dlopen) for the multi-GPU entry points

typedef double f64x4 __attribute__((ext_vector_type(4)));

// v_mfma_f64_16x16x4_f64: D(16x16) = A(16x4) * B(4x16) + C.
//   A operand: lane l holds A[l & 15][l >> 4];   B operand: lane l holds B[l >> 4][l & 15]
//   C/D: lane l, register g holds D[(l >> 4) + 4 g][l & 15]
__device__ __forceinline__ f64x4 mfma_f64_16x16x4(double a, double b, f64x4 c) {
    return __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c, 0, 0, 0);
}

// 16-byte accesses: global loads from 8-byte (4-byte) aligned addresses (global_load_dwordx4 takes them), LDS stores to 16-byte aligned ones
typedef double f64x2 __attribute__((ext_vector_type(2)));
typedef int i32x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ f64x2 ld_f64x2(const double *p) {
    typedef double f64x2u __attribute__((ext_vector_type(2), aligned(8)));
    return *(const f64x2u *)p;
}
__device__ __forceinline__ i32x4 ld_i32x4(const int *p) {
    typedef int i32x4u __attribute__((ext_vector_type(4), aligned(4)));
    return *(const i32x4u *)p;
}
__device__ __forceinline__ void st_lds_f64x2(double *p, f64x2 v) { *(f64x2 *)p = v; }
__device__ __forceinline__ void st_lds_i32x4(int *p, i32x4 v) { *(i32x4 *)p = v; }

// nothing is scheduled across this point (keeps the loads of an unrolled loop's later iterations from being hoisted above the earlier
// ones: register pressure)
#define HIPMF_SCHED_BARRIER() __builtin_amdgcn_sched_barrier(0)
// the value is needed (in a scalar register) at this point of the program: its load cannot sink below
#define HIPMF_KEEP_SCALAR(x) asm volatile("" ::"s"(x))
#define HIPMF_DYN_SHARED(T, name) extern __shared__ __attribute__((aligned(16))) T name[]
// a kernel that asks for more than 64 KB of dynamic LDS (gfx950 has 160 KB per workgroup) announces it once
#define HIPMF_ALLOW_LDS(kernel, bytes) (void)hipFuncSetAttribute(reinterpret_cast<const void *>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)(bytes))

// broadcast of lane `src` (wave-uniform) to every lane: two v_readlane_b32, result lives in SGPRs
__device__ __forceinline__ double wave_bcast(double v, int src) {
    int lo = __double2loint(v), hi = __double2hiint(v);
    lo = __builtin_amdgcn_readlane(lo, src);
    hi = __builtin_amdgcn_readlane(hi, src);
    return __hiloint2double(hi, lo);
}

__device__ __forceinline__ int wave_bcast_i32(int v, int src) { return __builtin_amdgcn_readlane(v, src); }
__device__ __forceinline__ long long wave_bcast_i64(long long v, int src) {
    const int lo = __builtin_amdgcn_readlane((int)(unsigned)(unsigned long long)v, src);
    const int hi = __builtin_amdgcn_readlane((int)(unsigned)((unsigned long long)v >> 32), src);
    return (long long)(((unsigned long long)(unsigned)hi << 32) | (unsigned)lo);
}
// tells the compiler that a value equal on all lanes is wave-uniform (moves it to an SGPR)
__device__ __forceinline__ int wave_uniform(int v) { return __builtin_amdgcn_readfirstlane(v); }

// wave-wide maximum of a 64-bit key (result wave-uniform).  16-lane rows are reduced with DPP
// (quad_perm xor 1, xor 2, row_half_mirror, row_mirror: no LDS crossbar round trips), the four row
// maxima are combined through v_readlane.
__device__ __forceinline__ unsigned long long wave_max_u64(unsigned long long key) {
#define HIPMF_DPP_MAX_STEP(ctrl)                                                         \
    {                                                                                    \
        int lo = (int)(unsigned)key, hi = (int)(unsigned)(key >> 32);                    \
        int olo = __builtin_amdgcn_update_dpp(lo, lo, ctrl, 0xf, 0xf, false);            \
        int ohi = __builtin_amdgcn_update_dpp(hi, hi, ctrl, 0xf, 0xf, false);            \
        unsigned long long o = ((unsigned long long)(unsigned)ohi << 32) | (unsigned)olo; \
        key = o > key ? o : key;                                                         \
    }
    HIPMF_DPP_MAX_STEP(0xB1)  // quad_perm [1,0,3,2]
    HIPMF_DPP_MAX_STEP(0x4E)  // quad_perm [2,3,0,1]
    HIPMF_DPP_MAX_STEP(0x141) // row_half_mirror
    HIPMF_DPP_MAX_STEP(0x140) // row_mirror
#undef HIPMF_DPP_MAX_STEP
    unsigned long long best = 0;
#pragma unroll
    for (int row = 0; row < 4; row++) {
        unsigned lo = (unsigned)__builtin_amdgcn_readlane((int)(unsigned)key, row * 16);
        unsigned hi = (unsigned)__builtin_amdgcn_readlane((int)(unsigned)(key >> 32), row * 16);
        unsigned long long k = ((unsigned long long)hi << 32) | lo;
        best = k > best ? k : best;
    }
    return best;
}

// wave-wide maximum of a 32-bit key (result wave-uniform): four DPP max steps per 16-lane row, then the
// four row maxima through v_readlane.
template <int ROWS = 4> __device__ __forceinline__ unsigned wave_max_u32(unsigned key) { // ROWS: 16-lane rows that can hold a candidate
    int k = (int)key;
    // one VOP2-with-DPP instruction per stage: k = max(k from the partner lane, k).  (Through __builtin_amdgcn_update_dpp the compiler emits
    // v_mov, s_nop, v_mov_dpp, v_max per stage; this reduction sits in the pivot search of every step of the register tile LU, the longest
    // sequential piece of a tiled step.  s_nop 1: a VALU result needs two wait states before a DPP read of it.)
#define HIPMF_DPP_MAX32(ctrl) asm volatile("s_nop 1\n\tv_max_u32_dpp %0, %0, %0 " ctrl " row_mask:0xf bank_mask:0xf" : "+v"(k));
    HIPMF_DPP_MAX32("quad_perm:[1,0,3,2]")
    HIPMF_DPP_MAX32("quad_perm:[2,3,0,1]")
    HIPMF_DPP_MAX32("row_half_mirror")
    HIPMF_DPP_MAX32("row_mirror")
#undef HIPMF_DPP_MAX32
    unsigned best = 0;
#pragma unroll
    for (int row = 0; row < ROWS; row++) {
        unsigned v = (unsigned)__builtin_amdgcn_readlane(k, row * 16);
        best = v > best ? v : best;
    }
    return best;
}

// ---- in-launch hand-off between workgroups (dependency-driven solve kernels) ----
// The 8 XCDs have private L2s and every CU a private L1, so data exchanged INSIDE a launch is written
// write-through and read around the L1 with agent-scope (sc1) accesses; flags are agent-scope counters.
__device__ __forceinline__ double ld_agent(const double *p) {
    return __longlong_as_double((long long)__hip_atomic_load((const unsigned long long *)p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
}
__device__ __forceinline__ void st_agent(double *p, double v) {
    __hip_atomic_store((unsigned long long *)p, (unsigned long long)__double_as_longlong(v), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ int flag_load(const int *p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ int flag_add(int *p, int v) { return __hip_atomic_fetch_add(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ void flag_store(int *p, int v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
// every wave that stored hand-off data drains its stores before the flag is raised (inline asm: the compiler
// must not drop or move this wait)
__device__ __forceinline__ void drain_stores() { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }
__device__ __forceinline__ void poll_nap() { __builtin_amdgcn_s_sleep(2); }
// reciprocal without the IEEE division sequence (v_div_scale / v_div_fmas / v_div_fixup): v_rcp_f64 and two Newton steps, accurate to
// about an ulp for normal arguments; 0 gives inf, as 1.0 / 0.0 does (callers test the pivot itself, not its reciprocal)
__device__ __forceinline__ double fast_rcp(double x) {
    double r = __builtin_amdgcn_rcp(x);
    r = __builtin_fma(__builtin_fma(-x, r, 1.0), r, r);
    r = __builtin_fma(__builtin_fma(-x, r, 1.0), r, r);
    return r;
}
// LDS written by some lanes of a wavefront becomes readable by its other lanes (single-wave phases of
// multi-wave workgroups: no s_barrier, the wave's LDS operations complete in order)
__device__ __forceinline__ void wave_sync() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// constant-rate device clock (100 MHz), for the optional per-task trace of the dependency-driven solve
__device__ __forceinline__ unsigned long long dev_clock() { return wall_clock64(); }
