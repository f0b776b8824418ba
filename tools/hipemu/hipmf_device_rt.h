// CPU-emulator twin of russell_amd/csrc/rt_hip/hipmf_device_rt.h (development tool, see
// tools/hipemu/hip/hip_runtime.h).
#pragma once
#include <hip/hip_runtime.h>

#define HIPMF_EMULATED 1 // (no graph capture, no RCCL: the emulator runs launches synchronously)

typedef double f64x4 __attribute__((vector_size(32)));

struct f64x2 {
    double x, y;
};
struct i32x4 {
    int x, y, z, w;
};
inline f64x2 ld_f64x2(const double *p) { return {p[0], p[1]}; }
inline i32x4 ld_i32x4(const int *p) { return {p[0], p[1], p[2], p[3]}; }
inline void st_lds_f64x2(double *p, f64x2 v) { p[0] = v.x, p[1] = v.y; }
inline void st_lds_i32x4(int *p, i32x4 v) { p[0] = v.x, p[1] = v.y, p[2] = v.z, p[3] = v.w; }

inline double hipemu_mfma_a[16][64], hipemu_mfma_b[16][64];

// same operand / result lane maps as v_mfma_f64_16x16x4_f64
inline f64x4 mfma_f64_16x16x4(double a, double b, f64x4 c) {
    int lin = hipemu::linear_tid(), lane = lin & 63, wave = lin >> 6;
    hipemu_mfma_a[wave][lane] = a;
    hipemu_mfma_b[wave][lane] = b;
    hipemu::sync_wave();
    int col = lane & 15;
    for (int g = 0; g < 4; g++) {
        int row = (lane >> 4) + 4 * g;
        double s = c[g];
        for (int k = 0; k < 4; k++) s = std::fma(hipemu_mfma_a[wave][row + 16 * k], hipemu_mfma_b[wave][col + 16 * k], s);
        c[g] = s;
    }
    hipemu::sync_wave();
    return c;
}

#define HIPMF_SCHED_BARRIER() ((void)0)
#define HIPMF_KEEP_SCALAR(x) ((void)(x))
#define HIPMF_ALLOW_LDS(kernel, bytes) ((void)0)
#define HIPMF_DYN_SHARED(T, name) T *name = (T *)(((uintptr_t)hipemu::g_dynshared.data() + 15) & ~(uintptr_t)15)

inline double wave_bcast(double v, int src) { return __shfl(v, src); }

inline int wave_bcast_i32(int v, int src) { return __shfl(v, src); }
inline long long wave_bcast_i64(long long v, int src) { return __shfl(v, src); }
inline int wave_uniform(int v) { return v; }

inline unsigned long long wave_max_u64(unsigned long long key) {
    for (int off = 32; off > 0; off >>= 1) {
        unsigned long long o = __shfl_xor(key, off);
        key = o > key ? o : key;
    }
    return key;
}

template <int ROWS = 4> inline unsigned wave_max_u32(unsigned key) {
    for (int off = 32; off > 0; off >>= 1) {
        unsigned o = __shfl_xor(key, off);
        key = o > key ? o : key;
    }
    return key;
}
inline unsigned __float_as_uint(float f) {
    unsigned u;
    memcpy(&u, &f, 4);
    return u;
}

// in-launch hand-off primitives (the emulator runs workgroups one after another in index order, so a
// dependency is always satisfied before its consumer starts; these keep the protocol's code path alive)
inline double ld_agent(const double *p) { return *(const volatile double *)p; }
inline void st_agent(double *p, double v) { *(volatile double *)p = v; }
inline int flag_load(const int *p) { return *(const volatile int *)p; }
inline int flag_add(int *p, int v) {
    int o = *p;
    *p = o + v;
    return o;
}
inline void flag_store(int *p, int v) { *(volatile int *)p = v; }
inline void drain_stores() {}
inline void poll_nap() {}
inline double fast_rcp(double x) { return 1.0 / x; }
inline void wave_sync() { hipemu::sync_wave(); }
inline unsigned long long dev_clock() { return 0; }
