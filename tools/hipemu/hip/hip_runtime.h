// hipemu -- a tiny single-process HIP emulator used ONLY to debug kernel logic on a machine
// without a GPU (the authoring container).  It is a development/test tool:
//   * it is never compiled into the product library (russell_amd/lib/librussell_hipmf.so);
//   * results produced through it are not parity evidence -- parity is measured on a real MI355X.
// Every GPU thread of a block is a ucontext fiber; __syncthreads() and the wave-level helpers
// yield to a round-robin scheduler, so barrier semantics (and barrier bugs) are reproduced.
// Blocks run one after another; streams and events are synchronous.
#pragma once
#include <ucontext.h>
#include <sys/mman.h>

#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <functional>
#include <unordered_map>
#include <vector>

struct dim3 {
    unsigned x, y, z;
    dim3(unsigned a = 1, unsigned b = 1, unsigned c = 1) : x(a), y(b), z(c) {}
};
struct hipemu_uint3 {
    unsigned x, y, z;
};

typedef int hipError_t;
typedef void *hipStream_t;
struct hipemu_event {
    std::chrono::steady_clock::time_point t;
};
typedef hipemu_event *hipEvent_t;
enum { hipSuccess = 0, hipErrorOutOfMemory = 2, hipErrorInvalidValue = 1 };
enum hipMemcpyKind { hipMemcpyHostToDevice = 1, hipMemcpyDeviceToHost = 2, hipMemcpyDeviceToDevice = 3, hipMemcpyDefault = 4 };

#define __global__
#define __device__
#define __host__
#define __forceinline__ inline
#define __shared__ static
#define __launch_bounds__(...)

namespace hipemu {
inline hipemu_uint3 g_threadIdx, g_blockIdx;
inline dim3 g_blockDim, g_gridDim;
inline std::vector<char> g_dynshared;

enum { RUNNABLE = 0, WAIT_BLOCK = 1, WAIT_WAVE = 2, DONE = 3 };
// Fiber switch.  glibc's swapcontext saves and restores the signal mask -- one system call per switch, two per yield: a third of the
// wall time of the emulated kernels went to the kernel.  On x86-64 the switch is done by hand (callee-saved registers + stack pointer; the
// fibers neither change the signal mask nor the floating-point control words); other hosts keep ucontext.
#if defined(__x86_64__) && !defined(HIPEMU_UCONTEXT)
#define HIPEMU_FAST_SWITCH 1
extern "C" void hipemu_switch(void **save_sp, void *load_sp);
asm(".text\n"
    ".weak hipemu_switch\n"
    ".type hipemu_switch,@function\n"
    "hipemu_switch:\n"
    "    pushq %rbp\n    pushq %rbx\n    pushq %r12\n    pushq %r13\n    pushq %r14\n    pushq %r15\n"
    "    movq %rsp, (%rdi)\n"
    "    movq %rsi, %rsp\n"
    "    popq %r15\n    popq %r14\n    popq %r13\n    popq %r12\n    popq %rbx\n    popq %rbp\n"
    "    ret\n"
    ".size hipemu_switch, .-hipemu_switch\n");
#endif
struct Fiber {
#ifdef HIPEMU_FAST_SWITCH
    void *sp = nullptr;
#else
    ucontext_t ctx;
#endif
    char *stack = nullptr;
    int state = DONE;
    hipemu_uint3 tid;
};
inline std::vector<Fiber> g_fibers;
#ifdef HIPEMU_FAST_SWITCH
inline void *g_sched_sp = nullptr;
#else
inline ucontext_t g_sched;
#endif
inline int g_cur = 0;
inline std::function<void()> g_body;
inline unsigned long long g_wave_buf[16][64];
constexpr size_t STACK_BYTES = 256 * 1024;

inline int linear_tid() { return (int)(g_threadIdx.x + g_blockDim.x * (g_threadIdx.y + g_blockDim.y * g_threadIdx.z)); }

inline void yield_with(int state) {
    Fiber &f = g_fibers[g_cur];
    f.state = state;
#ifdef HIPEMU_FAST_SWITCH
    hipemu_switch(&f.sp, g_sched_sp);
#else
    swapcontext(&f.ctx, &g_sched);
#endif
    g_threadIdx = f.tid; // restored by the scheduler as well; keep both for clarity
}
inline void sync_block() { yield_with(WAIT_BLOCK); }
inline void sync_wave() { yield_with(WAIT_WAVE); }

inline void fiber_entry() {
    g_body();
    g_fibers[g_cur].state = DONE;
#ifdef HIPEMU_FAST_SWITCH
    hipemu_switch(&g_fibers[g_cur].sp, g_sched_sp);
#else
    swapcontext(&g_fibers[g_cur].ctx, &g_sched);
#endif
}

inline void run_block(int nthreads) {
    if ((int)g_fibers.size() < nthreads) g_fibers.resize(nthreads);
    for (int t = 0; t < nthreads; t++) {
        Fiber &f = g_fibers[t];
        if (!f.stack) f.stack = (char *)malloc(STACK_BYTES);
#ifdef HIPEMU_FAST_SWITCH
        {
            // what hipemu_switch pops: six registers, then the entry point as its return address; the entry then finds the stack pointer
            // 8 bytes below a 16-byte boundary, as after a call (its own return address is never used: fiber_entry does not return)
            void **top = (void **)(((uintptr_t)f.stack + STACK_BYTES) & ~(uintptr_t)15);
            top[-1] = nullptr;
            top[-2] = (void *)fiber_entry;
            for (int r = 3; r <= 8; r++) top[-r] = nullptr;
            f.sp = (void *)(top - 8);
        }
#else
        getcontext(&f.ctx);
        f.ctx.uc_stack.ss_sp = f.stack;
        f.ctx.uc_stack.ss_size = STACK_BYTES;
        f.ctx.uc_link = nullptr;
        makecontext(&f.ctx, (void (*)())fiber_entry, 0);
#endif
        f.state = RUNNABLE;
        f.tid.x = t % g_blockDim.x;
        f.tid.y = (t / g_blockDim.x) % g_blockDim.y;
        f.tid.z = t / (g_blockDim.x * g_blockDim.y);
    }
    int nwaves = (nthreads + 63) / 64;
    for (;;) {
        bool ran = false;
        for (int t = 0; t < nthreads; t++) {
            if (g_fibers[t].state != RUNNABLE) continue;
            g_cur = t;
            g_threadIdx = g_fibers[t].tid;
#ifdef HIPEMU_FAST_SWITCH
            hipemu_switch(&g_sched_sp, g_fibers[t].sp);
#else
            swapcontext(&g_sched, &g_fibers[t].ctx);
#endif
            ran = true;
        }
        int done = 0, wblock = 0;
        for (int t = 0; t < nthreads; t++) {
            done += g_fibers[t].state == DONE;
            wblock += g_fibers[t].state == WAIT_BLOCK;
        }
        if (done == nthreads) break;
        bool released = false;
        if (done + wblock == nthreads) {
            for (int t = 0; t < nthreads; t++)
                if (g_fibers[t].state == WAIT_BLOCK) g_fibers[t].state = RUNNABLE;
            released = true;
        }
        for (int w = 0; w < nwaves; w++) {
            int lo = w * 64, hi = std::min(nthreads, lo + 64), live = 0, ww = 0;
            for (int t = lo; t < hi; t++) {
                live += g_fibers[t].state != DONE;
                ww += g_fibers[t].state == WAIT_WAVE;
            }
            if (live > 0 && ww == live) {
                for (int t = lo; t < hi; t++)
                    if (g_fibers[t].state == WAIT_WAVE) g_fibers[t].state = RUNNABLE;
                released = true;
            }
        }
        if (!ran && !released) {
            fprintf(stderr, "hipemu: deadlock (divergent barrier) in block (%u,%u,%u)\n", g_blockIdx.x, g_blockIdx.y, g_blockIdx.z);
            abort();
        }
    }
}

template <class K, class... Args>
inline void launch(K kernel, dim3 grid, dim3 block, size_t shmem, Args... args) {
    g_gridDim = grid;
    g_blockDim = block;
    if (g_dynshared.size() < shmem + 16) g_dynshared.resize(shmem + 16);
    int nthreads = (int)(block.x * block.y * block.z);
    if (nthreads > 1024) {
        fprintf(stderr, "hipemu: block too large\n");
        abort();
    }
    g_body = [=]() { kernel(args...); };
    for (unsigned z = 0; z < grid.z; z++)
        for (unsigned y = 0; y < grid.y; y++)
            for (unsigned x = 0; x < grid.x; x++) {
                g_blockIdx = {x, y, z};
                run_block(nthreads);
            }
}
} // namespace hipemu

#define threadIdx hipemu::g_threadIdx
#define blockIdx hipemu::g_blockIdx
#define blockDim hipemu::g_blockDim
#define gridDim hipemu::g_gridDim
#define hipLaunchKernelGGL(kernel, grid, block, shmem, stream, ...) hipemu::launch(kernel, dim3(grid), dim3(block), shmem, ##__VA_ARGS__)

inline void __syncthreads() { hipemu::sync_block(); }
inline void __threadfence() {}
inline void __threadfence_block() {}

template <class T>
inline T hipemu_shfl_abs(T v, int src_lane) {
    static_assert(sizeof(T) <= 8, "shfl type");
    int lin = hipemu::linear_tid(), lane = lin & 63, wave = lin >> 6;
    unsigned long long raw = 0;
    memcpy(&raw, &v, sizeof(T));
    hipemu::g_wave_buf[wave][lane] = raw;
    hipemu::sync_wave();
    unsigned long long got = hipemu::g_wave_buf[wave][src_lane & 63];
    hipemu::sync_wave();
    T r;
    memcpy(&r, &got, sizeof(T));
    return r;
}
template <class T>
inline T __shfl(T v, int src, int width = 64) {
    int lane = hipemu::linear_tid() & 63;
    int base = lane & ~(width - 1);
    return hipemu_shfl_abs(v, base + (src & (width - 1)));
}
template <class T>
inline T __shfl_xor(T v, int mask, int width = 64) {
    int lane = hipemu::linear_tid() & 63;
    (void)width;
    return hipemu_shfl_abs(v, lane ^ mask);
}
template <class T>
inline T __shfl_down(T v, unsigned delta, int width = 64) {
    int lane = hipemu::linear_tid() & 63;
    int src = lane + (int)delta;
    if ((src & ~(width - 1)) != (lane & ~(width - 1))) src = lane;
    return hipemu_shfl_abs(v, src);
}
inline int __popcll(unsigned long long v) { return __builtin_popcountll(v); }
inline unsigned long long __ballot(int pred) {
    unsigned long long mine = pred ? 1ull : 0ull, all = 0;
    int lin = hipemu::linear_tid(), lane = lin & 63, wave = lin >> 6;
    hipemu::g_wave_buf[wave][lane] = mine;
    hipemu::sync_wave();
    int lo = wave * 64, hi = std::min<int>((int)(blockDim.x * blockDim.y * blockDim.z), lo + 64);
    for (int t = lo; t < hi; t++)
        if (hipemu::g_fibers[t].state != hipemu::DONE && hipemu::g_wave_buf[wave][t - lo]) all |= 1ull << (t - lo);
    hipemu::sync_wave();
    return all;
}

inline double atomicAdd(double *p, double v) {
    double o = *p;
    *p = o + v;
    return o;
}
inline int atomicAdd(int *p, int v) {
    int o = *p;
    *p = o + v;
    return o;
}
inline unsigned atomicAdd(unsigned *p, unsigned v) {
    unsigned o = *p;
    *p = o + v;
    return o;
}
inline int atomicMax(int *p, int v) {
    int o = *p;
    *p = std::max(o, v);
    return o;
}
inline unsigned long long atomicMin(unsigned long long *p, unsigned long long v) {
    unsigned long long o = *p;
    if (v < o) *p = v;
    return o;
}
inline unsigned long long atomicMax(unsigned long long *p, unsigned long long v) {
    unsigned long long o = *p;
    *p = std::max(o, v);
    return o;
}
inline long long __double_as_longlong(double d) {
    long long r;
    memcpy(&r, &d, 8);
    return r;
}
inline double __longlong_as_double(long long v) {
    double r;
    memcpy(&r, &v, 8);
    return r;
}
inline int atomicOr(int *p, int v) {
    int o = *p;
    *p = o | v;
    return o;
}

// ---- host runtime --------------------------------------------------------------------------------
// (development: an allocation beyond HIPEMU_LAZY_GB -- the pool of a large problem whose PLAN is being timed -- is address space only:
// untouched pages cost nothing, no poison)
inline std::unordered_map<void *, size_t> &hipemu_lazy_blocks() {
    static std::unordered_map<void *, size_t> m;
    return m;
}
inline hipError_t hipMalloc(void **p, size_t bytes) {
    const char *lz = getenv("HIPEMU_LAZY_GB");
    if (lz && bytes > ((size_t)atoi(lz) << 30)) {
        void *q = mmap(nullptr, bytes, PROT_READ | PROT_WRITE, MAP_PRIVATE | MAP_ANONYMOUS | MAP_NORESERVE, -1, 0);
        if (q == MAP_FAILED) return hipErrorOutOfMemory;
        hipemu_lazy_blocks()[q] = bytes;
        *p = q;
        return hipSuccess;
    }
    *p = malloc(bytes ? bytes : 1);
    if (*p) memset(*p, 0xCD, bytes); // poison: catches reads of uninitialised device memory
    return *p ? hipSuccess : hipErrorOutOfMemory;
}
template <class T>
inline hipError_t hipMalloc(T **p, size_t bytes) {
    return hipMalloc((void **)p, bytes);
}
inline hipError_t hipFree(void *p) {
    auto &lz = hipemu_lazy_blocks();
    auto it = lz.find(p);
    if (it != lz.end()) {
        munmap(p, it->second);
        lz.erase(it);
        return hipSuccess;
    }
    free(p);
    return hipSuccess;
}
inline hipError_t hipHostMalloc(void **p, size_t bytes, unsigned = 0) {
    *p = malloc(bytes ? bytes : 1);
    return *p ? hipSuccess : hipErrorOutOfMemory;
}
template <class T>
inline hipError_t hipHostMalloc(T **p, size_t bytes, unsigned f = 0) {
    return hipHostMalloc((void **)p, bytes, f);
}
inline hipError_t hipHostFree(void *p) {
    free(p);
    return hipSuccess;
}
inline hipError_t hipMemcpy(void *d, const void *s, size_t n, hipMemcpyKind) {
    memcpy(d, s, n);
    return hipSuccess;
}
inline hipError_t hipMemcpyAsync(void *d, const void *s, size_t n, hipMemcpyKind, hipStream_t = nullptr) {
    memcpy(d, s, n);
    return hipSuccess;
}
inline hipError_t hipMemset(void *d, int v, size_t n) {
    memset(d, v, n);
    return hipSuccess;
}
inline hipError_t hipMemsetAsync(void *d, int v, size_t n, hipStream_t = nullptr) {
    memset(d, v, n);
    return hipSuccess;
}
constexpr unsigned hipStreamDefault = 0;
inline hipError_t hipStreamCreate(hipStream_t *s);
inline hipError_t hipDeviceGetStreamPriorityRange(int *least, int *greatest) {
    *least = 0, *greatest = 0;
    return hipSuccess;
}
inline hipError_t hipStreamCreateWithPriority(hipStream_t *s, unsigned, int) { return hipStreamCreate(s); }
inline hipError_t hipStreamCreate(hipStream_t *s) {
    *s = nullptr;
    return hipSuccess;
}
inline hipError_t hipStreamDestroy(hipStream_t) { return hipSuccess; }
inline hipError_t hipStreamSynchronize(hipStream_t) { return hipSuccess; }
inline hipError_t hipDeviceSynchronize() { return hipSuccess; }
inline hipError_t hipGetLastError() { return hipSuccess; }
inline hipError_t hipPeekAtLastError() { return hipSuccess; }
inline const char *hipGetErrorString(hipError_t) { return "hipemu"; }
inline hipError_t hipSetDevice(int) { return hipSuccess; }
inline hipError_t hipGetDevice(int *d) {
    *d = 0;
    return hipSuccess;
}
inline hipError_t hipGetDeviceCount(int *c) {
    *c = 1;
    return hipSuccess;
}
inline hipError_t hipMemGetInfo(size_t *fr, size_t *tot) {
    const char *e = getenv("HIPEMU_DEVICE_GB"); // (development: the plan of a large problem can be timed without its pool fitting the host)
    *fr = *tot = (size_t)(e ? atoi(e) : 8) << 30;
    return hipSuccess;
}
inline hipError_t hipEventCreate(hipEvent_t *e) {
    *e = new hipemu_event();
    return hipSuccess;
}
inline hipError_t hipEventDestroy(hipEvent_t e) {
    delete e;
    return hipSuccess;
}
constexpr unsigned hipEventDisableTiming = 2;
inline hipError_t hipEventCreateWithFlags(hipEvent_t *e, unsigned) { return hipEventCreate(e); }
inline hipError_t hipStreamWaitEvent(hipStream_t, hipEvent_t, unsigned) { return hipSuccess; } // launches run synchronously here
inline hipError_t hipEventRecord(hipEvent_t e, hipStream_t = nullptr) {
    e->t = std::chrono::steady_clock::now();
    return hipSuccess;
}
inline hipError_t hipEventSynchronize(hipEvent_t) { return hipSuccess; }
inline hipError_t hipEventElapsedTime(float *ms, hipEvent_t a, hipEvent_t b) {
    *ms = std::chrono::duration<float, std::milli>(b->t - a->t).count();
    return hipSuccess;
}
