#!/usr/bin/env python3
"""Host <-> device copy rates as the host-pointer boundary sees them: pageable memory, the same memory after hipHostRegister, hipHostMalloc.
usage: host_copy_rates.py [MB]"""
import ctypes as C
import sys
import time

import numpy as np

hip = C.CDLL("libamdhip64.so")
MB = int(sys.argv[1]) if len(sys.argv) > 1 else 40
n = MB * (1 << 20)
dev = C.c_void_p()
assert hip.hipMalloc(C.byref(dev), C.c_size_t(n)) == 0
H2D, D2H = 1, 2


def rate(ptr, kind, reps=8):
    best = 1e9
    for _ in range(reps):
        t0 = time.perf_counter()
        if kind == H2D:
            assert hip.hipMemcpy(dev, C.c_void_p(ptr), C.c_size_t(n), 1) == 0
        else:
            assert hip.hipMemcpy(C.c_void_p(ptr), dev, C.c_size_t(n), 2) == 0
        best = min(best, time.perf_counter() - t0)
    return n / best / 1e9, best * 1e3


a = np.random.default_rng(0).standard_normal(n // 8)
p = a.ctypes.data
print("%d MB" % MB)
print("pageable        H2D %.1f GB/s (%.2f ms)   D2H %.1f GB/s (%.2f ms)" % (rate(p, H2D) + rate(p, D2H)))
t0 = time.perf_counter()
assert hip.hipHostRegister(C.c_void_p(p), C.c_size_t(n), 0) == 0
t_reg = time.perf_counter() - t0
print("hipHostRegister %.2f ms" % (t_reg * 1e3))
print("registered      H2D %.1f GB/s (%.2f ms)   D2H %.1f GB/s (%.2f ms)" % (rate(p, H2D) + rate(p, D2H)))
t0 = time.perf_counter()
assert hip.hipHostUnregister(C.c_void_p(p)) == 0
print("hipHostUnregister %.2f ms" % ((time.perf_counter() - t0) * 1e3))
hp = C.c_void_p()
assert hip.hipHostMalloc(C.byref(hp), C.c_size_t(n), 0) == 0
print("hipHostMalloc   H2D %.1f GB/s (%.2f ms)   D2H %.1f GB/s (%.2f ms)" % (rate(hp.value, H2D) + rate(hp.value, D2H)))
t0 = time.perf_counter()
C.memmove(hp, C.c_void_p(p), n)
print("memcpy pageable -> pinned %.1f GB/s" % (n / (time.perf_counter() - t0) / 1e9))
# every repetition, not the best one: the same pageable buffer, then a fresh buffer each time
def one(ptr, kind):
    t0 = time.perf_counter()
    if kind == H2D:
        assert hip.hipMemcpy(dev, C.c_void_p(ptr), C.c_size_t(n), 1) == 0
    else:
        assert hip.hipMemcpy(C.c_void_p(ptr), dev, C.c_size_t(n), 2) == 0
    return (time.perf_counter() - t0) * 1e3
b = np.random.default_rng(1).standard_normal(n // 8)
print("same pageable buffer, H2D ms per repetition:", " ".join("%.2f" % one(b.ctypes.data, H2D) for _ in range(6)))
print("same pageable buffer, D2H ms per repetition:", " ".join("%.2f" % one(b.ctypes.data, D2H) for _ in range(6)))
ts = []
for k in range(5):
    c = np.random.default_rng(k).standard_normal(n // 8)
    ts.append(one(c.ctypes.data, H2D))
print("fresh pageable buffer each time, H2D ms:", " ".join("%.2f" % t for t in ts))
ts = []
for k in range(5):
    c = np.empty(n // 8)
    ts.append(one(c.ctypes.data, D2H))
print("fresh (untouched) pageable buffer each time, D2H ms:", " ".join("%.2f" % t for t in ts))
