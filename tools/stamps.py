#!/usr/bin/env python3
"""Device-clock stamps of an instrumented build (-DHIPMF_STAMPS, see kernels_common.hpp): factorise the 2D Poisson problem and
print the per-workgroup phase times of the kernels that carry HIPMF_STAMP marks.

usage: python tools/stamps.py <instrumented librussell_hipmf.so> [grid]
"""
import ctypes as C
import os
import sys

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from russell_amd import problems as P
from russell_amd.backend import Hipmf

lib = sys.argv[1]
N = int(sys.argv[2]) if len(sys.argv) > 2 else 1000
n, rp, ci, v = P.poisson2d(N)
s = Hipmf(lib)
assert s.initialize(n, rp, ci) == 0
for _ in range(3):
    assert s.factorize(v) == 0
raw = C.CDLL(lib)
buf = np.zeros(16 * 1024, np.uint64)
assert raw.hipmf_debug_read_stamps(buf.ctypes.data_as(C.c_void_p), C.c_int64(buf.size)) == 0
rows = buf.reshape(1024, 16).astype(np.int64)
print("row: p f | microseconds from the first stamp")
for k in range(1024):
    r = rows[k]
    if r[0] == 0:
        continue
    if k > 90 and k % 37:
        continue
    d = [(int(x) - int(r[0])) / 100.0 for x in r[1:8] if x]
    e = [(int(x) - int(r[1])) / 100.0 for x in r[10:14] if x]
    print("%4d: p %3d f %3d | %s | first block from stamp 1: %s" % (k, r[8], r[9], " ".join("%7.2f" % x for x in d), " ".join("%6.2f" % x for x in e)))
