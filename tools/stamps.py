#!/usr/bin/env python3
"""Device-clock stamps of an instrumented build (-DHIPMF_STAMPS, see kernels_common.hpp): factorise the 2D Poisson problem and
print what the HIPMF_STAMP(row, slot) / HIPMF_STAMP_VAL(row, slot, value) marks of the kernel under study recorded.
The tree carries no marks: add them to a kernel (row = e.g. blockIdx.x, slot 0 at its entry), build the library with
-DHIPMF_STAMPS, run this script on it.  profiles/r02_rejected_experiments.txt shows a table obtained this way.

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
print("row | microseconds from slot 0 for the clock slots (HIPMF_STAMP), raw values for the others (HIPMF_STAMP_VAL)")
for k in range(1024):
    r = rows[k]
    if r[0] == 0:
        continue
    if k > 90 and k % 37:
        continue
    out = []
    for x in r[1:]:
        if x == 0:
            continue
        d = int(x) - int(r[0])
        out.append("%8.2f" % (d / 100.0) if 0 <= d < 10 ** 9 else "%8d" % int(x))  # a clock near slot 0's, else a stored value
    print("%4d | %s" % (k, " ".join(out)))
