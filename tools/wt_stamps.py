#!/usr/bin/env python3
"""Device-clock stamps of k_wt_fwd (instrumented build, -DHIPMF_STAMPS): per sampled workgroup (wave 0) the start, the end of the
prologue, the end of the first front and the end of the subtree.  usage: python tools/wt_stamps.py <instrumented .so> [grid]"""
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
b = P.csr_matvec(n, rp, ci, v, P.manufactured_solution(n))
s = Hipmf(lib)
assert s.initialize(n, rp, ci, refinement_nstep=0) == 0
assert s.factorize(v) == 0
raw = C.CDLL(lib)
buf = np.zeros(16 * 1024, np.uint64)
for _ in range(3):
    s.solve(b)
raw.hipmf_debug_read_stamps(buf.ctypes.data_as(C.c_void_p), C.c_int64(buf.size))  # (clears)
s.solve(b)
assert raw.hipmf_debug_read_stamps(buf.ctypes.data_as(C.c_void_p), C.c_int64(buf.size)) == 0
rows = buf.reshape(1024, 16).astype(np.int64)
ok = rows[:, 0] > 0
t0 = rows[ok, 0].min()
print("row  start_us  prologue_us  first_batch_us  total_us  batches  us_per_batch")
R = rows[ok]
for k in range(0, len(R), max(1, len(R) // 60)):
    r = R[k]
    nf = max(int(r[4]), 1)
    print("%4d %8.2f %8.2f %8.2f %8.2f %4d %8.2f" % (k, (r[0] - t0) / 100.0, (r[1] - r[0]) / 100.0, (r[2] - r[0]) / 100.0, (r[3] - r[0]) / 100.0, nf,
                                                  (r[3] - r[1]) / 100.0 / nf))
tot = (R[:, 3] - R[:, 1]) / 100.0
nf = np.maximum(R[:, 4], 1)
print("sampled waves %d: launch span %.1f us; mean per-batch time %.2f us (median %.2f); mean prologue %.2f us" %
      (len(R), (R[:, 3].max() - t0) / 100.0, float(np.sum(tot) / np.sum(nf)), float(np.median(tot / nf)), float(np.mean((R[:, 1] - R[:, 0]) / 100.0))))
two = R[R[:, 4] >= 2]
if len(two):
    park = (two[:, 6] - two[:, 5]) / 100.0
    issue = (two[:, 7] - two[:, 6]) / 100.0
    comp = (two[:, 8] - two[:, 7]) / 100.0
    nrec = np.maximum(two[:, 9], 1)
    print("second batch of %d waves: wait+park %.2f us, issue next+sync %.2f us, compute %.2f us for %.1f fronts on average = %.2f us per front" %
          (len(two), float(np.mean(park)), float(np.mean(issue)), float(np.mean(comp)), float(np.mean(nrec)), float(np.sum(comp) / np.sum(nrec))))
# concurrency: how many sampled waves are alive over time
ts = np.arange(0, (R[:, 3].max() - t0) / 100.0, 5.0)
alive = [(int(np.sum(((R[:, 0] - t0) / 100.0 <= t) & ((R[:, 3] - t0) / 100.0 > t)))) for t in ts]
print("alive sampled waves every 5 us:", alive)
s.close()
