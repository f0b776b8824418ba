#!/usr/bin/env python3
"""Where a complex factorisation goes (BASELINE config 5's K_comp at npoint): wall time of complex_solver_hipmf_factorize against
the device times of its real-equivalent system (assemble = scaling + value expansion, factor = numeric LU)."""
import ctypes as C, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from russell_amd import problems as P, _capi
npoint = int(sys.argv[1]) if len(sys.argv) > 1 else 513
n, rp, ci, v0 = P.brusselator_pattern(npoint, gamma=0.0)
rows = np.repeat(np.arange(n), np.diff(rp))
h = 1e-4
kc = v0.astype(complex) + ((2.6810828736277521 + 3.0504301992474105j) / h) * (rows == ci)
zv = np.ascontiguousarray(np.stack([kc.real, kc.imag], axis=1).ravel())
lib = _capi.load()
hd = lib.complex_solver_hipmf_new()
t0 = time.perf_counter()
assert lib.complex_solver_hipmf_initialize(hd, 0, 1, -1.0, -1, 0, 0, n, rp, ci, zv.ctypes.data) == 0
print("initialize %.1f ms" % ((time.perf_counter() - t0) * 1e3))
for rep in range(4):
    t0 = time.perf_counter()
    assert lib.complex_solver_hipmf_factorize(hd, None, None, None, None, None, None, None, 0, 0, zv) == 0
    t1 = time.perf_counter()
    i, d = np.zeros(16, np.int64), np.zeros(16)
    lib.complex_solver_hipmf_get_stats(hd, i, d)
    print("factorize wall %.1f ms: device assemble %.2f ms, numeric LU %.2f ms; 2n = %d, levels %d, max front %d, launches %d, flops %.3e" %
          ((t1 - t0) * 1e3, d[4], d[5], 2 * n, i[3], i[6], i[11], d[0]))
# the real system of the same step for comparison
from russell_amd.backend import Hipmf
kr = v0 + (3.6378342527444957 / h) * (rows == ci)
s = Hipmf()
assert s.initialize(n, rp, ci) == 0
for rep in range(3):
    t0 = time.perf_counter(); assert s.factorize(kr) == 0; t1 = time.perf_counter()
st = s.stats()
print("real: factorize wall %.1f ms: device assemble %.2f ms, numeric LU %.2f ms; n = %d, levels %d, max front %d, launches %d, flops %.3e" %
      ((t1 - t0) * 1e3, st["assemble_ms"], st["factor_ms"], n, st["nlevels"], st["max_front"], st["factor_launches"], st["flops"]))
