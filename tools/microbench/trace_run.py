import ctypes as C, sys, os, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from russell_amd import problems as P
from russell_amd.backend import Hipmf
lib = os.path.join(os.path.dirname(os.path.abspath(__file__)), "libtrace.so")
n, rp, ci, v = P.poisson2d(int(sys.argv[1]) if len(sys.argv) > 1 else 1000)
s = Hipmf(lib); s.initialize(n, rp, ci); s.factorize(v); s.factorize(v)
raw = C.CDLL(lib); out = (C.c_longlong * 64)(); raw.hipmf_read_trace.argtypes = [C.c_void_p]; raw.hipmf_read_trace(out)
t = list(out)
print("update (block 1):", [t[i] - t[10] for i in range(10, 16)])
print("update (block 0):", [t[i] - t[20] for i in range(20, 27)])
