#!/usr/bin/env python3
"""Host-side cost of `initialize` (ordering, symbolic analysis, plan upload) for the 2D / 3D Poisson matrices, with the
nested dissection on 1 thread and on the default pool.  usage: python tools/init_time.py [2d N | 3d N] ..."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402

from russell_amd import problems as P  # noqa: E402
from russell_amd.backend import Hipmf  # noqa: E402


def main():
    args = sys.argv[1:] or ["2d", "1000"]
    for kind, size in zip(args[0::2], args[1::2]):
        n, rp, ci, v = P.poisson2d(int(size)) if kind == "2d" else P.poisson3d(int(size))
        ref = None
        for threads in ("1", "4", "16", "32", ""):
            if threads:
                os.environ["HIPMF_ND_THREADS"] = threads
            else:
                os.environ.pop("HIPMF_ND_THREADS", None)
            s = Hipmf()
            t0 = time.perf_counter()
            assert s.initialize(n, rp, ci) == 0
            dt = time.perf_counter() - t0
            st = s.stats()
            p = s.permutation().copy()
            s.close()
            if ref is None:
                ref = p
            print("%s %s n=%d threads=%-7s initialize %.3f s (ordering %.3f s, symbolic total %.3f s) same permutation: %s"
                  % (kind, size, n, threads or "default", dt, st["ordering_s"], st["symbolic_s"], bool(np.array_equal(p, ref))), flush=True)


main()
