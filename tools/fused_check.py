#!/usr/bin/env python3
"""Compare the dependency-driven solve (one launch per direction) with the level-set launches: same factor,
same right-hand side, identical slab shapes (HIPMF_SOLVE_SLAB64=1) -> the two must agree bit for bit."""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from russell_amd import problems as P  # noqa: E402
from russell_amd.backend import Hipmf  # noqa: E402


def run(grid, fused, slab64, reps):
    os.environ["HIPMF_FUSED_SOLVE"] = "1" if fused else "0"
    os.environ["HIPMF_SOLVE_SLAB64"] = "1" if slab64 else "0"
    n, rp, ci, v = P.poisson2d(grid)
    xs = P.manufactured_solution(n)
    b = P.csr_matvec(n, rp, ci, v, xs)
    s = Hipmf()
    assert s.initialize(n, rp, ci, refinement_nstep=0) == 0
    d_v, d_b, d_x = s.dev_alloc(v.nbytes), s.dev_alloc(b.nbytes), s.dev_alloc(b.nbytes)
    s.h2d(d_v, v), s.h2d(d_b, b)
    assert s.factorize_device(d_v) == 0
    outs = []
    t0 = time.perf_counter()
    for _ in range(reps):
        s.solve_device(d_x, d_b)
        x = np.zeros(n)
        s.d2h(x, d_x)
        outs.append(x)
    dt = (time.perf_counter() - t0) / reps
    st = s.stats()
    s.close()
    return outs, xs, st, dt


if __name__ == "__main__":
    grid = int(sys.argv[1]) if len(sys.argv) > 1 else 1000
    reps = int(sys.argv[2]) if len(sys.argv) > 2 else 10
    lv, xs, st_l, _ = run(grid, False, True, 1)
    fu, _, st_f, _ = run(grid, True, True, reps)
    bad = sum(0 if np.array_equal(lv[0], x) else 1 for x in fu)
    print("grid %d: level-set err %.2e; fused(slab64) %d/%d bitwise equal to level-set, max diff %.3e" %
          (grid, np.max(np.abs(lv[0] - xs)), reps - bad, reps, max(np.max(np.abs(lv[0] - x)) for x in fu)))
    fd, _, st_d, _ = run(grid, True, False, reps)
    print("fused(default slabs): max err %.2e, all runs identical: %s" %
          (max(np.max(np.abs(x - xs)) for x in fd), all(np.array_equal(fd[0], x) for x in fd)))
    print("sptrsv ms (fwd + bwd): level %.4f + %.4f, fused64 %.4f + %.4f, fused %.4f + %.4f" % tuple(
        v for st in (st_l, st_f, st_d) for v in (st["acc_fwd_ms"] / max(st["acc_tri_count"], 1), st["acc_bwd_ms"] / max(st["acc_tri_count"], 1))))
    sys.exit(1 if bad else 0)
