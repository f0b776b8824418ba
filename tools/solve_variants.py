#!/usr/bin/env python3
"""A/B of the triangular-solve schedules on one matrix: forward / backward pass times (HIP events on the solver's stream) per variant.

  python tools/solve_variants.py [grid=1000] [3d]      (3d: 7-point Poisson grid^3 as its lower triangle, L D L^T)
Variants are sets of environment knobs read at initialize."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from russell_amd import problems as P  # noqa: E402
from russell_amd.backend import Hipmf  # noqa: E402

grid = int(sys.argv[1]) if len(sys.argv) > 1 else 1000
three_d = len(sys.argv) > 2 and sys.argv[2] == "3d"
if three_d:
    n, rp, ci, v = P.poisson3d(grid)
    b = P.csr_matvec(n, rp, ci, v, P.manufactured_solution(n))
    rp, ci, v = P.lower_triangle(n, rp, ci, v)
    kw = {"general_symmetric": True}
else:
    n, rp, ci, v = P.poisson2d(grid)
    b = P.csr_matvec(n, rp, ci, v, P.manufactured_solution(n))
    kw = {}

VARIANTS = [
    ("defaults (tagged hand-offs, wave fronts)", {}),
    ("wave fronts in the forward pass only", {"HIPMF_WAVE_FRONTS_BWD": "0"}),
    ("slab tasks instead of wave fronts", {"HIPMF_WAVE_FRONTS": "0"}),
    ("completion counters (HIPMF_TAG_SOLVE=0)", {"HIPMF_TAG_SOLVE": "0"}),
    ("counters, no wave fronts (round 4)", {"HIPMF_TAG_SOLVE": "0", "HIPMF_WAVE_FRONTS": "0"}),
    ("top = levels with <= 16 fronts", {"HIPMF_UP_TOP_FRONTS": "16"}),
    ("top = levels with <= 120 fronts", {"HIPMF_UP_TOP_FRONTS": "120"}),
    ("mid stage 0", {"HIPMF_UP_STAGE_MID": "0"}),
    ("caps 40 fronts / 128 KB per wave-subtree", {"HIPMF_WT_FRONTS": "40", "HIPMF_WT_KB": "128"}),
]
if os.environ.get("SOLVE_VARIANTS_SHORT"):
    VARIANTS = VARIANTS[:3]
LIB = next((a[4:] for a in sys.argv[2:] if a.startswith("lib=")), None)
ONLY = next((a[5:] for a in sys.argv[2:] if a.startswith("only=")), None)
extra = [a for a in sys.argv[2:] if "=" in a and not a.startswith("lib=") and not a.startswith("only=")]
if extra:
    VARIANTS.append(("command line: " + " ".join(extra), dict(a.split("=", 1) for a in extra)))

ref = None
print("matrix: %s grid %d, n = %d" % ("3D 7-point (lower triangle)" if three_d else "2D 5-point", grid, n))
for name, env in VARIANTS:
    if ONLY and ONLY not in name:
        continue
    old = {k: os.environ.get(k) for k in env}
    os.environ.update(env)
    try:
        s = Hipmf(LIB) if LIB else Hipmf()
        assert s.initialize(n, rp, ci, refinement_nstep=0, **kw) == 0
        d_v, d_b, d_x = s.dev_alloc(v.nbytes), s.dev_alloc(b.nbytes), s.dev_alloc(b.nbytes)
        s.h2d(d_v, v), s.h2d(d_b, b)
        assert s.factorize_device(d_v) == 0
        for _ in range(3):
            s.solve_device(d_x, d_b)
        s.reset_timers()
        for _ in range(20):
            s.solve_device(d_x, d_b)
        st = s.stats()
        x = np.zeros(n)
        s.d2h(x, d_x)
        if ref is None:
            ref = x
        fwd, bwd = 1e3 * st["acc_fwd_ms"] / st["acc_tri_count"], 1e3 * st["acc_bwd_ms"] / st["acc_tri_count"]
        print("%-42s fwd %7.1f us  bwd %7.1f us  pair %7.1f us  launches %d  fallbacks %d  max|x - x_first| %.2e  bit-equal %s" %
              (name, fwd, bwd, fwd + bwd, st["solve_launches"], st.get("fused_fallbacks", 0), float(np.max(np.abs(x - ref))), np.array_equal(x, ref)))
        sys.stdout.flush()
        for p in (d_v, d_b, d_x):
            s.dev_free(p)
        s.close()
    finally:
        for k, val in old.items():
            if val is None:
                os.environ.pop(k, None)
            else:
                os.environ[k] = val
