"""Chained tiled steps (k_chain, one launch per level) against one launch per step: same factor bit for bit (solutions and determinant
compared), and the factorisation times at 1000 x 1000 for several settings of HIPMF_CHAIN_MAX_WGS.  Runs on the GPU box."""
import os
import sys

import numpy as np

sys.path.insert(0, __file__.rsplit("/", 2)[0])
from russell_amd import problems as P  # noqa: E402
from russell_amd.backend import Hipmf  # noqa: E402


def run(n, rp, ci, v, env, reps=1, **kw):
    old = {k: os.environ.get(k) for k in env}
    os.environ.update(env)
    try:
        s = Hipmf()
        assert s.initialize(n, rp, ci, refinement_nstep=0, **kw) == 0
        outs = []
        for _ in range(reps):
            assert s.factorize(v, compute_determinant=True) == 0
            outs.append((s.solve(np.cos(np.arange(n))), s.det_coefficient, s.det_exponent))
        st = s.stats()
        fms = []
        for _ in range(5):
            s.reset_timers()
            s.factorize(v)
            fms.append(s.stats()["factor_ms"])
        s.close()
        return outs, st, min(fms)
    finally:
        for k, val in old.items():
            if val is None:
                os.environ.pop(k, None)
            else:
                os.environ[k] = val


def sym(c):
    n, rp, ci, v = c
    rpl, cil, vl = P.lower_triangle(n, rp, ci, v)
    return n, rpl, cil, vl


cases = [("poisson2d 300", P.poisson2d(300), {}), ("convection-diffusion 200", P.convection_diffusion2d(200, peclet=30.0, scale_decades=0.0), {}),
         ("poisson3d 30", P.poisson3d(30), {}), ("poisson3d 32 lower", sym(P.poisson3d(32)), {"general_symmetric": True}),
         ("poisson2d 1000", P.poisson2d(1000), {}), ("poisson2d 1000 lower", sym(P.poisson2d(1000)), {"general_symmetric": True})]
only = sys.argv[1] if len(sys.argv) > 1 else ""
for name, (n, rp, ci, v), kw in cases:
    if only and only not in name:
        continue
    (ref,), st0, t0 = run(n, rp, ci, v, {"HIPMF_FACTOR_CHAIN": "0"}, **kw)
    print("%-26s per-step launches: %4d launches, factor %.3f ms" % (name, st0["factor_launches"], t0), flush=True)
    for mx in ("16384",):  # (the default)
        for fine in ("0",):
            outs, st1, t1 = run(n, rp, ci, v, {"HIPMF_FACTOR_CHAIN": "1", "HIPMF_CHAIN_FINE": fine, "HIPMF_CHAIN_MAX_WGS": mx}, reps=4, **kw)
            same = all(np.array_equal(ref[0], o[0]) and ref[1:] == o[1:] for o in outs)
            print("    chain max_wgs %6s fine %s: %4d launches, factor %.3f ms, bitwise equal %s" % (mx, fine, st1["factor_launches"], t1, same), flush=True)
