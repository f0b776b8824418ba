#!/usr/bin/env python3
"""bench.py -- factorize+solve of the 1M-DOF 2D 5-point Poisson matrix (BASELINE.json configs[1]) on MI355X.

A "step" is one pass of the hot path: numeric factorisation (values already resident in HBM) followed
by one solve (rhs resident in HBM, default iterative refinement).  Prints ONE JSON line (rank 0).

  python bench.py --gpus 1 --steps 5 --warmup 2
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N --steps K --warmup W

N > 1 (many-RHS path): the right-hand sides are sharded over the ranks (one block per
GPU, weak scaling); every rank holds the factor of the same matrix.  No data-path collective is needed
for the solves themselves; see DESIGN.md ("multi-GPU").
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

HBM_PEAK_GBS = 8000.0      # MI355X_MICROARCH.md: HBM3E 8 TB/s spec (6.29 TB/s measured copy)
FP64_PEAK_TFLOPS = 78.6    # MI355X FP64 (vector = matrix) peak, datasheet


def sptrsv_bytes(st, n, k=1):
    """(nnzL + nnzU)*(8+4) + 2(n+1)*4 + k*n*8*2*2 with the ACTUAL factor sizes."""
    return (st["nnz_l"] + st["nnz_u"]) * 12 + 2 * (n + 1) * 4 + k * n * 8 * 4


def measured_traffic(grid):
    """HBM bytes per SpTRSV pass from the PMC passes committed under profiles/ (rocprofv3 cannot run inside the timed
    loop; the counters were collected with this same command on the same workload, see profiles/r01_v6_pmc_hbm.txt)."""
    path = os.path.join(ROOT, "profiles", "r01_sptrsv_traffic.json")
    if grid != 1000 or not os.path.exists(path):
        return None, None
    with open(path) as fh:
        t = json.load(fh)
    return t["traffic_bytes_per_pass"], t["source"]


def cpu_baseline(n, rp, ci, v, b, perm):
    """The CPU oracle (kind 'port') timed on this box's host cores, single thread, same matrix/ordering."""
    import oracle_lib as O
    rows = np.repeat(np.arange(n, dtype=np.int32), np.diff(rp))
    cp, ri, vx = O.coo_to_csc(n, n, rows, ci, v)
    t0 = time.perf_counter()
    lu = O.OracleLU(n, cp, ri, vx, q=perm)
    t1 = time.perf_counter()
    x = lu.solve(b, nrefine=2)
    t2 = time.perf_counter()
    r = np.zeros(n)
    np.add.at(r, rows, v * x[ci])
    return {"value": round((t2 - t0) * 1e3, 2), "unit": "ms", "cores": 1, "kind": "port",
            "sample": "oracle/oracle.c left-looking LU (threshold pivoting, SUM scaling, <=2 refinement steps) on the SAME %d-DOF "
                      "matrix with the same fill-reducing ordering: factorize %.1f ms + solve %.1f ms, residual %.1e; host has %d cores"
                      % (n, (t1 - t0) * 1e3, (t2 - t1) * 1e3, float(np.max(np.abs(r - b))), os.cpu_count() or 0)}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--grid", type=int, default=1000, help="nx = ny of the 2D 5-point Poisson grid (1000 = BASELINE config)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-grid", type=int, default=0, help="grid of the CPU-baseline sample (0 = same as --grid)")
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    dist = None
    if world > 1 or os.environ.get("BENCH_FORCE_DIST") == "1":
        import torch
        import torch.distributed as dist
        torch.cuda.set_device(local_rank)
        dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))

    from russell_amd import problems as P
    from russell_amd.backend import Hipmf

    from russell_amd import _capi
    if _capi.load().hipmf_set_device(local_rank) != 0:
        raise RuntimeError("hipmf_set_device(%d) failed" % local_rank)

    n, rp, ci, v = P.poisson2d(args.grid)
    xs = P.manufactured_solution(n)
    b = P.csr_matvec(n, rp, ci, v, xs) + float(rank)  # every rank owns a different right-hand side

    s = Hipmf()
    t0 = time.perf_counter()
    code = s.initialize(n, rp, ci)
    t_init = time.perf_counter() - t0
    assert code == 0, code
    d_vals = s.dev_alloc(v.nbytes)
    d_b = s.dev_alloc(b.nbytes)
    d_x = s.dev_alloc(b.nbytes)
    s.h2d(d_vals, v)
    s.h2d(d_b, b)

    def step():
        c = s.factorize_device(d_vals)
        assert c == 0, c
        s.solve_device(d_x, d_b)

    def barrier():
        s.lib.hipmf_device_synchronize()
        if dist is not None:
            import torch
            dist.barrier()
            torch.cuda.synchronize()

    for _ in range(args.warmup):
        step()
    s.reset_timers()
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    barrier()
    elapsed = time.perf_counter() - t0
    if dist is not None:
        import torch
        from russell_amd.distributed import max_over_ranks
        elapsed = max_over_ranks(elapsed, dist, device=torch.device("cuda", local_rank))
    ms_per_step = elapsed * 1e3 / args.steps

    st = s.stats()
    x = np.zeros(n)
    s.d2h(x, d_x)
    resid = np.zeros(n)
    np.add.at(resid, np.repeat(np.arange(n), np.diff(rp)), v * x[ci])
    rel_err = float(np.max(np.abs(resid - b)) / (np.max(np.abs(v)) + 1.0))

    if rank == 0:
        tri = max(st["acc_tri_count"], 1.0)
        tri_ms = (st["acc_fwd_ms"] + st["acc_bwd_ms"]) / tri
        import ctypes
        copy_gbs = ctypes.c_double(0.0)
        if s.lib.hipmf_device_copy_bandwidth(1 << 30, 3, ctypes.byref(copy_gbs)) != 0:
            copy_gbs.value = 0.0
        mfma_tfs = ctypes.c_double(0.0)
        if s.lib.hipmf_device_mfma_rate(1024, 4000, ctypes.byref(mfma_tfs)) != 0:
            mfma_tfs.value = 0.0
        bytes_alg = sptrsv_bytes(st, n)
        traffic, traffic_src = measured_traffic(args.grid)
        achieved = bytes_alg / (tri_ms * 1e-3) / 1e9 if tri_ms > 0 else 0.0
        fact_ms = st["acc_factor_ms"] / max(st["acc_factor_count"], 1.0)
        asm_ms = st["acc_assemble_ms"] / max(st["acc_factor_count"], 1.0)
        out = {
            "metric": "factorize+solve time (ms) + SpTRSV GB/s, 1M-DOF 5-pt Poisson f64, 1/2/4/8 GPU",
            "value": round(ms_per_step, 3),
            "unit": "ms",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": round(ms_per_step, 3),
            "higher_is_better": False,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "f64",
            "data": "synthetic",
            "config": {"workload": "2D 5-point Poisson %dx%d grid (n=%d, nnz=%d) f64, 1 RHS per GPU, numeric factorize + solve "
                                   "with values and rhs resident in HBM" % (args.grid, args.grid, n, int(rp[-1])),
                       "rhs_per_gpu": 1, "refinement_steps": st["refinement_steps"]},
            "sptrsv_gbs": round(achieved, 1),
            "roofline": {"kernel": "multifrontal SpTRSV pass, forward + backward (%s, %d launches over %d tree levels)" %
                                   ("dependency-driven k_fwd_fused + k_bwd_fused" if st["solve_launches"] <= 4 else "level-set k_fwd/k_bwd[_big]",
                                    st["solve_launches"], st["nlevels"]),
                         "bound": "hbm", "achieved": round(achieved, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": round(achieved / HBM_PEAK_GBS, 4), "traffic": traffic, "traffic_source": traffic_src,
                         "algorithmic_bytes": int(bytes_alg), "avg_ms": round(tri_ms, 4),
                         "measured_copy_gbs": round(copy_gbs.value, 1),
                         "frac_of_measured_copy": round(achieved / copy_gbs.value, 4) if copy_gbs.value > 0 else None},
            "roofline_factor": {"kernel": "numeric multifrontal LU (k_small_factor, k_panel, k_update MFMA f64, k_extend_add)",
                                "bound": "mfma", "achieved": round(st["flops"] / (fact_ms * 1e-3) / 1e12, 3) if fact_ms > 0 else 0.0,
                                "peak": FP64_PEAK_TFLOPS, "unit": "TFLOP/s",
                                "frac": round(st["flops"] / (fact_ms * 1e-3) / 1e12 / FP64_PEAK_TFLOPS, 5) if fact_ms > 0 else 0.0,
                                "flops": st["flops"], "avg_ms": round(fact_ms, 3), "assemble_ms": round(asm_ms, 3),
                                "measured_mfma_tflops": round(mfma_tfs.value, 1)},
            "phases_ms": {"initialize_once": round(t_init * 1e3, 1), "ordering_s": st["ordering_s"], "symbolic_s": st["symbolic_s"],
                          "assemble": round(asm_ms, 3), "factor": round(fact_ms, 3), "sptrsv_pair": round(tri_ms, 4),
                          "solve_total_last": round(st["solve_total_ms"], 3)},
            "factor": {"nnz_l": st["nnz_l"], "nnz_u": st["nnz_u"], "nsuper": st["nsuper"], "nlevels": st["nlevels"],
                       "max_front": st["max_front"], "pool_gb": round(st["pool_bytes"] / 1e9, 3),
                       "factor_launches": st["factor_launches"], "perturbed": st["n_perturbed"]},
            "relative_error": rel_err,
        }
        if not args.no_cpu_baseline and world == 1:
            if args.cpu_grid and args.cpu_grid != args.grid:
                n2, rp2, ci2, v2 = P.poisson2d(args.cpu_grid)
                s2 = Hipmf()
                s2.initialize(n2, rp2, ci2)
                perm2 = s2.permutation()
                s2.close()
                out["cpu_baseline"] = cpu_baseline(n2, rp2, ci2, v2, P.csr_matvec(n2, rp2, ci2, v2, P.manufactured_solution(n2)), perm2)
            else:
                out["cpu_baseline"] = cpu_baseline(n, rp, ci, v, b, s.permutation())
        line = json.dumps(out)
    else:
        line = None
    s.dev_free(d_vals), s.dev_free(d_b), s.dev_free(d_x)
    s.close()
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()
    if line is not None:
        # the JSON line goes out LAST and unbuffered: RCCL prints its banner through C stdio, which would otherwise
        # land after Python's buffered print
        import ctypes
        sys.stdout.flush()
        try:
            ctypes.CDLL(None).fflush(None)
        except OSError:
            pass
        os.write(1, (line + "\n").encode())


if __name__ == "__main__":
    main()
