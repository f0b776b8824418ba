#!/usr/bin/env python3
"""Many right-hand sides sharded over the GPUs of one node, one process per GPU:

    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        tools/many_rhs_multi_gpu.py [2d|3d] SIZE NRHS [--replicate]

Every rank runs `initialize` (deterministic: same plan everywhere).  Default: rank 0 factorises and the numeric factor is
broadcast over RCCL / xGMI straight between the solvers' device buffers (russell_amd.distributed.broadcast_factor), then
every rank solves its contiguous block of the NRHS columns, resident in its HBM, with the blocked dependency-driven solves.
--replicate: every rank factorises itself instead (cheaper when factor bytes / link bandwidth exceeds the factorisation time).
Rank 0 prints one JSON line with the phase times (max over ranks) and the aggregate solve rate."""
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    import torch
    import torch.distributed as dist

    from russell_amd import _capi
    from russell_amd import problems as P
    from russell_amd.backend import Hipmf
    from russell_amd.distributed import broadcast_factor, max_over_ranks, rhs_block

    args = [a for a in sys.argv[1:] if not a.startswith("--")]
    replicate = "--replicate" in sys.argv
    kind, size, nrhs = (args + ["2d", "1000", "64"])[:3] if len(args) < 3 else args[:3]
    size, nrhs = int(size), int(nrhs)
    rank, world, local = int(os.environ.get("RANK", "0")), int(os.environ.get("WORLD_SIZE", "1")), int(os.environ.get("LOCAL_RANK", "0"))
    torch.cuda.set_device(local)
    device = torch.device("cuda", local)
    dist.init_process_group("nccl", device_id=device)
    if _capi.load().hipmf_set_device(local) != 0:
        raise RuntimeError("hipmf_set_device(%d) failed" % local)

    n, rp, ci, v = P.poisson2d(size) if kind == "2d" else P.poisson3d(size)
    start, count = rhs_block(nrhs, world, rank)
    rng = np.random.default_rng(20260927 + rank)
    XS = rng.standard_normal((max(count, 1), n))
    B = np.array([P.csr_matvec(n, rp, ci, v, XS[j]) for j in range(max(count, 1))])

    s = Hipmf()
    t0 = time.perf_counter()
    assert s.initialize(n, rp, ci) == 0
    t_init = time.perf_counter() - t0
    d_v, d_b, d_x = s.dev_alloc(v.nbytes), s.dev_alloc(B.nbytes), s.dev_alloc(B.nbytes)
    s.h2d(d_v, v)
    s.h2d(d_b, B)

    def sync():
        s.lib.hipmf_device_synchronize()
        dist.barrier()
        torch.cuda.synchronize()

    sync()
    t0 = time.perf_counter()
    if replicate or rank == 0:
        assert s.factorize_device(d_v) == 0
    s.lib.hipmf_device_synchronize()
    t_fact = time.perf_counter() - t0
    sync()
    t0 = time.perf_counter()
    nbytes = 0 if replicate else broadcast_factor(s, d_v, dist, src=0, device=device)
    sync()
    t_bcast = time.perf_counter() - t0
    t0 = time.perf_counter()
    if count > 0:
        s.solve_device(d_x, d_b, count, n)
    s.lib.hipmf_device_synchronize()
    t_solve = time.perf_counter() - t0
    X = np.zeros_like(B)
    s.d2h(X, d_x)
    err = float(np.max(np.abs(X[:count] - XS[:count]))) if count > 0 else 0.0
    t_fact, t_bcast, t_solve = (max_over_ranks(t, dist, device) for t in (t_fact, t_bcast, t_solve))
    err = max_over_ranks(err, dist, device)
    if rank == 0:
        st = s.stats()
        print(json.dumps({"workload": "%s Poisson %d, n=%d, %d right-hand sides over %d GPU(s)" % (kind, size, n, nrhs, world),
                          "mode": "replicated factorisation" if replicate else "factorise on rank 0 + RCCL broadcast of the factor",
                          "n_gpus": world, "initialize_s": round(t_init, 3), "factorize_ms": round(t_fact * 1e3, 3),
                          "broadcast_ms": round(t_bcast * 1e3, 3), "broadcast_bytes": nbytes,
                          "broadcast_gbs": round(nbytes / t_bcast / 1e9, 1) if nbytes and t_bcast > 0 else None,
                          "solve_ms": round(t_solve * 1e3, 3), "rhs_per_s": round(nrhs / t_solve, 1) if t_solve > 0 else None,
                          "total_ms": round((t_fact + t_bcast + t_solve) * 1e3, 3), "max_abs_error": err, "pool_gb": round(st["pool_bytes"] / 1e9, 3)}))
    dist.barrier()
    s.close()
    dist.destroy_process_group()


main()
