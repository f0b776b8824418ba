"""world_size-2 gloo test of the many-RHS sharding path (no GPU): the per-rank solver is the CPU oracle here,
what is under test is the block partition, the gather and the max-over-ranks reduction used by bench.py."""
import os
import socket
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_rhs_block_partition_is_exact_cover():
    from russell_amd.distributed import rhs_block
    for total in (0, 1, 7, 8, 256, 257):
        for world in (1, 2, 3, 8):
            blocks = [rhs_block(total, world, r) for r in range(world)]
            assert blocks[0][0] == 0 and sum(c for _, c in blocks) == total
            for (s0, c0), (s1, _) in zip(blocks, blocks[1:]):
                assert s0 + c0 == s1
            assert max(c for _, c in blocks) - min(c for _, c in blocks) <= 1
    with pytest.raises(ValueError):
        rhs_block(4, 2, 2)


def _worker(rank, world, port, out):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import torch.distributed as dist
    import oracle_lib as O
    from russell_amd import problems as P
    from russell_amd.distributed import max_over_ranks, solve_sharded
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    n, rp, ci, v = P.poisson2d(12, 11)
    rows = np.repeat(np.arange(n, dtype=np.int32), np.diff(rp))
    cp, ri, vx = O.coo_to_csc(n, n, rows, ci, v)
    lu = O.OracleLU(n, cp, ri, vx)
    B = np.random.default_rng(20260927).standard_normal((5, n))
    X = solve_sharded(lambda blk: np.vstack([lu.solve(b) for b in blk]), B, dist)
    t = max_over_ranks(1.0 + rank, dist)
    if rank == 0:
        ref = np.vstack([lu.solve(b) for b in B])
        np.save(out, np.array([float(np.max(np.abs(X - ref))), t]))
    dist.barrier()
    dist.destroy_process_group()


def test_sharded_many_rhs_gloo(tmp_path):
    import torch.multiprocessing as mp
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    out = str(tmp_path / "res.npy")
    mp.spawn(_worker, args=(2, port, out), nprocs=2, join=True)
    err, tmax = np.load(out)
    assert err == 0.0 and tmax == 2.0
