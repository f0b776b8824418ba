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


def _bcast_worker(rank, world, port, out, emu):
    sys.path.insert(0, ROOT)
    import torch.distributed as dist
    from russell_amd import problems as P
    from russell_amd.backend import Hipmf
    from russell_amd.distributed import broadcast_factor, rhs_block
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    n, rp, ci, v = P.poisson2d(30, 26)
    B = np.random.default_rng(7).standard_normal((6, n))
    s = Hipmf(emu)
    assert s.initialize(n, rp, ci) == 0
    d_v = s.dev_alloc(v.nbytes)
    s.h2d(d_v, v)
    if rank == 0:
        assert s.factorize_device(d_v) == 0  # only the source rank factorises
    nbytes = broadcast_factor(s, d_v, dist, src=0, device=None, chunk_bytes=1 << 16)
    start, count = rhs_block(B.shape[0], world, rank)
    X = np.array([s.solve(B[j]) for j in range(start, start + count)])
    parts = [None] * world if rank == 0 else None
    dist.gather_object((start, X), parts, dst=0)
    if rank == 0:
        full = np.zeros_like(B)
        for st, xb in parts:
            full[st:st + xb.shape[0]] = xb
        ref = np.array([s.solve(b) for b in B])  # rank 0 solves everything with its own factor
        resid = max(float(np.max(np.abs(P.csr_matvec(n, rp, ci, v, full[j]) - B[j]))) for j in range(B.shape[0]))
        np.save(out, np.array([float(np.max(np.abs(full - ref))), resid, float(nbytes), float(s.stats()["pool_bytes"])]))
    dist.barrier()
    s.close()
    dist.destroy_process_group()


def test_factor_broadcast_and_adopt_gloo(tmp_path, emu_lib):
    # factorise on one rank, broadcast the packed factor, every rank solves its block of right-hand sides.
    # Two processes over gloo with the emulated backend (its "device" buffers are host memory): the adopting rank must produce
    # bit-identical solutions to the rank that factorised.
    import torch.multiprocessing as mp
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    out = str(tmp_path / "bcast.npy")
    mp.spawn(_bcast_worker, args=(2, port, out, emu_lib), nprocs=2, join=True)
    diff, resid, nbytes, pool = np.load(out)
    assert diff == 0.0 and resid < 1e-11
    assert nbytes > pool  # pool + interchanges + row scaling


def _shard8_worker(rank, world, port, out, emu, nrhs_total):
    sys.path.insert(0, ROOT)
    import torch.distributed as dist
    from russell_amd import problems as P
    from russell_amd.backend import Hipmf
    from russell_amd.distributed import broadcast_factor, max_over_ranks, rhs_block
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    n, rp, ci, v = P.poisson2d(26, 22)
    # independent random columns, the same on every rank (one generator per column)
    B = np.stack([np.random.default_rng([20260927, j]).standard_normal(n) for j in range(nrhs_total)])
    s = Hipmf(emu)
    assert s.initialize(n, rp, ci) == 0
    d_v = s.dev_alloc(v.nbytes)
    s.h2d(d_v, v)
    if rank == 0:
        assert s.factorize_device(d_v) == 0  # ONE rank factorises ...
    else:
        d_w = s.dev_alloc(v.nbytes)  # ... the others hold the factor of ANOTHER matrix until the broadcast replaces it
        s.h2d(d_w, 3.0 * v)
        assert s.factorize_device(d_w) == 0
        s.dev_free(d_w)
    # (round 6: the pool travels as eight slices in two point-to-point steps -- scatter from the root, then everybody's slice to everybody --
    #  the path solver_hipmf_broadcast_factor takes over xGMI; the small parts and the tail by broadcast)
    broadcast_factor(s, d_v, dist, src=0, device=None, chunk_bytes=1 << 15, slices=True, slice_min_bytes=1 << 14)
    # this rank's columns, solved in place of the WHOLE n x nrhs arrays through the C-ABI's sharded entry point
    d_b, d_x = s.dev_alloc(B.nbytes), s.dev_alloc(B.nbytes)
    s.h2d(d_b, B)
    first, count = s.solve_many_sharded(d_x, d_b, nrhs_total, world, rank)
    assert (first, count) == rhs_block(nrhs_total, world, rank)
    X = np.zeros_like(B)
    s.d2h(X, d_x)
    worst = 0.0
    for j in range(first, first + count):
        worst = max(worst, float(np.max(np.abs(P.csr_matvec(n, rp, ci, v, X[j]) - B[j])) / (np.max(np.abs(v)) + 1.0)))
    worst = max_over_ranks(worst, dist)
    parts = [None] * world if rank == 0 else None
    dist.gather_object((first, count, X[first:first + count].copy()), parts, dst=0)
    if rank == 0:
        full = np.zeros_like(B)
        covered = 0
        for f0, c0, xb in parts:
            full[f0:f0 + c0] = xb
            covered += c0
        ref = np.array([s.solve(B[j]) for j in range(0, nrhs_total, 37)])  # single solves with the factorising rank's own factor
        np.save(out, np.array([worst, float(covered), float(np.max(np.abs(full[0:nrhs_total:37] - ref)) / np.max(np.abs(ref)))]))
    dist.barrier()
    s.dev_free(d_b), s.dev_free(d_x), s.dev_free(d_v)
    s.close()
    dist.destroy_process_group()


@pytest.mark.parametrize("nrhs_total", [256, 257])
def test_many_rhs_sharded_over_eight_ranks_gloo(tmp_path, emu_lib, nrhs_total):
    # north_star's split at its real width: EIGHT ranks, 256 right-hand sides (and 257: blocks that differ by one),
    # one rank factorises, the factor travels (gloo here, ncclBroadcast on the GPUs), every rank solves its contiguous block through
    # solver_hipmf_solve_many_sharded with blocks of 16 columns.  Every column's residual is checked on the rank that solved it.
    import torch.multiprocessing as mp
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    out = str(tmp_path / "shard8.npy")
    mp.spawn(_shard8_worker, args=(8, port, out, emu_lib, nrhs_total), nprocs=8, join=True)
    worst, covered, diff = np.load(out)
    assert covered == nrhs_total
    assert worst <= 1e-12  # relative_error of VerifyLinSys, every column of every rank
    assert diff <= 1e-12   # blocked solves on an adopting rank against single solves on the factorising rank (equal to rounding)
