"""Many-RHS sharding across the GPUs of one node: one process per GPU, columns of B split into
contiguous blocks, no data-path collective for the solves themselves.  torch.distributed (RCCL on GPUs, gloo in the CPU
tests) is used for barriers, the max-over-ranks timing reduction, optionally gathering X on rank 0, and -- when the
factorisation is done once instead of on every rank -- for the broadcast of the numeric factor over xGMI."""
import ctypes

import numpy as np


class _DeviceBytes:
    """A raw device allocation seen as a 1-D uint8 array through __cuda_array_interface__ (torch wraps it without a copy)."""

    def __init__(self, ptr, nbytes):
        self.__cuda_array_interface__ = {"shape": (int(nbytes),), "typestr": "|u1", "data": (int(ptr), False), "version": 2}


def _as_tensor(ptr, nbytes, device):
    import torch
    if device is None or str(device) == "cpu":  # emulated library / host buffers: the "device" pointers are host pointers
        buf = (ctypes.c_uint8 * nbytes).from_address(ptr)
        return torch.from_numpy(np.ctypeslib.as_array(buf))
    return torch.as_tensor(_DeviceBytes(ptr, nbytes), device=device)


def _spread_slices(ptr, nbytes, dist, src, device, rank, world):
    """The two point-to-point steps of solver_hipmf_broadcast_factor (interface_hipmf.cpp, round 6) on a torch.distributed group: the part
    [ptr, ptr + nbytes) is cut into `world` slices of whole 512-byte units; A: the source sends slice r to rank r; B: every rank sends ITS
    slice to every other rank but the source (the source: its own slice to everybody).  On xGMI every transfer of a step has a link of its
    own: 2 F / (N x link rate) instead of the F / link rate of a ring.  Returns the bytes covered (the tail goes by broadcast)."""
    sl = (nbytes // world) & ~511
    if sl <= 0:
        return 0
    ops = []
    if rank == src:
        ops = [dist.P2POp(dist.isend, _as_tensor(ptr + r * sl, sl, device), r) for r in range(world) if r != src]
    else:
        ops = [dist.P2POp(dist.irecv, _as_tensor(ptr + rank * sl, sl, device), src)]
    for req in dist.batch_isend_irecv(ops):
        req.wait()
    ops = []
    for q in range(world):
        if q == rank:
            continue
        if q != src:
            ops.append(dist.P2POp(dist.isend, _as_tensor(ptr + rank * sl, sl, device), q))
        if rank != src:
            ops.append(dist.P2POp(dist.irecv, _as_tensor(ptr + q * sl, sl, device), q))
    for req in dist.batch_isend_irecv(ops):
        req.wait()
    return sl * world


def broadcast_factor(solver, d_values, dist, src=0, device=None, chunk_bytes=256 << 20, slices=True, slice_min_bytes=64 << 20):
    """Numeric factor of `solver` (rank `src` has factorised; every rank has run `initialize` on the same structure, which
    is deterministic) -> all ranks, in place, straight between the solvers' own device buffers -- the FOUR parts of
    solver_hipmf_factor_parts: persistent part of the front pool, local row interchanges, row scaling, pivots (the D of the
    L D L^T fronts, the determinant and rcond come from them) -- in chunks of `chunk_bytes` (ring collectives over xGMI are per-link bound, large
    messages keep the links busy); with three or more ranks a part of at least `slice_min_bytes` travels as slices over all links
    (_spread_slices).  Afterwards the other ranks adopt the factor; `d_values` = device pointer of the matrix
    values of the calling rank (for the refinement SpMV).  Returns the number of bytes broadcast."""
    total = 0
    world, rank = dist.get_world_size(), dist.get_rank()
    for ptr, nbytes in solver.factor_buffers():
        done = 0
        if slices and world >= 3 and nbytes >= slice_min_bytes:
            done = _spread_slices(ptr, nbytes, dist, src, device, rank, world)
        for off in range(done, nbytes, chunk_bytes):
            n = min(chunk_bytes, nbytes - off)
            dist.broadcast(_as_tensor(ptr + off, n, device), src=src)
        total += nbytes
    if device is not None and str(device) != "cpu":
        import torch
        torch.cuda.synchronize(device)
    if dist.get_rank() != src:
        code = solver.adopt_factor(d_values)
        if code != 0:
            raise RuntimeError("solver_hipmf_adopt_factor failed with code %d" % code)
    return total


def rhs_block(nrhs_total, world_size, rank):
    """Contiguous block [start, start + count) of right-hand sides owned by `rank` (sizes differ by at most one)."""
    if nrhs_total < 0 or world_size < 1 or not (0 <= rank < world_size):
        raise ValueError("invalid sharding request")
    base, extra = divmod(nrhs_total, world_size)
    count = base + (1 if rank < extra else 0)
    start = rank * base + min(rank, extra)
    return start, count


def max_over_ranks(value, dist=None, device=None):
    """max of a python float over all ranks (identity without an initialised process group)."""
    if dist is None or not dist.is_initialized():
        return float(value)
    import torch
    t = torch.tensor([float(value)], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def solve_sharded(local_solve, B, dist=None, gather=True):
    """B: (nrhs_total, n) array known to every rank (rows = right-hand sides).  Every rank solves its block with
    `local_solve(block) -> X_block`; with gather=True rank 0 returns the full (nrhs_total, n) solution, other ranks None."""
    world = dist.get_world_size() if dist is not None and dist.is_initialized() else 1
    rank = dist.get_rank() if world > 1 else 0
    start, count = rhs_block(B.shape[0], world, rank)
    x_local = local_solve(B[start:start + count]) if count > 0 else np.zeros((0, B.shape[1]))
    if world == 1 or not gather:
        return x_local
    parts = [None] * world if rank == 0 else None
    dist.gather_object((start, np.asarray(x_local)), parts, dst=0)
    if rank != 0:
        return None
    X = np.zeros_like(B, dtype=np.float64)
    for s, xb in parts:
        X[s:s + xb.shape[0]] = xb
    return X
