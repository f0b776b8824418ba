"""Multi-GPU entry points of the C-ABI (include/russell_hipmf.h): RCCL communicator helpers, factor broadcast, sharded solve.

One GPU: a communicator of one rank exercises the whole call sequence (dlopen of RCCL, ncclCommInitRank, ncclBroadcast in place,
adopt bookkeeping).  Two or more GPUs: two processes, rank 0 factorises, rank 1 receives the factor over RCCL and must reproduce
rank 0's solutions bit for bit (skipped when the box has one GPU; the CPU twin of this test runs over gloo with the emulated
backend, tests/test_distributed_cpu.py)."""
import ctypes as C
import os
import subprocess
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

WORKER = r"""
import ctypes as C, os, sys
import numpy as np
sys.path.insert(0, %(root)r)
from russell_amd import problems as P
from russell_amd import _capi
from russell_amd.backend import Hipmf
rank, nranks, idfile, outfile = int(sys.argv[1]), int(sys.argv[2]), sys.argv[3], sys.argv[4]
lib = _capi.load()
assert lib.hipmf_set_device(rank %% max(lib.hipmf_device_count(), 1)) == 0
idbuf = (C.c_uint8 * 128)()
if rank == 0:
    assert lib.hipmf_comm_unique_id(idbuf) == 0
    with open(idfile + ".tmp", "wb") as fh:
        fh.write(bytes(idbuf))
    os.rename(idfile + ".tmp", idfile)
else:
    import time
    for _ in range(600):
        if os.path.exists(idfile):
            break
        time.sleep(0.1)
    C.memmove(idbuf, open(idfile, "rb").read(), 128)
comm = C.c_void_p()
assert lib.hipmf_comm_init_rank(C.byref(comm), nranks, idbuf, rank) == 0
n, rp, ci, v = P.poisson2d(90, 80)
nrhs = 11
B = np.asfortranarray(np.random.default_rng(4).standard_normal((n, nrhs)))
s = Hipmf()
assert s.initialize(n, rp, ci) == 0
if rank == 0:
    assert s.factorize(v) == 0
sec, nbytes = s.broadcast_factor(comm, 0, rank)
assert nbytes > 0
d_b, d_x = s.dev_alloc(B.nbytes), s.dev_alloc(B.nbytes)
s.h2d(d_b, np.ascontiguousarray(B.T))
first, count = s.solve_many_sharded(d_x, d_b, nrhs, nranks, rank)
X = np.zeros((nrhs, n))
s.d2h(X, d_x)
# reference: rank 0's own factor, one blocked solve of all columns through the host entry point (a column's result does not depend
# on which other columns share its block) and single solves (scalar kernels: equal to rounding)
if rank == 0:
    ref = s.solve_many(np.ascontiguousarray(B.T))
    np.save(outfile + ".ref.npy", ref)
    np.save(outfile + ".ref1.npy", np.stack([s.solve(B[:, j].copy()) for j in range(nrhs)]))
np.save(outfile + ".%%d.npy" %% rank, np.concatenate([[first, count], X[first:first + count].ravel()]))
lib.hipmf_comm_destroy(comm)
s.close()
print("RANK-OK", rank, first, count, "%%.3f s %%d bytes" %% (sec, nbytes))
"""


def _run(nranks, tmp_path):
    script = tmp_path / "worker.py"
    script.write_text(WORKER % {"root": ROOT})
    idfile, out = str(tmp_path / "nccl_id"), str(tmp_path / "x")
    procs = [subprocess.Popen([sys.executable, str(script), str(r), str(nranks), idfile, out], stdout=subprocess.PIPE, stderr=subprocess.STDOUT,
                              env=dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")) for r in range(nranks)]
    outs = [p.communicate(timeout=600)[0].decode() for p in procs]
    for r, (p, o) in enumerate(zip(procs, outs)):
        assert p.returncode == 0 and "RANK-OK" in o, "rank %d:\n%s" % (r, o[-3000:])
    ref = np.load(out + ".ref.npy")
    n = ref.shape[1]
    covered = 0
    for r in range(nranks):
        a = np.load(out + ".%d.npy" % r)
        first, count = int(a[0]), int(a[1])
        X = a[2:].reshape(count, n)
        # blocked solves give the same bits whichever rank holds the (received) factor and however the columns are grouped
        # (a rank with a single column runs the scalar kernels: equal to rounding), and agree with single solves to rounding
        if count > 1:
            assert np.array_equal(X, ref[first:first + count]), "rank %d block differs" % r
        ref1 = np.load(out + ".ref1.npy")
        assert np.max(np.abs(X - ref1[first:first + count])) <= 1e-12 * np.max(np.abs(ref1)), "rank %d block differs from single solves" % r
        covered += count
    assert covered == ref.shape[0]


@pytest.mark.gpu
def test_rccl_broadcast_and_sharded_solve_one_rank(tmp_path):
    _run(1, tmp_path)


@pytest.mark.gpu
def test_rccl_broadcast_and_sharded_solve_two_gpus(tmp_path):
    from russell_amd import _capi
    if _capi.load().hipmf_device_count() < 2:
        pytest.skip("needs two GPUs (the driver's multi-GPU node); one-rank variant and the gloo CPU twin cover the logic")
    _run(2, tmp_path)


@pytest.mark.gpu
def test_bench_distributed_branch_with_one_rank():
    """bench.py's many-RHS section under BENCH_FORCE_DIST=1: a torch `nccl` (= RCCL) process group of ONE rank drives the same code as
    `--gpus N` -- unique id through the group, hipmf_comm_init_rank, solver_hipmf_broadcast_factor (plan check, MIN all-reduce,
    ncclBroadcast of the factor parts), the sharded solve -- with nranks as data.  What a node with N GPUs adds is ranks, not code."""
    import json
    import socket

    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    env = dict(os.environ, BENCH_FORCE_DIST="1", RANK="0", LOCAL_RANK="0", WORLD_SIZE="1", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port),
               HSA_ENABLE_IPC_MODE_LEGACY="0")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--steps", "2", "--warmup", "1", "--no-cpu-baseline", "--grid", "300",
                        "--nrhs", "32", "--grid3d", "0"], env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    line = [ln for ln in r.stdout.splitlines() if ln.startswith("{")][-1]
    out = json.loads(line)
    many = out["many_rhs"]
    assert "error" not in many, many
    assert many["nrhs_total"] == 32 and many["rhs_per_gpu"] == 32
    bc = many["broadcast"]
    assert "error" not in bc, bc
    assert bc["broadcast_bytes"] > 0 and bc["broadcast_ms"] > 0.0
    assert bc["max_relative_error_all_columns"] < 1e-10 and many["max_relative_error_all_columns"] < 1e-10
    assert many["roofline"]["fused_solve_fallbacks"] == 0
    assert out["relative_error"] < 1e-10
