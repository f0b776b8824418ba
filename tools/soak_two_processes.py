#!/usr/bin/env python3
"""Two PROCESSES on one GPU (round 6): each solves the same kind of system again and again through the dependency-driven kernels; the
per-device gate is a mutex inside a process and an advisory file lock between processes (numeric.cpp, DeviceGate).  The parent starts two
workers and prints one JSON object: solves, largest / median solve time, hand-off fallbacks and waits at the gate of either worker.
usage: soak_two_processes.py [N [SOLVES]]          worker: soak_two_processes.py --worker N SOLVES"""
import json
import os
import subprocess
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def worker(N, solves):
    from russell_amd import problems as P
    from russell_amd.backend import Hipmf
    n, rp, ci, v = P.poisson2d(N)
    xs = P.manufactured_solution(n)
    b = P.csr_matvec(n, rp, ci, v, xs)
    s = Hipmf()
    assert s.initialize(n, rp, ci) == 0
    d_v, d_b, d_x = s.dev_alloc(v.nbytes), s.dev_alloc(b.nbytes), s.dev_alloc(b.nbytes)
    s.h2d(d_v, v), s.h2d(d_b, b)
    assert s.factorize_device(d_v) == 0
    s.solve_device(d_x, d_b)
    times = []
    t_end = time.perf_counter() + 60.0
    for _ in range(solves):
        t0 = time.perf_counter()
        s.solve_device(d_x, d_b)
        s.lib.hipmf_device_synchronize()
        times.append(time.perf_counter() - t0)
        if time.perf_counter() > t_end:
            break
    x = np.zeros(n)
    s.d2h(x, d_x)
    out = {"solves": len(times), "median_ms": float(np.median(times) * 1e3), "max_ms": float(np.max(times) * 1e3),
           "fallbacks": s.counter("fused_fallbacks"), "gate_waits": s.counter("gate_waits"), "max_error": float(np.max(np.abs(x - xs)))}
    s.close()
    print(json.dumps(out))


def main():
    if len(sys.argv) > 1 and sys.argv[1] == "--worker":
        worker(int(sys.argv[2]), int(sys.argv[3]))
        return
    N = int(sys.argv[1]) if len(sys.argv) > 1 else 500
    solves = int(sys.argv[2]) if len(sys.argv) > 2 else 1500
    procs = [subprocess.Popen([sys.executable, os.path.abspath(__file__), "--worker", str(N), str(solves)], stdout=subprocess.PIPE, text=True) for _ in range(2)]
    res = []
    for p in procs:
        out, _ = p.communicate(timeout=600)
        lines = [ln for ln in out.splitlines() if ln.startswith("{")]
        res.append(json.loads(lines[-1]) if lines and p.returncode == 0 else {"error": p.returncode})
    print(json.dumps({"grid": N, "workers": res, "process_gate": os.environ.get("HIPMF_PROCESS_GATE", "1")}))


if __name__ == "__main__":
    main()
