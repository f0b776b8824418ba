#!/usr/bin/env python3
"""bench.py -- factorize+solve of the 1M-DOF 2D 5-point Poisson matrix (BASELINE.json configs[1]) on MI355X.

A "step" (the headline `value`) is one pass of the hot path on every rank: numeric LU factorisation (values already resident in
HBM) followed by one solve (rhs resident in HBM, default iterative refinement) of the general-storage matrix -- the call the
reference's Radau5 / Newton callers repeat, and the one the north_star compares with UMFPACK.  Prints ONE JSON line (rank 0).

  python bench.py --gpus 1 --steps 5 --warmup 2
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N --steps K --warmup W

Beside the headline the same line carries (all measured in this run, outside the K timed steps):
  symmetric    the same matrix handed over as its lower triangle (Sym::YesLower, what russell_pde gives a GPU genie):
               L D L^T on the tiled fronts;
  poisson3d    3D 7-point Poisson 100^3 as its lower triangle (one GPU, rank 0): factorize ms / LU-equivalent TFLOP/s, SpTRSV GB/s and
               fraction of the HBM peak -- the regime where the level-to-level latency of the 2D headline amortises
  host_api     solver_hipmf_factorize / _solve with HOST pointers through the mirror of the Rust layer, i.e. including the
               COO -> CSR value refresh and the H2D / D2H copies the reference's boundary includes (interface_cudss.cu:424,524,553);
  many_rhs     the north_star split: 256 right-hand sides sharded over the ranks; one rank factorises, the factor travels over
               RCCL (solver_hipmf_broadcast_factor), every rank solves its block -- with the factorize / broadcast / solve split,
               beside the replicated-factorisation alternative;
  config3      BASELINE config 3: data/bbmat.mtx / data/af_shell10.mtx through the reference's harness when the files exist (there is no
               network here: they are absent), else the two stand-ins at their published sizes (factorize / solve ms, perturbed pivots,
               relative_error);
  config4      BASELINE config 4's matrix (3D 7-point Poisson 200^3 as its lower triangle) with the LARGEST single-GPU shard of its 256
               right-hand sides (32 columns = one rank of eight), when 240 GB of device memory are free;
  config5      BASELINE config 5: russell_amd/lib/brusselator_pde --npoint 513 (Radau5, real + complex handle on two threads): totals,
               largest factorisation / solve, fallback counters;
  roofline     SpTRSV pass (HBM) of the headline, roofline_factor (FP64 MFMA), fused-solve fallback count;
  cpu_baseline the CPU path on this box's host cores, best available tier: UMFPACK itself (oracle/umfpack_probe.c, when a
               libumfpack can be loaded), else Intel MKL PARDISO (threaded, phases timed apart, in a child process; labelled: NOT
               UMFPACK), else SuperLU through scipy (sequential), else the repo's own CPU port.
"""
import argparse
import ctypes
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
NRHS_TOTAL = 256           # BASELINE config 4 / north_star: 256 right-hand sides over the GPUs of the node


def sptrsv_bytes(st, n, k=1):
    """(nnzL + nnzU)*(8+4) + 2(n+1)*4 + k*n*8*2*2 with the ACTUAL factor sizes."""
    return (st["nnz_l"] + st["nnz_u"]) * 12 + 2 * (n + 1) * 4 + k * n * 8 * 4


def measured_traffic(grid, st):
    """HBM bytes per SpTRSV pass pair from the PMC passes committed under profiles/ (rocprofv3 cannot run inside the timed loop: the
    counters were collected with this same command on the same workload).  The file is STAMPED with the factor it was taken on (nnz(L),
    nnz(U), supernodes, solve launches): a figure whose stamp does not match the build that is being timed is refused (None) -- the
    committed line of round 4 carried the constant of an older tree."""
    if grid != 1000:
        return None, None
    names = sorted((f for f in os.listdir(os.path.join(ROOT, "profiles")) if f.endswith("_sptrsv_traffic.json")), reverse=True)
    for name in names:
        try:
            with open(os.path.join(ROOT, "profiles", name)) as fh:
                t = json.load(fh)
        except (OSError, ValueError):
            continue
        stamp = t.get("stamp")
        if not stamp:
            return None, "profiles/%s carries no stamp of the build it was taken on: refused" % name
        want = {"nnz_l": st["nnz_l"], "nnz_u": st["nnz_u"], "nsuper": st["nsuper"], "solve_launches": st["solve_launches"]}
        if any(stamp.get(k) != val for k, val in want.items()):
            return None, "profiles/%s was taken on another build (stamp %r, this run %r): refused" % (name, stamp, want)
        return t["traffic_bytes_per_pass"], "profiles/%s: %s" % (name, t["source"])
    return None, None


def lower_triangle(n, rp, ci, v):
    rows = np.repeat(np.arange(n, dtype=np.int32), np.diff(rp))
    keep = ci <= rows
    lrp = np.concatenate([[0], np.cumsum(np.bincount(rows[keep], minlength=n))]).astype(np.int32)
    return lrp, np.ascontiguousarray(ci[keep]), np.ascontiguousarray(v[keep])


def residual_metric(n, rp, ci, v, x, b):
    """relative_error of VerifyLinSys (verify_lin_sys.rs:60-96): |A x - b|_inf / (max|a| + 1)"""
    r = np.zeros(n)
    np.add.at(r, np.repeat(np.arange(n), np.diff(rp)), v * x[ci])
    return float(np.max(np.abs(r - b)) / (np.max(np.abs(v)) + 1.0))


def _pardiso_probe(grid):
    """Child process of pardiso_baseline (a crash inside a third-party library must not cost the benchmark its JSON line): MKL PARDISO on
    the 2D Poisson matrix of the headline at a few thread counts, every count on a handle of its own (analysis 11, numeric 22, solve 33,
    then numeric + solve again = the repeat call).  Prints one JSON object."""
    from russell_amd import problems as P
    n, rp, ci, v = P.poisson2d(grid)
    b = P.csr_matvec(n, rp, ci, v, P.manufactured_solution(n))
    lib = None
    for name in ("libmkl_rt.so.2", "libmkl_rt.so.1", "libmkl_rt.so", "/opt/conda/lib/libmkl_rt.so.2", "/opt/conda/lib/libmkl_rt.so.1", "/opt/conda/lib/libmkl_rt.so"):
        try:
            lib = ctypes.CDLL(name, mode=ctypes.RTLD_GLOBAL)
            break
        except OSError:
            pass
    if lib is None or not hasattr(lib, "pardiso"):
        print(json.dumps({"error": "libmkl_rt cannot be loaded"}))
        return
    I = ctypes.c_int32
    try:
        default_threads = int(lib.MKL_Get_Max_Threads())
    except Exception:
        default_threads = 0
    ia, ja, a = np.ascontiguousarray(rp, dtype=np.int32), np.ascontiguousarray(ci, dtype=np.int32), np.ascontiguousarray(v, dtype=np.float64)
    runs = []
    # on a 256-core host the library's default (one thread per physical core) is several times SLOWER than 16 - 32 threads on a matrix of
    # this size: the best count is what the benchmark is set against
    for nt in sorted({t for t in (16, 32, 64) if t <= max(default_threads, 16)} | ({default_threads} if 0 < default_threads <= 16 else set())):
        try:
            lib.MKL_Set_Num_Threads(ctypes.c_int(nt))
        except Exception:
            pass
        pt = (ctypes.c_void_p * 64)()
        iparm = (I * 64)()
        mtype = I(11)  # real, unsymmetric (the reference hands UMFPACK the full matrix, enums.rs:355-365)
        lib.pardisoinit(pt, ctypes.byref(mtype), iparm)
        iparm[34] = 1  # zero-based indices
        iparm[7] = 2   # at most two refinement steps, like UMFPACK_IRSTEP
        x, bb, perm = np.zeros(n), np.ascontiguousarray(b, dtype=np.float64).copy(), np.zeros(n, dtype=np.int32)
        maxfct, mnum, nrhs, msglvl, err, nn = I(1), I(1), I(1), I(0), I(0), I(n)

        def call(phase):
            ph = I(phase)
            t0 = time.perf_counter()
            lib.pardiso(pt, ctypes.byref(maxfct), ctypes.byref(mnum), ctypes.byref(mtype), ctypes.byref(ph), ctypes.byref(nn), a.ctypes.data_as(ctypes.c_void_p),
                        ia.ctypes.data_as(ctypes.c_void_p), ja.ctypes.data_as(ctypes.c_void_p), perm.ctypes.data_as(ctypes.c_void_p), ctypes.byref(nrhs), iparm,
                        ctypes.byref(msglvl), bb.ctypes.data_as(ctypes.c_void_p), x.ctypes.data_as(ctypes.c_void_p), ctypes.byref(err))
            return (time.perf_counter() - t0) * 1e3, err.value

        t = [call(ph) for ph in (11, 22, 33, 22, 33)]
        call(-1)
        if any(e != 0 for _, e in t):
            runs.append({"threads": nt, "error": [e for _, e in t]})
            continue
        runs.append({"threads": nt, "analysis": t[0][0], "numeric": t[1][0], "solve": t[2][0], "repeat_numeric": t[3][0], "repeat_solve": t[4][0],
                     "nnz_factor": int(iparm[17]), "relative_error": residual_metric(n, rp, ci, v, x, b)})
    print(json.dumps({"default_threads": default_threads, "runs": runs}))


def pardiso_baseline(grid):
    """A THREADED CPU comparator where one can be loaded: Intel MKL PARDISO through libmkl_rt (present in this image under /opt/conda/lib),
    phases timed apart like the reference's Stopwatch around its three FFI calls (solver_umfpack.rs:282-299,304-324,370-385).  Runs in a
    child process; None when MKL cannot be loaded or the probe fails."""
    import subprocess
    try:
        r = subprocess.run([sys.executable, os.path.abspath(__file__), "--pardiso-probe", str(grid)], capture_output=True, text=True, timeout=240)
        lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
        if r.returncode != 0 or not lines:
            return {"error": "probe exited with %d" % r.returncode}
        d = json.loads(lines[-1])
    except Exception as exc:
        return {"error": repr(exc)}
    if "error" in d:
        return d
    good = [q for q in d["runs"] if "error" not in q]
    if not good:
        return {"error": "pardiso returned errors: %r" % d["runs"]}
    best = min(good, key=lambda q: q["repeat_numeric"] + q["repeat_solve"])
    best["sweep"] = [(q["threads"], round(q.get("repeat_numeric", -1.0), 1), round(q.get("repeat_solve", -1.0), 1)) for q in d["runs"]]
    best["default_threads"] = d["default_threads"]
    return best


def superlu_tier(n, rp, ci, v, b):
    """SURVEY.md 8(d) tier 2: SuperLU through scipy (`splu`, permc_spec MMD_AT_PLUS_A for the symmetric pattern), sequential, labelled as
    what it is.  Printed BESIDE whichever tier gives `cpu_baseline.value`: ~15 s of one core at the 1M-DOF matrix."""
    try:
        import scipy.sparse as sp
        import scipy.sparse.linalg as spla
        A = sp.csr_matrix((v, ci, rp), shape=(n, n)).tocsc()
        t0 = time.perf_counter()
        lu = spla.splu(A, permc_spec="MMD_AT_PLUS_A")  # symmetric-pattern ordering, SuperLU's counterpart of UMFPACK's AMD choice
        t1 = time.perf_counter()
        x = lu.solve(b)
        t2 = time.perf_counter()
        return {"kind": "third-party stand-in: SuperLU (scipy.sparse.linalg.splu), sequential -- NOT the reference's UMFPACK", "cores": 1,
                "factorize_ms": round((t1 - t0) * 1e3, 1), "solve_ms": round((t2 - t1) * 1e3, 1), "total_ms": round((t2 - t0) * 1e3, 2),
                "nnz_factor": int(lu.L.nnz + lu.U.nnz), "relative_error": residual_metric(n, rp, ci, v, x, b),
                "sample": "SuperLU, permc_spec MMD_AT_PLUS_A, one thread, on the SAME %d-DOF matrix: factorize (ordering + symbolic + numeric) "
                          "%.1f ms + solve %.1f ms, nnz(L+U) %d, relative_error %.1e"
                          % (n, (t1 - t0) * 1e3, (t2 - t1) * 1e3, int(lu.L.nnz + lu.U.nnz), residual_metric(n, rp, ci, v, x, b))}
    except Exception as exc:
        return {"error": repr(exc)}


# ---------------------------------------------------------------------------------------------------------------------
# CPU baseline (rank 0, N = 1 only): the only part of this file that touches oracle/
def cpu_baseline(n, rp, ci, v, b, perm, tier, grid=0):
    rows = np.repeat(np.arange(n, dtype=np.int32), np.diff(rp))
    ncores = os.cpu_count() or 0
    tried = []
    UMFPACK_NAMES = ["libumfpack.so", "libumfpack.so.6", "libumfpack.so.5", "libumfpack.so.7"]  # (what oracle/umfpack_probe.c hands to dlopen, in this order)
    if tier in ("auto", "umfpack"):
        so = os.path.join(ROOT, "oracle", "libumfpack_probe.so")
        if os.path.exists(so):
            import oracle_lib as O
            cp, ri, vx = O.coo_to_csc(n, n, rows, ci, v)
            lib = ctypes.CDLL(so)
            i32p = np.ctypeslib.ndpointer(dtype=np.int32, flags="C_CONTIGUOUS")
            f64p = np.ctypeslib.ndpointer(dtype=np.float64, flags="C_CONTIGUOUS")
            lib.umfpack_probe.restype = ctypes.c_int
            lib.umfpack_probe.argtypes = [ctypes.c_int32, i32p, i32p, f64p, f64p, f64p, f64p, ctypes.c_char_p, ctypes.c_int32]
            x, sec, name = np.zeros(n), np.zeros(3), ctypes.create_string_buffer(256)
            rc = lib.umfpack_probe(n, cp, ri, vx, np.ascontiguousarray(b), x, sec, name, 256)
            if rc == 0:
                return {"value": round((sec[1] + sec[2]) * 1e3, 2), "unit": "ms", "cores": int(os.environ.get("OMP_NUM_THREADS", ncores) or ncores),
                        "kind": "reference",
                        "sample": "UMFPACK (%s, dlopen) with the reference shim's call sequence and controls (strategy AUTO, ordering AMD, scale SUM; "
                                  "interface_umfpack.c:47,99-109,167,229) on the SAME %d-DOF matrix: symbolic %.1f ms, numeric %.1f ms, solve %.1f ms, "
                                  "relative_error %.1e; BLAS threads as configured on this host (%d cores)"
                                  % (name.value.decode(), n, sec[0] * 1e3, sec[1] * 1e3, sec[2] * 1e3, residual_metric(n, rp, ci, v, x, b), ncores)}
            tried.append("UMFPACK: no libumfpack can be loaded on this box (umfpack_probe rc %d; dlopen tried %s)" % (rc, ", ".join(UMFPACK_NAMES)))
        else:
            tried.append("UMFPACK: oracle/libumfpack_probe.so not built")
    if tier in ("auto", "pardiso"):
        r = pardiso_baseline(grid) if grid > 0 else {"error": "no grid"}
        if "error" not in r:
            return {"value": round(r["repeat_numeric"] + r["repeat_solve"], 2), "unit": "ms", "cores": r["threads"],
                    "kind": "third-party stand-in: Intel MKL PARDISO (libmkl_rt, dlopen, child process), threaded -- NOT the reference's UMFPACK",
                    "phases_ms": {"analysis": round(r["analysis"], 1), "numeric": round(r["numeric"], 1), "solve": round(r["solve"], 1),
                                  "repeat_numeric": round(r["repeat_numeric"], 1), "repeat_solve": round(r["repeat_solve"], 1)},
                    # (the first numeric / solve of a process also pay the library's thread start-up: the faster of the two calls counts)
                    "one_shot_ms": round(r["analysis"] + min(r["numeric"], r["repeat_numeric"]) + min(r["solve"], r["repeat_solve"]), 1),
                    "threaded_tier": "MKL PARDISO, best of the thread counts %s = %d threads (library default on this host: %d)"
                                     % ([t[0] for t in r["sweep"]], r["threads"], r["default_threads"]),
                    "thread_sweep_repeat_numeric_solve_ms": r["sweep"],
                    "umfpack_libs_tried": UMFPACK_NAMES, "mkl_libs_tried": ["libmkl_rt.so.2", "libmkl_rt.so.1", "libmkl_rt.so", "/opt/conda/lib/libmkl_rt.so*"],
                    "sample": "MKL PARDISO (mtype 11, phases 11 / 22 / 33, <= 2 refinement steps) on the SAME %d-DOF matrix with %d threads: analysis "
                              "%.1f ms, numeric %.1f ms, solve %.1f ms; numeric + solve again on the same handle (the repeat call `value` is set "
                              "against) %.1f + %.1f ms; nnz(L+U) %d, relative_error %.1e; host has %d cores; %s"
                              % (n, r["threads"], r["analysis"], r["numeric"], r["solve"], r["repeat_numeric"], r["repeat_solve"], r["nnz_factor"],
                                 r["relative_error"], ncores, "; ".join(tried))}
        tried.append("MKL PARDISO: %s" % r["error"])
    if tier in ("auto", "superlu"):
        r = superlu_tier(n, rp, ci, v, b)
        if "error" not in r:
            return {"value": r["total_ms"], "unit": "ms", "cores": 1, "kind": r["kind"],
                    "sample": r["sample"] + "; host has %d cores; %s" % (ncores, "; ".join(tried))}
        tried.append("SuperLU: %s" % r["error"])  # scipy missing or out of memory: fall through to the port
    import oracle_lib as O
    cp, ri, vx = O.coo_to_csc(n, n, rows, ci, v)
    t0 = time.perf_counter()
    lu = O.OracleLU(n, cp, ri, vx, q=perm)
    t1 = time.perf_counter()
    x = lu.solve(b, nrefine=2)
    t2 = time.perf_counter()
    return {"value": round((t2 - t0) * 1e3, 2), "unit": "ms", "cores": 1, "kind": "port",
            "sample": "oracle/oracle.c left-looking LU (threshold pivoting, SUM scaling, <=2 refinement steps) on the SAME %d-DOF matrix with the "
                      "same fill-reducing ordering: factorize %.1f ms + solve %.1f ms, relative_error %.1e; host has %d cores; %s"
                      % (n, (t1 - t0) * 1e3, (t2 - t1) * 1e3, residual_metric(n, rp, ci, v, x, b), ncores, "; ".join(tried))}


def config3_section(Hipmf, P, lib):
    """BASELINE config 3 (SuiteSparse bbmat / af_shell10, `bin/solve_matrix_market.rs:97-305`).  The real files are not in the tree and
    there is no network: when data/bbmat.mtx / data/af_shell10.mtx exist they go through the reference's harness
    (russell_amd/lib/solve_matrix_market -g hipmf), else the stand-ins at the published sizes are built and solved through the C-ABI."""
    import subprocess
    out = {}
    harness = os.path.join(ROOT, "russell_amd", "lib", "solve_matrix_market")
    for name in ("bbmat", "af_shell10"):
        path = os.path.join(ROOT, "data", name + ".mtx")
        if os.path.exists(path) and os.path.exists(harness):
            try:
                r = subprocess.run([harness, "-g", "hipmf", path], capture_output=True, text=True, timeout=900)
                d = json.loads(r.stdout)
                out[name] = {"source": "data/%s.mtx through the reference's harness" % name, "harness": d}
            except Exception as exc:
                out[name] = {"source": "data/%s.mtx" % name, "error": repr(exc)}
    import scipy.sparse as sp

    def run(tag, n, rp, ci, v, A, kw, note):
        xs = P.manufactured_solution(n)
        b = A @ xs
        s = Hipmf()
        t0 = time.perf_counter()
        code = s.initialize(n, rp, ci, **kw)
        t_init = time.perf_counter() - t0
        if code != 0:
            s.close()
            out[tag] = {"error": "initialize returned %d" % code}
            return
        d_v, d_b, d_x = s.dev_alloc(v.nbytes), s.dev_alloc(b.nbytes), s.dev_alloc(b.nbytes)
        s.h2d(d_v, v), s.h2d(d_b, b)
        tf = ts = 0.0
        for rep_i in range(3):  # (the first repetition warms the code objects up)
            ta = time.perf_counter()
            code = s.factorize_device(d_v)
            lib.hipmf_device_synchronize()
            tb = time.perf_counter()
            s.solve_device(d_x, d_b)
            lib.hipmf_device_synchronize()
            tc = time.perf_counter()
            if rep_i > 0:
                tf, ts = tf + (tb - ta) / 2.0, ts + (tc - tb) / 2.0
        x = np.zeros(n)
        s.d2h(x, d_x)
        st = s.stats()
        r = A @ x - b
        out[tag] = {"workload": note, "n": int(n), "nnz": int(rp[-1]), "initialize_s": round(t_init, 2), "factorize_ms": round(tf * 1e3, 2),
                    "solve_ms": round(ts * 1e3, 2), "factorize_code": int(code), "perturbed_pivots": int(st["n_perturbed"]), "matched": int(st.get("matched", 0)),
                    "refinement_steps": int(st["refinement_steps"]), "fused_fallbacks": int(st.get("fused_fallbacks", 0)),
                    "relative_error": float(np.max(np.abs(r)) / (np.max(np.abs(v)) + 1.0)),
                    "forward_error": float(np.max(np.abs(x - xs)) / np.max(np.abs(xs)))}
        for ptr in (d_v, d_b, d_x):
            s.dev_free(ptr)
        s.close()

    if "bbmat" not in out:
        # convection-dominated, NOT diagonally dominant, rows scaled over twelve decades, rows shuffled inside every second node: needs the
        # maximum-product matching (values handed to initialize, as the reference's shims do) -- tests/test_round3_gpu.py
        n, rp, ci, v = P.fe_block2d(88, 88, 5, symmetric=False, scale_decades=6.0, shift=0.02)
        A = sp.csr_matrix((v, ci, rp), shape=(n, n))
        rng = np.random.default_rng(38744)
        perm = np.arange(n)
        for node in range(0, n // 5, 2):
            perm[5 * node:5 * node + 5] = 5 * node + rng.permutation(5)
        A = A[perm, :].tocsr()
        A.sort_indices()
        rp2, ci2, v2 = A.indptr.astype(np.int32), A.indices.astype(np.int32), A.data.astype(np.float64)
        run("bbmat_standin", n, rp2, ci2, v2, A, {"values": v2},
            "stand-in for bbmat at its published size (n = 38 720 ~ 38 744, ~45 entries per row): 5 x 5 node blocks, weak diagonal, rows scaled 10^U(-6,6), "
            "rows shuffled inside every second node; general storage, values at initialize (matching)")
    if "af_shell10" not in out:
        n, rp, ci, v = P.fe_block2d(612, 612, 4, symmetric=True)
        A = sp.csr_matrix((v, ci, rp), shape=(n, n))
        lrp, lci, lv = P.lower_triangle(n, rp, ci, v)
        run("af_shell10_standin", n, lrp, lci, lv, A, {"positive_definite": True},
            "stand-in for af_shell10 at its published size (n = 1 498 176 ~ 1 508 065, ~36 entries per row): 4 x 4 node blocks, lower triangle, "
            "positive_definite = 1 (L D L^T)")
        out["af_shell10_standin"]["relative_error"] = out["af_shell10_standin"].get("relative_error")
    out["real_files"] = "absent (no network in this environment): drop bbmat.mtx / af_shell10.mtx under data/ and they are solved through the harness instead"
    return out


def config4_section(Hipmf, P, lib, edge=200, nrhs=32):
    """BASELINE config 4's matrix with the largest single-GPU shard of its 256 right-hand sides (one rank of eight: 32 columns), resident in
    HBM; needs ~240 GB of free device memory (the factor alone takes ~193 GB)."""
    free_b, total_b = ctypes.c_size_t(0), ctypes.c_size_t(0)
    if hasattr(lib, "hipmf_device_mem_info") and lib.hipmf_device_mem_info(ctypes.byref(free_b), ctypes.byref(total_b)) == 0:
        if free_b.value < 240e9:
            return {"skipped": "%.0f GB of device memory free, 240 GB needed" % (free_b.value / 1e9)}
    n, rp, ci, v = P.poisson3d(edge)
    import scipy.sparse as sp
    A = sp.csr_matrix((v, ci, rp), shape=(n, n))
    lrp, lci, lv = P.lower_triangle(n, rp, ci, v)
    s = Hipmf()
    t0 = time.perf_counter()
    code = s.initialize(n, lrp, lci, general_symmetric=True)
    t_init = time.perf_counter() - t0
    if code != 0:
        s.close()
        return {"skipped": "initialize returned %d (not enough device memory?)" % code}
    d_v = s.dev_alloc(lv.nbytes)
    s.h2d(d_v, lv)
    t0 = time.perf_counter()
    code = s.factorize_device(d_v)
    lib.hipmf_device_synchronize()
    t_fac = time.perf_counter() - t0
    B = np.empty((nrhs, n))
    for j in range(nrhs):
        B[j] = np.random.default_rng([20260927, j]).standard_normal(n)
    d_b, d_x = s.dev_alloc(B.nbytes), s.dev_alloc(B.nbytes)
    s.h2d(d_b, B)
    # (round 6: the block buffers of the blocked solve are allocated and touched ahead of the timed call -- solver_hipmf_prepare_solve_many, what
    #  a rank does while it waits for the root's factor; VERDICT r05: the first call used to pay 0.4 s for them)
    t0 = time.perf_counter()
    s.prepare_solve_many(nrhs)
    t_prep = time.perf_counter() - t0
    t0 = time.perf_counter()
    s.solve_device(d_x, d_b, nrhs, n)
    lib.hipmf_device_synchronize()
    t_solve = time.perf_counter() - t0
    X = np.zeros_like(B)
    s.d2h(X, d_x)
    # (the same shard once more: the first blocked solve of a process allocates its block buffers, and the first large job on a fresh
    #  box has measured 1.3 - 1.5 x the later ones -- profiles/r05_split_dot_products.txt; both numbers are reported)
    t0 = time.perf_counter()
    s.solve_device(d_x, d_b, nrhs, n)
    lib.hipmf_device_synchronize()
    t_solve2 = time.perf_counter() - t0
    st = s.stats()
    worst = 0.0
    for j0 in range(0, nrhs, 8):
        R = A @ X[j0:j0 + 8].T - B[j0:j0 + 8].T
        worst = max(worst, float(np.max(np.abs(R))) / (float(np.max(np.abs(v))) + 1.0))
    res = {"workload": "3D 7-point Poisson %d^3 (n = %d) as its lower triangle (L D L^T), %d right-hand sides = one rank's shard of the 256 "
                       "(default_rng([20260927, column]).standard_normal), resident in HBM" % (edge, n, nrhs),
           "initialize_s": round(t_init, 2), "factorize_s": round(t_fac, 3), "factorize_code": int(code), "solve_s": round(t_solve, 3),
           "solve_repeat_s": round(t_solve2, 3), "prepare_solve_many_s": round(t_prep, 3),
           "ms_per_rhs": round(t_solve * 1e3 / nrhs, 2), "pool_gb": round(st["pool_bytes"] / 1e9, 1),
           "lu_equivalent_tflops": round(st["flops"] / t_fac / 1e12, 1), "max_relative_error_all_columns": worst,
           "fused_fallbacks": int(st.get("fused_fallbacks", 0)),
           # replicate-or-broadcast for the 8-GPU split, from THIS GPU's numbers: moving the persistent factor over one
           # xGMI link (153 GB/s, ring broadcast: per-link bound) against factorising it again on every rank
           # Round 6: solver_hipmf_broadcast_factor moves a part of >= 64 MB as N slices in two point-to-point steps (root -> rank r: slice r;
           # then every rank its own slice to every other rank), every transfer of a step on a link of its own: 2 F / (N x 153 GB/s)
           # instead of the F / 153 GB/s of a ring that leaves the root over one link.  A MODEL from this GPU's numbers: the 8-GPU run is the
           # driver's (SCALE_rNN.json) -- no collective is simulated here.
           "multi_gpu_model": {"persistent_factor_gb": round(s.counter("persistent_bytes") / 1e9, 1),
                               "ring_broadcast_s_at_153_gbs": round(s.counter("persistent_bytes") / 153e9, 3),
                               "sliced_broadcast_s_8_ranks": round(2.0 * s.counter("persistent_bytes") / 8.0 / 153e9, 3),
                               "replicate_s": round(t_fac, 2),
                               "decision": "broadcast" if 2.0 * s.counter("persistent_bytes") / 8.0 / 153e9 < t_fac else "replicate",
                               "solve_s_256_rhs_over_8_gpus_model": round(min(t_solve, t_solve2), 3),
                               "solve_s_256_rhs_on_one_gpu_model": round(8.0 * min(t_solve, t_solve2), 3),
                               "predicted_scaling_8_gpus": round(8.0 * min(t_solve, t_solve2) / (2.0 * s.counter("persistent_bytes") / 8.0 / 153e9 + min(t_solve, t_solve2)), 2),
                               "note": "8 ranks x 32 columns run concurrently: the sharded solve takes what this rank's shard takes; the root factorises once, "
                                       "the factor travels as slices over all seven links (predicted_scaling = 8 shards on one GPU / (sliced broadcast + one shard); "
                                       "with the ring broadcast of rounds 1 - 5 the same formula gives %.1f)"
                                       % (8.0 * min(t_solve, t_solve2) / (s.counter("persistent_bytes") / 153e9 + min(t_solve, t_solve2)))}}
    for ptr in (d_v, d_b, d_x):
        s.dev_free(ptr)
    s.close()
    return res


def config5_section():
    """BASELINE config 5: the Brusselator PDE under Radau5 at the reference's published size (npoint = 513, data/logs/brus_pde_2nd_umfpack_24.txt),
    real and complex handle on two host threads like radau5.rs:270-296; the harness prints its counters as JSON."""
    import subprocess
    exe = os.path.join(ROOT, "russell_amd", "lib", "brusselator_pde")
    r = subprocess.run([exe, "--npoint", "513", "--json", "-g", "hipmf"], capture_output=True, text=True, timeout=600)
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    if r.returncode != 0 or not lines:
        return {"error": "brusselator_pde exited with %d: %s" % (r.returncode, r.stderr[-300:])}
    d = json.loads(lines[-1])
    keep = ("npoint", "ndim", "jac_nnz", "n_function", "n_jacobian", "n_factor", "n_lin_sol", "n_steps", "n_accepted", "n_rejected", "n_iterations_max",
            "h_accepted", "ms_total", "ms_factor_max", "ms_factor_avg", "ms_lin_sol_max", "ms_lin_sol_avg", "fused_fallbacks", "chain_fallbacks", "gate_waits")
    out = {k: d[k] for k in keep if k in d}
    out["workload"] = "russell_amd/lib/brusselator_pde --npoint 513 (second-book problem, tolerance 1e-4, real + complex system on two threads)"
    out["reference_log"] = {"file": "data/logs/brus_pde_2nd_umfpack_24.txt (reference, 24 threads + MKL)", "n_function": 266, "n_jacobian": 23, "n_factor": 44,
                            "n_lin_sol": 75, "n_steps": 44, "n_accepted": 35, "n_rejected": 9, "total": "4m34s"}
    return out


def main():
    if len(sys.argv) == 3 and sys.argv[1] == "--pardiso-probe":
        _pardiso_probe(int(sys.argv[2]))
        return
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--grid", type=int, default=1000, help="nx = ny of the 2D 5-point Poisson grid (1000 = BASELINE config)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-tier", default="auto", choices=["auto", "umfpack", "pardiso", "superlu", "port"])
    ap.add_argument("--nrhs", type=int, default=NRHS_TOTAL, help="right-hand sides of the many-RHS section (0: skip)")
    ap.add_argument("--no-extras", action="store_true", help="headline only (profiling runs)")
    ap.add_argument("--no-configs", action="store_true", help="skip the config3 / config4 / config5 sections")
    ap.add_argument("--grid3d", type=int, default=100, help="edge of the 3D 7-point Poisson problem of the `poisson3d` extra (0: skip)")
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    dist = None
    tdev = None
    if world > 1 or os.environ.get("BENCH_FORCE_DIST") == "1":
        import torch
        import torch.distributed as dist
        torch.cuda.set_device(local_rank)
        tdev = torch.device("cuda", local_rank)
        dist.init_process_group("nccl", device_id=tdev)

    from russell_amd import problems as P
    from russell_amd import _capi
    from russell_amd.backend import Hipmf
    from russell_amd.distributed import max_over_ranks, rhs_block

    lib = _capi.load()
    if lib.hipmf_set_device(local_rank) != 0:
        raise RuntimeError("hipmf_set_device(%d) failed" % local_rank)

    def sync_all():
        lib.hipmf_device_synchronize()
        if dist is not None:
            import torch
            dist.barrier()
            torch.cuda.synchronize()

    def rank_max(value):
        return max_over_ranks(value, dist, device=tdev) if dist is not None else float(value)

    n, rp, ci, v = P.poisson2d(args.grid)
    xs = P.manufactured_solution(n)
    b = P.csr_matvec(n, rp, ci, v, xs) + float(rank)  # every rank owns a different right-hand side

    # ---------------------------------------------------------------- headline: general storage, LU, 1 RHS per GPU
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

    for _ in range(args.warmup):
        step()
    s.reset_timers()
    sync_all()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    sync_all()
    elapsed = rank_max(time.perf_counter() - t0)
    ms_per_step = elapsed * 1e3 / args.steps

    st = s.stats()
    x = np.zeros(n)
    s.d2h(x, d_x)
    rel_err = residual_metric(n, rp, ci, v, x, b)
    perm = s.permutation() if (rank == 0 and world == 1 and not args.no_cpu_baseline) else None

    extras = {}
    # ---------------------------------------------------------------- the same matrix as its lower triangle: L D L^T
    if not args.no_extras:
        lrp, lci, lv = lower_triangle(n, rp, ci, v)
        s2 = Hipmf()
        assert s2.initialize(n, lrp, lci, general_symmetric=True) == 0
        d_lv = s2.dev_alloc(lv.nbytes)
        s2.h2d(d_lv, lv)
        for _ in range(max(args.warmup, 1)):
            assert s2.factorize_device(d_lv) == 0
            s2.solve_device(d_x, d_b)
        s2.reset_timers()
        sync_all()
        t0 = time.perf_counter()
        for _ in range(args.steps):
            assert s2.factorize_device(d_lv) == 0
            s2.solve_device(d_x, d_b)
        sync_all()
        sym_ms = rank_max(time.perf_counter() - t0) * 1e3 / args.steps
        st2 = s2.stats()
        x2 = np.zeros(n)
        s2.d2h(x2, d_x)
        tri2 = (st2["acc_fwd_ms"] + st2["acc_bwd_ms"]) / max(st2["acc_tri_count"], 1.0)
        extras["symmetric"] = {
            "workload": "same matrix handed over as its LOWER triangle with general_symmetric = 1 (Sym::YesLower, interface_cudss.cu:324-333): "
                        "L D L^T on the tiled fronts, values and rhs resident in HBM",
            "value_ms": round(sym_ms, 3), "factor_ms": round(st2["acc_factor_ms"] / max(st2["acc_factor_count"], 1.0), 3),
            "sptrsv_pair_ms": round(tri2, 4), "pool_gb": round(st2["pool_bytes"] / 1e9, 3), "relative_error": residual_metric(n, rp, ci, v, x2, b),
            "max_abs_diff_vs_lu": float(np.max(np.abs(x2 - x)))}
        s2.dev_free(d_lv)
        s2.close()

    # ---------------------------------------------------------------- a 3D problem beside the headline (where the latency floors amortise)
    if not args.no_extras and rank == 0 and world == 1 and args.grid3d > 0:
        try:
            n3, rp3, ci3, v3 = P.poisson3d(args.grid3d)
            b3 = P.csr_matvec(n3, rp3, ci3, v3, P.manufactured_solution(n3))
            lrp3, lci3, lv3 = lower_triangle(n3, rp3, ci3, v3)
            s3 = Hipmf()
            t0 = time.perf_counter()
            assert s3.initialize(n3, lrp3, lci3, general_symmetric=True) == 0
            t_init3 = time.perf_counter() - t0
            d_v3, d_b3, d_x3 = s3.dev_alloc(lv3.nbytes), s3.dev_alloc(b3.nbytes), s3.dev_alloc(b3.nbytes)
            s3.h2d(d_v3, lv3), s3.h2d(d_b3, b3)
            assert s3.factorize_device(d_v3) == 0
            s3.solve_device(d_x3, d_b3)
            s3.reset_timers()
            for _ in range(3):
                assert s3.factorize_device(d_v3) == 0
                s3.solve_device(d_x3, d_b3)
            lib.hipmf_device_synchronize()
            st3 = s3.stats()
            x3 = np.zeros(n3)
            s3.d2h(x3, d_x3)
            tri3 = (st3["acc_fwd_ms"] + st3["acc_bwd_ms"]) / max(st3["acc_tri_count"], 1.0)
            fac3 = st3["acc_factor_ms"] / max(st3["acc_factor_count"], 1.0)
            bytes3 = sptrsv_bytes(st3, n3)
            extras["poisson3d"] = {
                "workload": "3D 7-point Poisson %d^3 (n = %d) as its lower triangle (L D L^T on the tiled fronts), values and rhs resident in HBM"
                            % (args.grid3d, n3),
                "initialize_s": round(t_init3, 2), "factor_ms": round(fac3, 2), "lu_equivalent_tflops": round(st3["flops"] / (fac3 * 1e-3) / 1e12, 1),
                "sptrsv_pair_ms": round(tri3, 3), "sptrsv_gbs": round(bytes3 / (tri3 * 1e-3) / 1e9, 1),
                "sptrsv_frac_of_hbm_peak": round(bytes3 / (tri3 * 1e-3) / 1e9 / HBM_PEAK_GBS, 4), "pool_gb": round(st3["pool_bytes"] / 1e9, 2),
                "relative_error": residual_metric(n3, rp3, ci3, v3, x3, b3)}
            for ptr in (d_v3, d_b3, d_x3):
                s3.dev_free(ptr)
            s3.close()
        except Exception as exc:  # never lose the headline to an extra
            extras["poisson3d"] = {"error": repr(exc)}

    # ---------------------------------------------------------------- host-pointer boundary through the mirror of the Rust layer
    if not args.no_extras and rank == 0:
        try:
            from russell_amd import sparse as RS
            rows = np.repeat(np.arange(n, dtype=np.int32), np.diff(rp))
            coo = RS.CooMatrix(n, n, len(v))
            coo.put_many(rows, ci.astype(np.int32), v)
            hs = RS.LinSolver(RS.Genie.Hipmf)
            t0 = time.perf_counter()
            hs.actual.factorize(coo)  # first call: COO -> CSR, initialize, value map, factorize
            t_first = time.perf_counter() - t0
            tf = ts = 0.0
            reps, first_solves = 5, []
            xh = np.zeros(n)  # (the caller's x, reused from call to call as russell's solvers do)
            for it in range(reps + 2):
                t0 = time.perf_counter()
                hs.actual.factorize(coo)  # repeat call (params None): values through the device-side map + numeric LU
                t1 = time.perf_counter()
                hs.actual.solve(b, x=xh)
                t2 = time.perf_counter()
                if it < 2:  # the first two host solves pay one-time runtime costs (code objects, pinned staging, first DMA): reported apart
                    first_solves.append(round((t2 - t1) * 1e3, 1))
                    continue
                tf += t1 - t0
                ts += t2 - t1
            extras["host_api"] = {
                "workload": "LinSolTrait::factorize(&coo, None) repeat call + solve(&mut x, &rhs) with HOST vectors (H2D of 4 996 000 triplet values, "
                            "device-side COO->CSR value refresh, numeric LU; H2D rhs, solve, D2H x), wall clock",
                "factorize_ms": round(tf / reps * 1e3, 3), "solve_ms": round(ts / reps * 1e3, 3), "total_ms": round((tf + ts) / reps * 1e3, 3),
                "first_call_ms": round(t_first * 1e3, 1), "first_two_solves_ms": first_solves, "timed_repeats": reps,
                "relative_error": residual_metric(n, rp, ci, v, xh, b)}
            del hs
        except Exception as exc:  # never lose the headline to an extra
            extras["host_api"] = {"error": repr(exc)}

    # ---------------------------------------------------------------- one driver-timed number per BASELINE config (rank 0, one GPU)
    if not args.no_extras and not args.no_configs and rank == 0 and world == 1:
        for key, fn in (("config5", lambda: config5_section()), ("config3", lambda: config3_section(Hipmf, P, lib))):
            try:
                extras[key] = fn()
            except Exception as exc:  # never lose the headline to an extra
                extras[key] = {"error": repr(exc)}

    # ---------------------------------------------------------------- many right-hand sides, sharded (north_star)
    # Every rank runs the same sequence of collectives whatever happens locally: local work sits in try blocks that only set a
    # flag, the ranks agree on the flag (MIN over ranks) before the next collective step -- a failure on one rank skips the
    # section everywhere instead of leaving the others waiting in a collective.
    def all_ok(flag):
        if dist is None:
            return bool(flag)
        import torch
        t = torch.tensor([1 if flag else 0], dtype=torch.int32, device=tdev)
        dist.all_reduce(t, op=dist.ReduceOp.MIN)
        return bool(int(t.item()))

    if not args.no_extras and args.nrhs > 0:
        many, err = {}, None
        first, count = rhs_block(args.nrhs, world, rank)
        d_B = d_X = None
        Bh = Xh = None
        try:
            # B = default_rng(20260927).standard_normal((n, nrhs)) -- independent columns (round 4 used scalar multiples
            # of ONE vector, which a column-mixing bug inside a 16-column block that preserves direction would survive: VERDICT r04);
            # column j of the WHOLE block is the same whatever the number of ranks (one generator per column)
            Bh = np.empty((max(count, 1), n))
            for j in range(count):
                Bh[j] = np.random.default_rng([20260927, first + j]).standard_normal(n)
            Xh = np.zeros_like(Bh)
            d_B = s.dev_alloc(Bh.nbytes)
            d_X = s.dev_alloc(Bh.nbytes)
            s.h2d(d_B, Bh)
        except Exception as exc:
            err = "setup: %r" % (exc,)
        ok = all_ok(err is None)

        def column_error():
            """largest relative_error (VerifyLinSys: |A x - b|_inf / (max|a| + 1)) over EVERY column of this rank's block"""
            s.d2h(Xh, d_X)
            if count == 0:
                return 0.0
            import scipy.sparse as sp
            A = sp.csr_matrix((v, ci, rp), shape=(n, n))
            worst_here = 0.0
            for j0 in range(0, count, 32):  # (32 columns at a time: 0.25 GB of residuals)
                R = A @ Xh[j0:min(count, j0 + 32)].T - Bh[j0:min(count, j0 + 32)].T
                worst_here = max(worst_here, float(np.max(np.abs(R))) / (float(np.max(np.abs(v))) + 1.0))
            return worst_here

        # (a) replicated: every rank factorises (no data-path collective at all)
        if ok:
            t_fact_rep = t_solve = 0.0
            worst = -1.0
            sync_all()
            t0 = time.perf_counter()
            try:
                assert s.factorize_device(d_vals) == 0
                lib.hipmf_device_synchronize()
            except Exception as exc:
                err = "replicate factorize: %r" % (exc,)
            t_fact_rep = rank_max(time.perf_counter() - t0)
            sync_all()
            t0 = time.perf_counter()
            try:
                if count > 0 and err is None:
                    s.solve_device(d_X, d_B, nrhs=count)
                lib.hipmf_device_synchronize()
            except Exception as exc:
                err = "replicate solve: %r" % (exc,)
            t_solve = rank_max(time.perf_counter() - t0)
            try:
                if err is None:
                    worst = column_error()
            except Exception as exc:
                err = "replicate check: %r" % (exc,)
            worst = rank_max(worst)
            ok = all_ok(err is None)
            if ok:
                many = {"nrhs_total": args.nrhs, "rhs_per_gpu": count, "solve_ms": round(t_solve * 1e3, 3),
                        "replicate": {"factorize_ms": round(t_fact_rep * 1e3, 3), "total_ms": round((t_fact_rep + t_solve) * 1e3, 3),
                                      "rhs_per_s": round(args.nrhs / (t_fact_rep + t_solve), 1)},
                        "max_relative_error_all_columns": worst, "rhs": "default_rng([20260927, column]).standard_normal(n) per column (SURVEY.md 8d)"}
                try:
                    # replicate or broadcast, from THIS run's numbers: re-factorising on every rank against moving the
                    # persistent factor over one xGMI link (ring broadcast: per-link bound, 153 GB/s) after the root's factorisation
                    pb = s.counter("persistent_bytes")
                    many["multi_gpu_model"] = {"persistent_factor_gb": round(pb / 1e9, 3), "ring_broadcast_ms_at_153_gbs": round(pb / 153e9 * 1e3, 2),
                                               "sliced_broadcast_ms_8_ranks": round(2.0 * pb / 8.0 / 153e9 * 1e3, 2),
                                               "replicate_ms": round(t_fact_rep * 1e3, 2),
                                               "decision": "broadcast" if 2.0 * pb / 8.0 / 153e9 < 0.5 * t_fact_rep else "replicate",
                                               "note": "broadcast pays when moving the factor (as slices over all links: 2 F / (8 x 153 GB/s), round 6) takes less "
                                                       "than half a factorisation (the root still factorises once)"}
                except Exception:
                    pass
        # (b) north_star: ONE rank factorises, the factor goes to the others over RCCL / xGMI
        if ok and dist is not None:  # (also with ONE rank under BENCH_FORCE_DIST=1: the same collectives, nranks is data)
            import torch
            comm = ctypes.c_void_p()
            idt = torch.zeros(129, dtype=torch.uint8, device=tdev)  # 128 bytes of id + a "valid" byte
            try:
                if rank == 0:
                    idbuf = (ctypes.c_uint8 * 128)()
                    if lib.hipmf_comm_unique_id(idbuf) == 0:
                        idt.copy_(torch.tensor(list(bytes(idbuf)) + [1], dtype=torch.uint8))
            except Exception as exc:
                err = "unique id: %r" % (exc,)
            dist.broadcast(idt, src=0)
            idh = bytes(idt.cpu().numpy().tobytes())
            have_comm = False
            try:
                if idh[128] == 1:
                    idb = (ctypes.c_uint8 * 128).from_buffer_copy(idh[:128])
                    have_comm = lib.hipmf_comm_init_rank(ctypes.byref(comm), world, idb, rank) == 0
            except Exception as exc:
                err = "comm init: %r" % (exc,)
            ok_b = all_ok(idh[128] == 1 and have_comm and err is None)
            if ok_b:
                t_fact = t_bc = t_solve2 = 0.0
                nbytes, worst2 = 0, -1.0
                try:
                    s.broadcast_factor(comm, 0, rank)  # warm-up of the communicator (the first collective sets up the rings)
                    if rank != 0:
                        # the other ranks overwrite their replicated factor with that of ANOTHER matrix (values x 2): the columns below
                        # only come out right if the broadcast really delivered rank 0's factor
                        d_v2 = s.dev_alloc(v.nbytes)
                        s.h2d(d_v2, 2.0 * v)
                        assert s.factorize_device(d_v2) == 0
                        s.dev_free(d_v2)
                except Exception as exc:
                    err = "broadcast warm-up: %r" % (exc,)
                if all_ok(err is None):
                    sync_all()
                    t0 = time.perf_counter()
                    try:
                        if rank == 0:
                            assert s.factorize_device(d_vals) == 0
                        lib.hipmf_device_synchronize()
                    except Exception as exc:
                        err = "root factorize: %r" % (exc,)
                    sync_all()
                    t_fact = rank_max(time.perf_counter() - t0)
                    if all_ok(err is None):
                        t0 = time.perf_counter()
                        try:
                            sec, nbytes = s.broadcast_factor(comm, 0, rank)
                        except Exception as exc:  # (an RCCL failure inside the collective may still hang the others: nothing a caller can do)
                            err = "broadcast: %r" % (exc,)
                        sync_all()
                        t_bc = rank_max(time.perf_counter() - t0)
                        t0 = time.perf_counter()
                        try:
                            if count > 0 and err is None:
                                s.solve_device(d_X, d_B, nrhs=count)
                            lib.hipmf_device_synchronize()
                            if err is None:
                                worst2 = column_error()
                        except Exception as exc:
                            err = "sharded solve: %r" % (exc,)
                        t_solve2 = rank_max(time.perf_counter() - t0)
                        worst2 = rank_max(worst2)
                        if all_ok(err is None) and t_bc > 0:
                            many["broadcast"] = {"factorize_ms": round(t_fact * 1e3, 3), "broadcast_ms": round(t_bc * 1e3, 3),
                                                 "broadcast_bytes": int(nbytes), "broadcast_gbs": round(nbytes / t_bc / 1e9, 1),
                                                 "solve_ms": round(t_solve2 * 1e3, 3), "total_ms": round((t_fact + t_bc + t_solve2) * 1e3, 3),
                                                 "rhs_per_s": round(args.nrhs / (t_fact + t_bc + t_solve2), 1),
                                                 "max_relative_error_all_columns": worst2}
                try:
                    lib.hipmf_comm_destroy(comm)
                except Exception:
                    pass
            if "broadcast" not in many:
                many["broadcast"] = {"error": err or "RCCL communicator not available on every rank"}
        if many:
            many["rhs_per_s"] = max(many["replicate"]["rhs_per_s"], many.get("broadcast", {}).get("rhs_per_s", 0.0))
            try:
                # blocks of BLK columns read the factor once per block and triangular pass pair; every column is solved once and
                # refined (st["refinement_steps"] further pass pairs): physical factor bytes moved / solve time against the HBM peak
                stm = s.stats()
                blk = 16 if count > 12 else 8
                passes = 1 + int(stm.get("refinement_steps", 1))
                nblocks = (count + blk - 1) // blk
                fbytes = (stm["nnz_l"] + stm["nnz_u"]) * 8
                moved = nblocks * passes * fbytes
                many["roofline"] = {"bound": "hbm", "block_columns": blk, "blocks": nblocks, "pass_pairs_per_block": passes,
                                    "physical_factor_bytes_per_pass_pair": int(fbytes), "achieved": round(moved / (many["solve_ms"] * 1e-3) / 1e9, 1),
                                    "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(moved / (many["solve_ms"] * 1e-3) / 1e9 / HBM_PEAK_GBS, 4),
                                    "ms_per_rhs": round(many["solve_ms"] / max(count, 1), 4),
                                    "blocks_per_launch": int(s.counter("block_groups")),
                                    "fused_solve_fallbacks": int(stm.get("fused_fallbacks", 0))}
            except Exception as exc:
                many["roofline"] = {"error": repr(exc)}
            extras["many_rhs"] = many
        else:
            extras["many_rhs"] = {"error": err or "skipped: another rank failed"}
        for ptr in (d_B, d_X):
            if ptr:
                s.dev_free(ptr)

    if rank == 0:
        tri = max(st["acc_tri_count"], 1.0)
        tri_ms = (st["acc_fwd_ms"] + st["acc_bwd_ms"]) / tri
        copy_gbs = ctypes.c_double(0.0)
        if lib.hipmf_device_copy_bandwidth(1 << 30, 3, ctypes.byref(copy_gbs)) != 0:
            copy_gbs.value = 0.0
        mfma_tfs = ctypes.c_double(0.0)
        if lib.hipmf_device_mfma_rate(1024, 4000, ctypes.byref(mfma_tfs)) != 0:
            mfma_tfs.value = 0.0
        bytes_alg = sptrsv_bytes(st, n)
        phys_bytes = (st["nnz_l"] + st["nnz_u"]) * 8 + n * 8 * 4
        traffic, traffic_src = measured_traffic(args.grid, st)
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
            "config": {"workload": "values+rhs resident in HBM: numeric LU factorize + solve, 2D 5-point Poisson %dx%d grid (n=%d, nnz=%d) f64, general "
                                   "storage, 1 RHS per GPU (through host pointers: value_host_boundary_ms)" % (args.grid, args.grid, n, int(rp[-1])),
                       "rhs_per_gpu": 1, "refinement_steps": st["refinement_steps"]},
            "sptrsv_gbs": round(achieved, 1),
            "roofline": {"kernel": "multifrontal SpTRSV pass, forward + backward (%s, %d launches over %d tree levels)" %
                                   ("wave-subtrees k_wt_fwd / k_wt_bwd + dependency-driven k_fwd_fused / k_bwd_fused (mid and top levels)"
                                    if st["solve_launches"] <= 6 else "level-set k_fwd/k_bwd[_big]", st["solve_launches"], st["nlevels"]),
                         "bound": "hbm", "achieved": round(achieved, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": round(achieved / HBM_PEAK_GBS, 4), "traffic": traffic, "traffic_source": traffic_src,
                         "traffic_over_algorithmic": round(traffic / bytes_alg, 3) if traffic else None,
                         "traffic_over_physical": round(traffic / phys_bytes, 3) if traffic else None,
                         "algorithmic_bytes": int(bytes_alg), "avg_ms": round(tri_ms, 4),
                         # the factor is stored supernodally: 8 B per stored entry and pass (SURVEY.md 8d charges 12 B per entry)
                         "physical_bytes": int(phys_bytes), "physical_gbs": round(phys_bytes / (tri_ms * 1e-3) / 1e9, 1) if tri_ms > 0 else 0.0,
                         "physical_frac": round(phys_bytes / (tri_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4) if tri_ms > 0 else 0.0,
                         "measured_copy_gbs": round(copy_gbs.value, 1),
                         "frac_of_measured_copy": round(achieved / copy_gbs.value, 4) if copy_gbs.value > 0 else None,
                         "fused_solve_fallbacks": st.get("fused_fallbacks", 0)},
            "roofline_factor": {"kernel": "numeric multifrontal LU (k_small_factor, k_front_lu, k_panel, k_update / k_update32 MFMA f64, k_extend_add_lds)",
                                "bound": "mfma", "achieved": round(st["flops"] / (fact_ms * 1e-3) / 1e12, 3) if fact_ms > 0 else 0.0,
                                "peak": FP64_PEAK_TFLOPS, "unit": "TFLOP/s",
                                "frac": round(st["flops"] / (fact_ms * 1e-3) / 1e12 / FP64_PEAK_TFLOPS, 5) if fact_ms > 0 else 0.0,
                                "flops": st["flops"], "avg_ms": round(fact_ms, 3), "assemble_ms": round(asm_ms, 3),
                                "measured_mfma_tflops": round(mfma_tfs.value, 1),
                                # against what v_mfma_f64_16x16x4_f64 delivers on THIS device (tools/microbench/mfma_peak.hip through the
                                # library's probe), not the data-sheet figure: profiles/r03_mfma_ceiling.txt
                                "frac_of_measured": round(st["flops"] / (fact_ms * 1e-3) / 1e12 / mfma_tfs.value, 5) if fact_ms > 0 and mfma_tfs.value > 0 else None},
            "phases_ms": {"initialize_once": round(t_init * 1e3, 1), "ordering_s": st["ordering_s"], "symbolic_s": st["symbolic_s"],
                          "assemble": round(asm_ms, 3), "factor": round(fact_ms, 3), "sptrsv_pair": round(tri_ms, 4),
                          "solve_total_last": round(st["solve_total_ms"], 3)},
            "factor": {"nnz_l": st["nnz_l"], "nnz_u": st["nnz_u"], "nsuper": st["nsuper"], "nlevels": st["nlevels"],
                       "max_front": st["max_front"], "pool_gb": round(st["pool_bytes"] / 1e9, 3),
                       "factor_launches": st["factor_launches"], "solve_launches": st["solve_launches"], "perturbed": st["n_perturbed"],
                       "tagged_solve": s.counter("tagged_solve"), "wave_fronts": s.counter("wave_fronts")},
            "relative_error": rel_err,
        }
        out.update(extras)
        # the call the reference's LinSolTrait makes (H2D values, refresh, factorise; H2D rhs, solve, D2H x -- interface_cudss.cu:424,524,553)
        if isinstance(extras.get("host_api"), dict) and "total_ms" in extras["host_api"]:
            out["value_host_boundary_ms"] = extras["host_api"]["total_ms"]
        # StatsLinSol's total_ifs (stats_lin_sol.rs:75-103): initialize + factorize + solve of a ONE-SHOT call; the headline is the
        # repeat call (factorize + solve on a handle that is initialised)
        out["total_ifs_ms"] = round(t_init * 1e3 + ms_per_step, 1)
        if perm is not None:
            try:
                cb = cpu_baseline(n, rp, ci, v, b, perm, args.cpu_tier, args.grid)
            except Exception as exc:  # (the baseline is a reported extra: never lose the headline to it)
                cb = {"value": None, "unit": "ms", "cores": 0, "kind": "unavailable", "sample": "cpu_baseline failed: %r" % (exc,)}
            cb["host_cores"] = os.cpu_count() or 0
            #  a threaded tier exists only where a threaded sparse direct solver can be loaded: UMFPACK with a threaded
            # BLAS is the first tier above (cores = its thread count); SuperLU through scipy is sequential, the port is scalar
            cb["threads"] = cb.get("cores", 1)
            if "threaded_tier" not in cb:  # (the probe's outcome: which threaded solver could be loaded, if any)
                cb["threaded_tier"] = "UMFPACK + threaded BLAS" if cb.get("kind") == "reference" else "none could be loaded on this box (scipy's SuperLU is sequential)"
            # SURVEY.md 8(d) tier 2 beside whatever tier gave `value`: the sequential SuperLU figures, labelled
            if "SuperLU" not in str(cb.get("kind", "")) and args.cpu_tier == "auto":
                cb["tier2_superlu"] = superlu_tier(n, rp, ci, v, b)
            out["cpu_baseline"] = cb
            # like for like: a one-shot CPU call (analysis + numeric + solve) against total_ifs_ms, a repeat call (numeric + solve on an
            # analysed handle) against the GPU's repeat call THROUGH THE HOST-POINTER BOUNDARY (H2D values + rhs, D2H x: what the
            # reference's LinSolTrait call pays, interface_cudss.cu:424,524,553) -- not against the HBM-resident `value`; the sequential
            # tiers redo everything per call and only have the first ratio
            one_shot = cb.get("one_shot_ms", cb["value"])
            out["speedup_one_shot"] = round(one_shot / out["total_ifs_ms"], 1) if one_shot else None
            gpu_repeat = out.get("value_host_boundary_ms") or out["value"]
            if "one_shot_ms" in cb:
                out["speedup_repeat_call"] = round(cb["value"] / gpu_repeat, 1)
                out["speedup_repeat_call_hbm_resident"] = round(cb["value"] / out["value"], 1)
            t2 = cb.get("tier2_superlu")
            if isinstance(t2, dict) and "total_ms" in t2:
                out["speedup_one_shot_vs_superlu_1core"] = round(t2["total_ms"] / out["total_ifs_ms"], 1)
            out["speedup_note"] = ("speedup_one_shot = CPU (analysis + numeric + solve) / total_ifs_ms; speedup_repeat_call = CPU (numeric + solve on "
                                   "an analysed handle) / value_host_boundary_ms (the call the reference's boundary makes: host values, host rhs, host x; "
                                   "`value` itself keeps them resident in HBM: speedup_repeat_call_hbm_resident) -- only where the CPU tier times its "
                                   "phases apart; the CPU tier is a third-party stand-in unless cpu_baseline.kind says 'reference'")
        line = json.dumps(out)
    else:
        line = None
    s.dev_free(d_vals), s.dev_free(d_b), s.dev_free(d_x)
    s.close()
    if line is not None and not args.no_extras and not args.no_configs and world == 1 and args.grid == 1000:
        # BASELINE config 4's matrix needs the device to itself (193 GB of factor): after every other handle is closed
        try:
            c4 = config4_section(Hipmf, P, lib)
        except Exception as exc:
            c4 = {"error": repr(exc)}
        out["config4"] = c4
        line = json.dumps(out)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()
    if line is not None:
        # the JSON line goes out LAST and unbuffered: RCCL prints its banner through C stdio, which would otherwise
        # land after Python's buffered print
        sys.stdout.flush()
        try:
            ctypes.CDLL(None).fflush(None)
        except OSError:
            pass
        os.write(1, (line + "\n").encode())


if __name__ == "__main__":
    main()
