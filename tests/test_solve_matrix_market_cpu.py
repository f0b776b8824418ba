"""The benchmark harness of the reference (bin/solve_matrix_market.rs:10-305) and the StatsLinSol JSON it prints
(stats_lin_sol.rs:14-113,212-340), run on the CPU against the emulated backend (host logic only; the same binary is
run against the real HIP library by tests/test_reference_api_gpu.py)."""
import json
import os
import subprocess

import numpy as np
import pytest

from russell_amd.sparse import MMsym, Sym, format_nanoseconds, is_memory_error, read_matrix_market_any

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
MTX = os.path.join(ROOT, "tests", "golden", "mtx")
HARNESS = os.path.join(ROOT, "russell_amd", "lib", "solve_matrix_market")

# every section / field of stats_lin_sol.rs:14-113 that has a meaning for this backend
SCHEMA = {
    "main": {"platform", "blas_lib", "solver", "local_sparse", "out_of_memory"},
    "matrix": {"name", "nrow", "ncol", "nnz", "nnz_actual", "complex", "symmetric"},
    "requests": {"ordering", "scaling", "matching", "pivoting", "mumps_num_threads", "positive_definite", "hybrid_memory_factor"},
    "output": {"effective_ordering", "effective_scaling", "effective_matching", "effective_pivoting", "effective_mumps_num_threads",
               "openmp_num_threads", "umfpack_strategy", "umfpack_rcond_estimate", "rcond_estimate", "perturbed_pivots"},
    "determinant": {"mantissa_real", "mantissa_imag", "base", "exponent"},
    "verify": {"max_abs_a", "max_abs_ax", "max_abs_diff", "relative_error"},
    "time_human": {"read_matrix", "initialize_array", "initialize", "factorize_array", "factorize", "solve_array", "solve", "total_ifs_array",
                   "total_ifs", "verify"},
    "time_nanoseconds": {"read_matrix", "initialize_array", "initialize", "factorize_array", "factorize", "solve_array", "solve",
                         "total_ifs_array", "total_ifs", "verify"},
    "mumps_stats": {"inf_norm_a", "inf_norm_x", "scaled_residual", "backward_error_omega1", "backward_error_omega2", "normalized_delta_x",
                    "condition_number1", "condition_number2"},
}


def run(emu_lib, *args):
    env = dict(os.environ, RUSSELL_HIPMF_LIB=emu_lib)
    p = subprocess.run([HARNESS] + list(args), env=env, capture_output=True, text=True, timeout=300)
    return p.returncode, p.stdout, p.stderr


def test_format_nanoseconds_known_answers():
    # russell_lab base/formatters.rs:238-334
    cases = [(0, "0ns"), (250, "250ns"), (2_500, "2.5µs"), (25_000, "25µs"), (250_000, "250µs"), (2_500_000, "2.5ms"), (25_000_000, "25ms"),
             (250_000_000, "250ms"), (2_500_000_000, "2.5s"), (25_000_000_000, "25s"), (250_000_000_000, "4m10s"),
             (2_500_000_000_000, "41m40s"), (25_000_000_000_000, "6h56m40s"), (250_000_000_000_000, "69h26m40s"), (60_000_000_000, "1m"),
             (120_000_000_000, "2m"), (3_600_000_000_000, "1h"), (3_723_000_000_000, "1h2m3s"), (3_600_000_000_001, "1h1ns"),
             (3_600_000_001_000, "1h1µs"), (3_600_000_100_001, "1h100.001µs"), (3_600_001_000_000, "1h1ms"), (3_601_000_000_000, "1h1s"),
             (3_601_100_000_000, "1h1.1s"),
             # values of data/logs/ASIC_680k_CUDSS.json (time_nanoseconds -> time_human)
             (335862105, "335.862105ms"), (2854383978, "2.854383978s"), (22844039, "22.844039ms")]
    for ns, want in cases:
        assert format_nanoseconds(ns) == want, (ns, format_nanoseconds(ns), want)


def test_is_memory_error():
    # stats_lin_sol.rs:334-340 and the strings this backend produces for a failed device allocation
    for m in ("Error(-1): Not enough memory", "cuDSS: ALLOC_FAILED", "MALLOC failed", "cudaMalloc failed", "device memory is too small"):
        assert is_memory_error(m)
    assert not is_memory_error("Error(1): Matrix is singular")
    from russell_amd.sparse import handle_hipmf_error_code

    assert is_memory_error(handle_hipmf_error_code(100))  # ERROR_HIP_MALLOC of include/russell_hipmf.h: the harness's OOM path


def test_read_matrix_market_complex():
    # read_matrix_market.rs: complex files give the complex matrix only (ok_simple_complex_general.mtx, ok_complex_symmetric_small.mtx)
    coo, ccoo = read_matrix_market_any(os.path.join(MTX, "ok_simple_complex_general.mtx"))
    assert coo is None and ccoo.get_info() == (3, 3, 5, Sym.No)
    ai, aj, ax = ccoo.triplets()
    assert ai.tolist() == [0, 0, 1, 1, 2] and aj.tolist() == [0, 1, 0, 1, 2]
    assert ax.tolist() == [1 - 5j, 2 + 4j, 3 - 3j, 4 + 2j, 5 - 1j]
    _, low = read_matrix_market_any(os.path.join(MTX, "ok_complex_symmetric_small.mtx"), MMsym.LeaveAsLower)
    assert low.get_info() == (5, 5, 7, Sym.YesLower)
    _, full = read_matrix_market_any(os.path.join(MTX, "ok_complex_symmetric_small.mtx"), MMsym.MakeItFull)
    assert full.get_info() == (5, 5, 11, Sym.YesFull)
    fi, fj, fx = full.triplets()
    assert (fi[1], fj[1], fx[1]) == (1, 0, 3 + 2j) and (fi[2], fj[2], fx[2]) == (0, 1, 3 + 2j)  # mirrored without conjugation
    real, none = read_matrix_market_any(os.path.join(MTX, "ok_general.mtx"))
    assert none is None and real.get_info()[:2] == (5, 5)


def test_harness_bfwb62_json_and_golden_solution(emu_lib):
    rc, out, err = run(emu_lib, "-d", "-r", "2", os.path.join(MTX, "bfwb62.mtx"))
    assert rc == 0, err
    assert "BFWB62 FAILED" not in out  # solve_matrix_market.rs:217-229: |x - x_correct| <= 1e-10 with rhs = ones
    d = json.loads(out)
    for section, fields in SCHEMA.items():
        assert fields <= set(d[section]), (section, fields - set(d[section]))
    assert d["main"]["solver"] == "HIPMF" and d["main"]["out_of_memory"] is False
    # bfwb62: 202 stored lower entries, 342 pattern entries; the backend keeps the lower triangle like cuDSS / MUMPS
    assert d["matrix"] == {"name": "bfwb62", "nrow": 62, "ncol": 62, "nnz": 202, "nnz_actual": 342, "complex": False, "symmetric": "YesLower"}
    assert d["requests"]["ordering"] == "Auto" and d["requests"]["scaling"] == "Auto"
    assert d["verify"]["relative_error"] < 1e-12
    assert d["determinant"]["base"] == 10.0 and 1.0 <= abs(d["determinant"]["mantissa_real"]) < 10.0
    t = d["time_nanoseconds"]
    assert len(t["initialize_array"]) == len(t["factorize_array"]) == len(t["solve_array"]) == len(t["total_ifs_array"]) == 2
    for i in range(2):
        assert t["total_ifs_array"][i] == t["initialize_array"][i] + t["factorize_array"][i] + t["solve_array"][i]
    assert abs(t["total_ifs"] - sum(t["total_ifs_array"]) / 2) <= 1
    assert d["time_human"]["total_ifs"] == format_nanoseconds(t["total_ifs"])
    # the pretty form is what serde_json::to_string_pretty prints: two-space indentation, one field per line
    assert out.startswith('{\n  "main": {\n    "platform": ')


def test_harness_request_strings_and_determinant_of_the_umfpack_demo(emu_lib):
    rc, out, err = run(emu_lib, "-o", "metis", "-s", "max", "-d", os.path.join(MTX, "umfpack_di_demo.mtx"))
    assert rc == 0, err
    d = json.loads(out)
    assert d["requests"]["ordering"] == "Metis" and d["requests"]["scaling"] == "Max" and d["output"]["effective_scaling"] == "Max"
    assert d["matrix"]["symmetric"] == "No" and d["matrix"]["nnz"] == d["matrix"]["nnz_actual"] == 12
    det = d["determinant"]["mantissa_real"] * 10.0 ** d["determinant"]["exponent"]
    assert abs(det - 114.0) < 1e-10  # solver_umfpack.rs:585-606
    assert d["verify"]["max_abs_diff"] < 1e-13


def test_harness_complex_files(emu_lib):
    for name in ("ok_simple_complex_general.mtx", "ok_complex_symmetric_small.mtx", "ok_complex_general.mtx"):
        rc, out, err = run(emu_lib, os.path.join(MTX, name))
        assert rc == 0, (name, err)
        d = json.loads(out)
        assert d["matrix"]["complex"] is True and d["verify"]["relative_error"] < 1e-13, (name, d["verify"])
    assert json.loads(run(emu_lib, os.path.join(MTX, "ok_complex_symmetric_small.mtx"))[1])["matrix"]["nnz_actual"] == 11


def test_harness_errors(emu_lib, tmp_path):
    rc, out, err = run(emu_lib, os.path.join(MTX, "bad_missing_data.mtx"))
    assert rc == 1 and "not all values have been found" in err
    rect = tmp_path / "rect.mtx"
    rect.write_text("%%MatrixMarket matrix coordinate real general\n2 3 3\n1 1 1.0\n2 2 1.0\n2 3 1.0\n")
    rc, out, err = run(emu_lib, str(rect))
    assert rc == 1 and "square" in err
    rc, out, err = run(emu_lib, "-g", "umfpack", os.path.join(MTX, "ok_general.mtx"))
    assert rc == 1 and "UMFPACK solver is not available" in err
    rc, out, err = run(emu_lib, "--hide-json", os.path.join(MTX, "ok_general.mtx"))
    assert rc == 0 and out == ""
    rc, out, err = run(emu_lib, "-h", "1.5", os.path.join(MTX, "ok_general.mtx"))
    assert rc == 1 and "hybrid memory factor must be in [0.01, 0.99]" in err


def test_out_of_memory_is_refused_before_any_allocation(emu_lib, tmp_path):
    # a matrix without small separators (random sparse graph): the analysis sees from the column counts that the fronts cannot fit
    # and returns the out-of-memory status instead of asking the host for the row structures; the harness then prints the JSON
    # with main.out_of_memory = true and exits 0 (solve_matrix_market.rs:181-190, stats_lin_sol.rs:334-340)
    import scipy.sparse as sp

    from russell_amd.backend import Hipmf

    n = 30000
    rng = np.random.default_rng(6)
    R = sp.coo_matrix((rng.random(4 * n), (rng.integers(0, n, 4 * n), rng.integers(0, n, 4 * n))), shape=(n, n)).tocsr()
    R = (R + R.T + sp.diags(np.asarray(abs(R + R.T).sum(axis=1)).ravel() + 1.0)).tocsr()
    R.sort_indices()
    s = Hipmf(emu_lib)
    code = s.initialize(n, R.indptr.astype(np.int32), R.indices.astype(np.int32))
    assert code == 100  # ERROR_HIP_MALLOC
    assert "Not enough memory" in str(s._err(code, "initialize")) and is_memory_error(str(s._err(code, "initialize")))
    s.close()
    coo = R.tocoo()
    path = tmp_path / "random.mtx"
    with open(path, "w") as f:
        f.write("%%MatrixMarket matrix coordinate real general\n")
        f.write("%d %d %d\n" % (n, n, coo.nnz))
        np.savetxt(f, np.column_stack([coo.row + 1, coo.col + 1, coo.data]), fmt="%d %d %.17g")
    rc, out, err = run(emu_lib, str(path))
    assert rc == 0, err
    d = json.loads(out)
    assert d["main"]["out_of_memory"] is True and d["matrix"]["nrow"] == n


def test_sweep_runner_consumes_whatever_mtx_is_present(emu_lib, tmp_path):
    # tools/sweep_matrix_market.py: the reference's benchmark sweep (tools/sweep.txt = the matrices its download script names) over
    # any *.mtx found in a directory; there is no network, so the listed matrices are reported absent and whatever IS there is solved
    import shutil
    import sys

    src = os.path.join(ROOT, "tests", "golden", "mtx")
    for name in ("ok_general.mtx", "ok_symmetric.mtx", "ok_complex_general.mtx"):
        shutil.copy(os.path.join(src, name), tmp_path / name)
    env = dict(os.environ, RUSSELL_HIPMF_LIB=emu_lib)
    p = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "sweep_matrix_market.py"), str(tmp_path)], env=env, capture_output=True, text=True,
                       timeout=300)
    assert p.returncode == 0, p.stdout + p.stderr
    last = p.stdout.strip().splitlines()[-1]
    assert last.startswith("3 solved, 0 failed, 43 of the list absent: bbmat af_shell10"), last
    for name in ("ok_general", "ok_symmetric", "ok_complex_general"):
        assert os.path.exists(os.path.join(ROOT, "gpurun_out", "sweep", name + ".json"))
