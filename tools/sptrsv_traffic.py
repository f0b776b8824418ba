#!/usr/bin/env python3
"""profiles/rNN_sptrsv_traffic.json from the FETCH_SIZE / WRITE_SIZE summary tools/rocpd_pmc.py printed (profiles/rNN_pmc_hbm.txt):
HBM bytes of ONE SpTRSV pass pair = sum over the solve kernels of (FETCH_SIZE x 2 + WRITE_SIZE) KB per call x calls per pass pair.
The x 2 on FETCH_SIZE is the gfx950 correction of /opt/skills/guides/MI355X_MICROARCH.md, checked for this code's access shapes by
tools/microbench/fetch_calib.hip (profiles/r04_fetch_calib.txt: the counter moves 64 bytes per 128-byte line for every shape).
The file is STAMPED with the factor the counters were taken on (nnz(L), nnz(U), supernodes, solve launches: the `factor` object and
roofline of the bench line of the SAME build): bench.py refuses a traffic figure whose stamp does not match the build it is timing.
usage: sptrsv_traffic.py <pmc_hbm.txt> <pass pairs per timed step> <algorithmic bytes> <physical bytes> <source label> [<bench.json of the same build>]"""
import json
import re
import sys

path, pairs_per_step, algorithmic, physical, label = sys.argv[1], int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4]), sys.argv[5]
SOLVE = ("k_wt_fwd", "k_wt_bwd", "k_fwd_fused", "k_bwd_fused")
per = {}
for line in open(path):
    m = re.match(r"^(\S.*?)\s+(FETCH_SIZE|WRITE_SIZE)\s+(\d+)\s+([\d.]+)\s+([\d.]+)\s+([\d.]+)\s*$", line)
    if not m or not m.group(1).startswith(SOLVE):
        continue
    per.setdefault(m.group(1).strip(), {})[m.group(2)] = (int(m.group(3)), float(m.group(4)))
calls = {k: v["FETCH_SIZE"][0] for k, v in per.items()}
steps = min(calls.values())  # a kernel that runs once per pass pair (k_wt_fwd) gives the number of pass pairs in the run
fetch = sum(v["FETCH_SIZE"][1] for v in per.values()) / steps
write = sum(v["WRITE_SIZE"][1] for v in per.values()) / steps
traffic = int((2.0 * fetch + write) * 1024)
out = {
    "kernel": "SpTRSV pass pair (k_wt_fwd, k_fwd_fused mid + top, k_bwd_fused top + mid, k_wt_bwd)",
    "source": label,
    "pass_pairs_in_the_run": steps,
    "fetch_size_kb_per_pass": round(fetch, 2),
    "write_size_kb_per_pass": round(write, 2),
    "gfx950_fetch_correction": 2.0,
    "fetch_correction_calibration": "profiles/r04_fetch_calib.txt: counter / true bytes = 0.500 for 16-byte and 8-byte-per-lane flat pieces and for 128-byte segments, 1.000 for aligned 64-byte segments (a 128-byte line moves whole), 1.499 for 64-byte segments 32 bytes off a line boundary",
    "traffic_bytes_per_pass": traffic,
    "algorithmic_bytes": algorithmic,
    "physical_bytes": physical,
    "ratio_to_algorithmic_bytes": round(traffic / algorithmic, 3),
    "ratio_to_physical_bytes": round(traffic / physical, 3),
    "per_kernel_kb_per_pass": {k: {c: round(v[c][1] / steps, 2) for c in v} for k, v in per.items()},
}
if len(sys.argv) > 6:
    d = json.loads(open(sys.argv[6]).read().strip().split("\n")[-1])
    out["stamp"] = {"nnz_l": d["factor"]["nnz_l"], "nnz_u": d["factor"]["nnz_u"], "nsuper": d["factor"]["nsuper"], "solve_launches": d["factor"]["solve_launches"]}
print(json.dumps(out, indent=1))
