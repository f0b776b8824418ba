#!/usr/bin/env python3
"""Run the reference's benchmark harness (bin/solve_matrix_market.rs, mirrored by russell_amd/lib/solve_matrix_market) over every
MatrixMarket file found under data/ (or the directory given): the matrices of tools/sweep.txt that are present are solved in the
order of the list, then any other *.mtx; one JSON object per matrix (the harness's own StatsLinSol record) goes to
gpurun_out/sweep/<name>.json and a one-line summary to stdout.  Missing matrices are listed, not fetched (no network here).

usage: python tools/sweep_matrix_market.py [DIR] [-- extra harness options, e.g. -r 3]"""
import glob
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HARNESS = os.path.join(ROOT, "russell_amd", "lib", "solve_matrix_market")


def main():
    args = sys.argv[1:]
    extra = []
    if "--" in args:
        k = args.index("--")
        args, extra = args[:k], args[k + 1:]
    data = args[0] if args else os.path.join(ROOT, "data")
    names = []
    for line in open(os.path.join(ROOT, "tools", "sweep.txt")):
        t = line.split("#")[0].split()
        if len(t) >= 2:
            names.append(t[1])
    present = {os.path.splitext(os.path.basename(f))[0]: f for f in sorted(glob.glob(os.path.join(data, "**", "*.mtx"), recursive=True))}
    order = [n for n in names if n in present] + [n for n in sorted(present) if n not in names]
    absent = [n for n in names if n not in present]
    out_dir = os.path.join(ROOT, "gpurun_out", "sweep")
    os.makedirs(out_dir, exist_ok=True)
    failures = 0
    for n in order:
        r = subprocess.run([HARNESS, "-g", "hipmf"] + extra + [present[n]], capture_output=True, text=True)
        rec = None
        try:
            rec = json.loads(r.stdout[r.stdout.index("{"):])
        except Exception:
            pass
        if r.returncode != 0 or rec is None:
            failures += 1
            print("%-24s FAILED (rc %d): %s" % (n, r.returncode, (r.stderr or r.stdout).strip().splitlines()[-1:] or ""))
            continue
        json.dump(rec, open(os.path.join(out_dir, n + ".json"), "w"), indent=1)
        m, t, v = rec.get("matrix", {}), rec.get("time_human", {}), rec.get("verify", {})
        print("%-24s n %9s nnz %11s  total %-12s factorize %-12s solve %-12s rel. error %s" %
              (n, m.get("nrow"), m.get("nnz"), t.get("total_ifs"), t.get("factorize"), t.get("solve"), v.get("relative_error")))
    print("%d solved, %d failed, %d of the list absent%s" % (len(order) - failures, failures, len(absent), (": " + " ".join(absent)) if absent else ""))
    return 1 if failures else 0


if __name__ == "__main__":
    sys.exit(main())
