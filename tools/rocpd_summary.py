#!/usr/bin/env python3
"""Summarise a rocprofv3 (rocpd sqlite) kernel trace: per-kernel count / total / avg / min / max.

usage: python tools/rocpd_summary.py gpurun_out/prof/run_results.db [--by-dispatch N] > profiles/xxx.txt
"""
import re
import sqlite3
import sys


def main():
    db = sys.argv[1]
    c = sqlite3.connect(db)
    tables = [r[0] for r in c.execute("select name from sqlite_master where type='table'")]
    kd = [t for t in tables if t.startswith("rocpd_kernel_dispatch")][0]
    ks = [t for t in tables if t.startswith("rocpd_info_kernel_symbol")][0]
    cols = [r[1] for r in c.execute("pragma table_info(%s)" % kd)]
    scol = [r[1] for r in c.execute("pragma table_info(%s)" % ks)]
    name_col = "kernel_name" if "kernel_name" in scol else "display_name"
    rows = c.execute("select s.%s, d.start, d.end, d.grid_size_x, d.workgroup_size_x from %s d join %s s on d.kernel_id = s.id order by d.start"
                     % (name_col, kd, ks)).fetchall()
    agg = {}
    for name, st, en, gx, wx in rows:
        short = re.sub(r"\(.*", "", name)
        a = agg.setdefault(short, [0, 0, 1 << 62, 0])
        d = en - st
        a[0] += 1
        a[1] += d
        a[2] = min(a[2], d)
        a[3] = max(a[3], d)
    total = sum(a[1] for a in agg.values())
    print("%-60s %8s %12s %10s %10s %10s %6s" % ("kernel", "calls", "total_us", "avg_us", "min_us", "max_us", "pct"))
    for k, a in sorted(agg.items(), key=lambda kv: -kv[1][1]):
        print("%-60s %8d %12.1f %10.2f %10.2f %10.2f %6.2f" % (k[:60], a[0], a[1] / 1e3, a[1] / a[0] / 1e3, a[2] / 1e3, a[3] / 1e3, 100.0 * a[1] / total))
    print("total kernel time: %.3f ms over %d dispatches" % (total / 1e6, len(rows)))
    if len(sys.argv) > 3 and sys.argv[2] == "--by-dispatch":
        n = int(sys.argv[3])
        print("\nlongest %d dispatches:" % n)
        for name, st, en, gx, wx in sorted(rows, key=lambda r: r[1] - r[2])[:n]:
            print("%-40s %10.2f us grid=%d wg=%d" % (re.sub(r"\(.*", "", name)[:40], (en - st) / 1e3, gx, wx))


if __name__ == "__main__":
    main()
