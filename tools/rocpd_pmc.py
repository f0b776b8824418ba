#!/usr/bin/env python3
"""Per-kernel totals of the hardware counters in a rocprofv3 (rocpd sqlite) database collected with --pmc.

usage: rocpd_pmc.py run_results.db [more.db ...]    -> one table per database: kernel, calls, counter, sum, avg/call
"""
import re
import sqlite3
import sys
from collections import defaultdict


def short(name):
    m = re.search(r"k_[a-z_0-9]+(<[^>]*>)?|__amd_rocclr_\w+", name)
    return m.group(0) if m else name[:40]


def main():
    for path in sys.argv[1:]:
        db = sqlite3.connect(path)
        rows = db.execute("select kernel_name, counter_name, value, duration from counters_collection").fetchall()
        agg = defaultdict(lambda: [0, 0.0, 0.0])
        for name, cname, val, dur in rows:
            a = agg[(short(name), cname)]
            a[0] += 1
            a[1] += float(val)
            a[2] += float(dur)
        print("# %s" % path)
        print("%-28s %-12s %7s %16s %14s %10s" % ("kernel", "counter", "calls", "sum", "avg/call", "avg_us"))
        for (k, c), (n, tot, dur) in sorted(agg.items(), key=lambda kv: -kv[1][1]):
            print("%-28s %-12s %7d %16.1f %14.2f %10.2f" % (k, c, n, tot, tot / n, dur / n / 1e3))


if __name__ == "__main__":
    main()
