#!/usr/bin/env python3
"""Print the dispatch sequence (kernel, grid, duration, gap to the previous end) of the LAST numeric factorisation in a
rocprofv3 rocpd kernel trace (the dispatches from the last k_absmax to the first solve kernel after it).

usage: python tools/factor_sequence.py run_results.db > sequence.txt
"""
import re
import sqlite3
import sys


def main():
    c = sqlite3.connect(sys.argv[1])
    tables = [r[0] for r in c.execute("select name from sqlite_master where type='table'")]
    kd = [t for t in tables if t.startswith("rocpd_kernel_dispatch")][0]
    ks = [t for t in tables if t.startswith("rocpd_info_kernel_symbol")][0]
    scol = [r[1] for r in c.execute("pragma table_info(%s)" % ks)]
    name_col = "kernel_name" if "kernel_name" in scol else "display_name"
    rows = c.execute("select s.%s, d.start, d.end, d.grid_size_x, d.workgroup_size_x from %s d join %s s on d.kernel_id = s.id order by d.start"
                     % (name_col, kd, ks)).fetchall()
    names = [re.sub(r"\(.*", "", r[0]).replace("hipmf::", "").replace("void ", "") for r in rows]
    last = max(i for i, n in enumerate(names) if "k_absmax" in n)
    end = next((i for i in range(last, len(rows)) if "k_perm_in" in names[i]), len(rows))
    prev_end = rows[last][1]
    t0 = rows[last][1]
    tot = {}
    for i in range(last, end):
        _, st, en, gx, wx = rows[i]
        print("%-28s wgs=%7d  %9.2f us  gap %7.2f us  t=%9.1f" % (names[i][:28], gx // max(wx, 1), (en - st) / 1e3, (st - prev_end) / 1e3, (st - t0) / 1e3))
        tot[names[i]] = tot.get(names[i], 0) + (en - st)
        prev_end = max(prev_end, en)
    print("span %.1f us" % ((prev_end - t0) / 1e3))
    for k, v in sorted(tot.items(), key=lambda kv: -kv[1]):
        print("  %-28s %10.1f us" % (k, v / 1e3))


main()
