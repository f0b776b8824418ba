#!/usr/bin/env python3
"""Print the per-launch kernel times of the last triangular-solve pass in a rocprofv3 rocpd database."""
import re
import sqlite3
import sys

db = sqlite3.connect(sys.argv[1])
cur = db.cursor()
tabs = [r[0] for r in cur.execute("select name from sqlite_master where type='table'")]
kd = [t for t in tabs if "kernel_dispatch" in t][0]
ks = [t for t in tabs if "kernel_symbol" in t][0]
rows = cur.execute(f"select s.kernel_name, d.start, d.end, d.grid_size_x, d.workgroup_size_x from {kd} d join {ks} s on d.kernel_id=s.id order by d.start").fetchall()
names = [(r[0], (r[2] - r[1]) / 1e3, r[3] // max(r[4], 1), r[4]) for r in rows]
idx = [i for i, n in enumerate(names) if "perm_in" in n[0]][-1]
tot = 0.0
for n in names[idx:idx + 80]:
    m = re.search(r"k_[a-z_]+", n[0])
    print("%-12s wg=%-6d x%-5d %7.1f us" % (m.group(0) if m else n[0][:12], n[2], n[3], n[1]))
    tot += n[1]
    if "perm_out" in n[0]:
        break
print("total %.1f us" % tot)
