#!/usr/bin/env python3
"""Summarise a HIPMF_SF_TRACE dump: per direction and tree level, when the slab tasks of the big fronts started, finished waiting,
gathered their inputs, finished the dot products, stored and published (microseconds from the first stamp of the pass)."""
import sys
from collections import defaultdict

rows = [l.split() for l in open(sys.argv[1])]
for d in "FB":
    rs = [(int(r[1]), int(r[3]), int(r[4]), int(r[5]), [int(v) / 100.0 for v in r[6:12]]) for r in rows if r[0] == d and int(r[3]) > 1 and int(r[9]) > 0]
    if not rs:
        continue
    t0 = min(r[4][0] for r in rs)
    lv = defaultdict(list)
    for lev, kind, p, f, ts in rs:
        lv[lev].append((p, f, [t - t0 for t in ts]))
    print("direction %s: level tasks pmax first_start last_wait_end last_store last_publish | median: start->wait_end, gather, dots, reduce+store, publish (us)" % d)
    for lev in sorted(lv, reverse=(d == "B")):
        v = lv[lev]
        med = lambda xs: sorted(xs)[len(xs) // 2]
        # stamps: 0 start, 1 wait end, 2 stored, 3 published, 4 gathered, 5 dots done
        print("   %2d %6d %5d %10.1f %12.1f %10.1f %10.1f | %6.1f %6.1f %6.1f %6.1f %6.1f" % (
            lev, len(v), max(x[0] for x in v), min(x[2][0] for x in v), max(x[2][1] for x in v), max(x[2][2] for x in v),
            max(x[2][3] for x in v), med([x[2][1] - x[2][0] for x in v]), med([x[2][4] - x[2][1] for x in v]),
            med([x[2][5] - x[2][4] for x in v]), med([x[2][2] - x[2][5] for x in v]), med([x[2][3] - x[2][2] for x in v])))
