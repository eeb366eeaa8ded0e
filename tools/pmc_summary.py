#!/usr/bin/env python3
"""Summarise a rocprofv3 rocpd database: per-kernel average of every collected counter + kernel durations."""
import collections
import sqlite3
import sys

for path in sys.argv[1:]:
    con = sqlite3.connect(path)
    cur = con.cursor()
    print("#", path)
    try:
        rows = list(cur.execute("select kernel_name, counter_name, value from counters_collection"))
        agg = collections.defaultdict(list)
        for k, c, v in rows:
            agg[(k[:70], c)].append(v)
        for (k, c), v in sorted(agg.items()):
            if "dexr" in k:
                print(f"{k:70s} {c:28s} n={len(v):3d} avg={sum(v)/len(v):.4g}")
    except Exception as e:  # no counters in this db
        print("no counters:", e)
    try:
        for r in cur.execute("select name,total_calls,average from top_kernels"):
            if "dexr" in r[0]:
                print(f"{r[0][:70]:70s} calls={r[1]} avg_us={r[2]:.2f}")
    except Exception:
        pass
