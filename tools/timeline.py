#!/usr/bin/env python3
"""Per-dispatch timeline of the last N kernels in a rocprofv3 --kernel-trace rocpd database (start relative to the first
listed dispatch, duration, stream/queue): shows whether launches on forked streams really overlap.

    python tools/timeline.py <dir with *.db> [N]
"""
import glob
import os
import sqlite3
import sys

root, n = sys.argv[1], int(sys.argv[2]) if len(sys.argv) > 2 else 40
for p in sorted(glob.glob(os.path.join(root, "**", "*.db"), recursive=True)):
    con = sqlite3.connect(p)
    cols = [r[1] for r in con.execute("pragma table_info(kernels)")]
    extra = [c for c in ("queue_id", "stream_id", "stream") if c in cols]
    rows = list(con.execute(f"select name, start, end, grid_x, workgroup_x {''.join(', ' + c for c in extra)} from kernels order by start"))
    rows = rows[-n:]
    t0 = rows[0][1]
    print(f"# {p}: last {len(rows)} dispatches; start_us dur_us end_us grid wg {' '.join(extra)} name")
    for r in rows:
        print(f"{(r[1] - t0) * 1e-3:10.1f} {(r[2] - r[1]) * 1e-3:9.1f} {(r[2] - t0) * 1e-3:10.1f} {r[3]:8d} {r[4]:4d} "
              + " ".join(str(v) for v in r[5:]) + " " + r[0][:70])
