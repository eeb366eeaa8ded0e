#!/usr/bin/env python3
"""Summarise the rocpd databases written by tools/profile_round.sh into one text file + a small JSON.

    python tools/prof_summary.py <workload> <gpurun_out dir>
"""
import collections
import glob
import json
import os
import sqlite3
import sys

w, out = sys.argv[1], sys.argv[2]


def dbs(tag):
    return sorted(glob.glob(os.path.join(out, f"prof_{w}_{tag}", "**", "*.db"), recursive=True))


def q(path, sql):
    con = sqlite3.connect(path)
    try:
        return list(con.execute(sql))
    finally:
        con.close()


summary = {"workload": w}
try:  # key of the sources this profile was taken on (bench.py only attaches the counters while it still matches)
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    from dex_retargeting_amd._build import source_hash

    summary["source_sha16"] = source_hash(w)
except Exception as e:
    print("# no source hash:", e)
try:
    line = open(os.path.join(out, f"prof_{w}_bench.json")).read().strip().splitlines()[-1]
    b = json.loads(line)
    summary["batch"] = b["config"]["batch_per_gpu"]
    summary["bench_kernel_ms"] = b["roofline"]["kernel_ms"]
    print(f"# bench line of the profiled run: value {b['value']:.4g} {b['unit']}, kernel_ms (HIP events) {b['roofline']['kernel_ms']:.4f}")
except Exception as e:
    print("# no bench line:", e)

print(f"# rocprofv3 --kernel-trace --stats -- python bench.py --workload {w} --no-cpu-baseline --steps 20 --warmup 3")
print("# view top_kernels of the rocpd database: name, calls, total_us, avg_us, pct")
for p in dbs("kt"):
    rows = q(p, "select name,total_calls,total_duration,average,percentage from top_kernels")
    scale = 1.0  # the view reports microseconds
    for r in rows[:10]:
        print(f"{r[0][:100]:100s} {r[1]:6d} {r[2] * scale:12.3f} {r[3] * scale:10.3f} {r[4]:6.2f}")
    solve = [r for r in rows if "dexr" in r[0]]
    if solve:
        summary["rocprof_kernel_avg_us"] = {r[0][:60]: r[3] * scale for r in solve}
    try:
        rows = q(p, "select name, duration, grid_x, workgroup_x, lds_size, scratch_size, vgpr_count, accum_vgpr_count, sgpr_count "
                    "from kernels where name like '%dexr%' order by start")
        print("\n# per dispatch (view kernels): duration_us grid wg lds scratch vgpr agpr sgpr")
        for r in rows[:30]:
            print(f"{r[0][:60]:60s} {r[1] * 1e-3:10.3f} " + " ".join(str(v) for v in r[2:]))
        # steady state: the first four dispatches of a solve kernel are the untimed staging launches (cold start from the
        # joint-limit midpoint, 3-4 x the iterations of a tracking step) -- they are not part of any timed region
        by = collections.defaultdict(list)
        for r in rows:
            by[r[0][:60]].append(r[1] * 1e-3)
        print("\n# steady state (dispatches after the four staging launches): name, n, mean_us, min_us, max_us")
        summary["rocprof_kernel_steady_avg_us"] = {}
        for k, v in by.items():
            tail = v[4:] if len(v) > 8 else v
            print(f"{k:60s} {len(tail):4d} {sum(tail) / len(tail):10.3f} {min(tail):10.3f} {max(tail):10.3f}")
            summary["rocprof_kernel_steady_avg_us"][k] = sum(tail) / len(tail)
    except Exception as e:
        print("# (per-dispatch view unavailable:", e, ")")

for tag in ("fetch", "write", "sq"):
    print(f"\n# separate PMC pass ({tag}): counter, dispatches, per-dispatch average; FETCH/WRITE_SIZE in KB as reported "
          f"(gfx950: FETCH_SIZE tallies 128-B requests at 64 B -> x2, MI355X_MICROARCH.md)")
    for p in dbs(tag):
        agg = collections.defaultdict(list)
        for k, c, v in q(p, "select kernel_name, counter_name, value from counters_collection"):
            if "dexr" in k:
                agg[c].append(v)
        for c, v in sorted(agg.items()):
            # the first dispatches are the untimed warm-start / warm-up launches: report the steady state (last half)
            tail = v[len(v) // 2:]
            avg = sum(tail) / len(tail)
            print(f"{c:24s} n={len(v):3d} avg(last half)={avg:.6g} last={v[-1]:.6g}")
            summary[c] = avg
if "FETCH_SIZE" in summary and "WRITE_SIZE" in summary:
    summary["hbm_bytes_per_launch"] = (2.0 * summary["FETCH_SIZE"] + summary["WRITE_SIZE"]) * 1024.0
    print(f"\n# HBM traffic per launch = 2 x FETCH_SIZE + WRITE_SIZE = {summary['hbm_bytes_per_launch'] / 1e6:.2f} MB")
json.dump(summary, open(os.path.join(out, f"pmc_{w}.json"), "w"), indent=1)
