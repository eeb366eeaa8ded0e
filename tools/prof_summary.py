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

# Which kernels run INSIDE a timed step, and how often: the kernel-trace run and the PMC runs are the same command with KS / PS
# timed steps (+ 3 / + 1 warm-up steps), so a kernel dispatched c times per step appears c x (KS + 3 - PS - 1) more often in the
# trace than in a PMC pass; staging / diagnostic launches appear equally often in both and drop out.  A step's counters are the
# sum over its kernels -- the solve kernels AND the ordering / bucketing kernels around them (round 4 counted kernels named
# dexr::* only: the hard-frames-first pass, whose kernels live in an anonymous namespace, was missing from the traffic figure).
KS = int(sys.argv[3]) if len(sys.argv) > 3 else 20
PS = int(sys.argv[4]) if len(sys.argv) > 4 else 5
trace_counts = collections.Counter()
for p in dbs("kt"):
    try:
        for name, n in q(p, "select name, count(*) from kernels group by name"):
            trace_counts[name] += n
    except Exception as e:
        print("# (kernel counts unavailable:", e, ")")
per_kernel = {}
for tag in ("fetch", "write", "sq"):
    print(f"\n# separate PMC pass ({tag}): per kernel of a timed step -- dispatches per step, counter, per-dispatch average over the "
          f"timed steps; FETCH/WRITE_SIZE in KB as reported (gfx950: FETCH_SIZE tallies 128-B requests at 64 B -> x2, MI355X_MICROARCH.md)")
    for p in dbs(tag):
        agg = collections.defaultdict(lambda: collections.defaultdict(list))
        try:
            rows = q(p, "select kernel_name, counter_name, value from counters_collection order by dispatch_id")
        except Exception:
            rows = q(p, "select kernel_name, counter_name, value from counters_collection")
        for k, c, v in rows:
            agg[k][c].append(v)
        for k, cs in agg.items():
            n_pmc = len(next(iter(cs.values())))
            per_step = (trace_counts.get(k, 0) - n_pmc) / float(KS + 2 - PS) if trace_counts else (1.0 if "dexr" in k else 0.0)
            c_k = int(round(per_step))
            if c_k <= 0 or abs(per_step - c_k) > 0.2:
                continue  # not a kernel of the timed loop
            for c, v in sorted(cs.items()):
                tail = v[-c_k * PS:]
                avg = sum(tail) / len(tail)
                print(f"{k[:70]:70s} x{c_k} {c:20s} n={len(v):3d} avg={avg:.6g}")
                summary[c] = summary.get(c, 0.0) + c_k * avg
                per_kernel.setdefault(k[:70], {"per_step": c_k})[c] = avg
if per_kernel:
    summary["per_kernel"] = per_kernel
if "FETCH_SIZE" in summary and "WRITE_SIZE" in summary:
    summary["hbm_bytes_per_launch"] = (2.0 * summary["FETCH_SIZE"] + summary["WRITE_SIZE"]) * 1024.0
    print(f"\n# HBM traffic per step (all kernels of the step) = 2 x FETCH_SIZE + WRITE_SIZE = {summary['hbm_bytes_per_launch'] / 1e6:.2f} MB")
    for k, v in per_kernel.items():
        if "FETCH_SIZE" in v and "WRITE_SIZE" in v:
            print(f"#   {k:70s} x{v['per_step']}: {(2.0 * v['FETCH_SIZE'] + v['WRITE_SIZE']) * 1024.0 / 1e6:8.2f} MB per dispatch")
json.dump(summary, open(os.path.join(out, f"pmc_{w}.json"), "w"), indent=1)
