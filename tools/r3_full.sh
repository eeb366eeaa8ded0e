#!/bin/bash
# Run ON THE GPU BOX: the whole GPU suite, then the default bench line, the 1-rank native-collective line and the fleet.
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out
T=${1:-r3f}
mkdir -p "$O"
cd "$R"
timeout 2400 python -m pytest tests -m gpu -x -q > "$O/${T}_tests.txt" 2>&1
tail -4 "$O/${T}_tests.txt"
timeout 900 python bench.py > "$O/${T}_bench_default.json" 2> "$O/${T}_bench_default.err"
timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29533 bench.py --gpus 1 --headline-only --no-cpu-baseline > "$O/${T}_bench_1rank_native.json" 2> "$O/${T}_bench_1rank_native.err"
timeout 600 python bench.py --workload mixed_fleet > "$O/${T}_bench_fleet.json" 2> "$O/${T}_bench_fleet.err"
python - "$T" <<'PY'
import json,os,sys
T=sys.argv[1]
O=os.environ.get("GRAFT_REPO_ROOT","/root/repo")+"/gpurun_out"
for f in (f"{T}_bench_default.json",f"{T}_bench_1rank_native.json",f"{T}_bench_fleet.json"):
    try:
        d=json.loads([l for l in open(os.path.join(O,f)) if l.startswith("{")][-1])
        print(f, "%.4g"%d["value"], "%.4f"%d["ms_per_step"], d.get("roofline",{}).get("kernel_ms"), "cpu", {k:(round(v["value"],1), v.get("cores")) for k,v in d.items() if k.startswith("cpu_baseline")})
        for k,v in d.get("also",{}).items(): print("  also", k, "%.4f"%v["ms_per_step"], v.get("two_streams",{}).get("ms_per_step"))
        if "online_teleop" in d:
            for k,v in d["online_teleop"]["robots"].items(): print("  online", k, round(v.get("mean_ms",0),4), round(v.get("p99_ms",0),4), v.get("c_abi_call",{}).get("p50_ms"), v.get("cpu_port_same_loop",{}).get("mean_ms"))
        if "multi_gpu" in d: print("  multi", d["multi_gpu"]["gather_on_solve_stream"]["ms_per_step"])
        print("  parity", d.get("parity",{}) if "subset" not in d.get("parity",{}) else {k:d["parity"][k] for k in ("max_abs_dq_rad","frac_within_1e-4")})
    except Exception as e: print(f, "ERR", e)
PY
