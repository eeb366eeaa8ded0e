#!/bin/bash
# Run ON THE GPU BOX: new round-3 tests, the default bench line, the 1-rank native-collective run and the fleet bench.
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out
mkdir -p "$O"
cd "$R"
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "host_pointer or native_allgather or damping_default or mixed_fleet or native_fleet or two_streams or device_pointer" > "$O/r3a_tests.txt" 2>&1
tail -5 "$O/r3a_tests.txt"
timeout 900 python bench.py > "$O/r3a_bench_default.json" 2> "$O/r3a_bench_default.err"
tail -c 600 "$O/r3a_bench_default.err"
timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29533 bench.py --gpus 1 --headline-only --no-cpu-baseline > "$O/r3a_bench_1rank_native.json" 2> "$O/r3a_bench_1rank_native.err"
tail -c 600 "$O/r3a_bench_1rank_native.err"
timeout 600 python bench.py --workload mixed_fleet --no-cpu-baseline > "$O/r3a_bench_fleet.json" 2> "$O/r3a_bench_fleet.err"
tail -c 600 "$O/r3a_bench_fleet.err"
python - <<'PY'
import json,os
O=os.environ.get("GRAFT_REPO_ROOT","/root/repo")+"/gpurun_out"
for f in ("r3a_bench_default.json","r3a_bench_1rank_native.json","r3a_bench_fleet.json"):
    try:
        d=json.loads([l for l in open(os.path.join(O,f)) if l.startswith("{")][-1])
        print(f, d["value"], d["ms_per_step"], d.get("roofline",{}).get("kernel_ms"))
        if "online_teleop" in d:
            for k,v in d["online_teleop"]["robots"].items(): print("  online", k, {a:(round(b,4) if isinstance(b,float) else b) for a,b in v.items() if a!="c_abi_call" and a!="cpu_port_same_loop"}, v.get("c_abi_call",{}).get("p50_ms"))
        if "multi_gpu" in d: print("  multi", d["multi_gpu"]["gather_on_solve_stream"], d["multi_gpu"]["rccl_version"])
    except Exception as e: print(f, "ERR", e)
PY
