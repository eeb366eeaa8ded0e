#!/bin/bash
# Run ON THE GPU BOX: everything the round's record rests on -- the GPU suite, smoke(), the default bench line, the
# launcher path with one rank for all three multi-GPU workloads.
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out
T=${1:-r4z}
cd "$R"
timeout 2400 python -m pytest tests -m gpu -q > "$O/${T}_tests.txt" 2>&1
grep "passed\|failed" "$O/${T}_tests.txt" | tail -1
timeout 600 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | grep -v amdgpu | tail -2
timeout 900 python bench.py > "$O/${T}_bench_default.json" 2> "$O/${T}_bench_default.err"
for w in allegro_vector leap_position mixed_fleet; do
  timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29533 bench.py --gpus 1 --workload $w --headline-only --no-cpu-baseline > "$O/${T}_1rank_$w.json" 2> "$O/${T}_1rank_$w.err"
  python - "$O/${T}_1rank_$w.json" <<'PY'
import json,sys
try:
    d=json.loads([l for l in open(sys.argv[1]) if l.startswith("{")][-1])
    m=d["multi_gpu"]
    print(sys.argv[1].split("/")[-1], "n_gpus", d["n_gpus"], "rccl_world_size", d["config"].get("rccl_world_size"), "%.4g frames/s" % d["value"], "%.4f ms" % d["ms_per_step"], "k", m.get("steps_per_gather"), "no gather", round(m["no_gather"]["ms_per_step"],4), "serial", round(m["gather_on_solve_stream"]["ms_per_step"],4), "graph", m.get("graph_replay"))
except Exception as e: print(sys.argv[1], "ERR", e)
PY
done
timeout 600 python bench.py --workload mixed_fleet > "$O/${T}_bench_fleet.json" 2> "$O/${T}_bench_fleet.err"
python - "$O/${T}_bench_default.json" "$O/${T}_bench_fleet.json" <<'PY'
import json,sys
for f in sys.argv[1:]:
    d=json.loads([l for l in open(f) if l.startswith("{")][-1])
    print(f.split("/")[-1], "%.4g" % d["value"], "%.4f ms" % d["ms_per_step"], "frac", d["roofline"]["frac"], "cpu", d.get("cpu_baseline",{}).get("value"))
PY
