#!/bin/bash
# Run ON THE GPU BOX: the round's rocprofv3 kernel-trace / PMC summaries for EVERY record of the default bench line, then the default
# bench line itself, the mixed-fleet line, the one-rank RCCL lines and the 39-config timing table.  Outputs under gpurun_out/
# (pmc_<name>.json, prof_<name>_summary.txt, r06_bench_*.json / .log, r06_all_configs.txt).
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out
cd "$R"
for w in allegro_vector allegro_vector_f64 allegro_vector_cold shadow_dexpilot leap_position mixed_fleet general_kernel; do
  bash tools/profile_round.sh $w > /dev/null 2>&1
  grep "HBM traffic per step" "$O/prof_${w}_summary.txt" | sed "s/^/$w: /"
done
# the committed PMC summaries decide whether the bench line carries traffic figures: copy them in place BEFORE the bench runs
cp "$O"/pmc_*.json "$R/profiles/" 2>/dev/null
(time python bench.py --steps 20 --warmup 5) > "$O/r06_bench_default.log" 2>&1
tail -1 "$O/r06_bench_default.log" > /dev/null
cp "$O/bench_detail.json" "$O/r06_bench_detail_default.json" 2>/dev/null
python bench.py --workload mixed_fleet --steps 20 --warmup 5 > "$O/r06_bench_mixed_fleet.log" 2>&1
for w in allegro_vector leap_position mixed_fleet; do
  python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29571 bench.py --gpus 1 --workload $w --steps 20 --warmup 5 --headline-only --no-cpu-baseline > "$O/r06_bench_1rank_native_rccl_$w.log" 2>&1
done
python tools/all_configs.py 65536 > "$O/r06_all_configs.txt" 2>/dev/null
for f in r06_bench_default r06_bench_mixed_fleet r06_bench_1rank_native_rccl_allegro_vector r06_bench_1rank_native_rccl_leap_position r06_bench_1rank_native_rccl_mixed_fleet; do
  grep -v "^DETAIL\|^real\|^user\|^sys\|^$" "$O/$f.log" | tail -1 > "$O/$f.json"
  echo "$f: $(head -c 240 "$O/$f.json")"
done
