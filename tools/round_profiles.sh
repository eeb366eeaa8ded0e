#!/bin/bash
# Run ON THE GPU BOX: rocprofv3 kernel-trace + separate PMC passes of the three single-GPU bench workloads, the fleet's
# dispatch timeline, and the all-configs table.
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out
cd "$R"
for w in allegro_vector shadow_dexpilot leap_position; do
  bash tools/profile_round.sh $w > /dev/null 2>&1
  tail -12 "$O/prof_${w}_summary.txt"
done
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats -d $O/prof_fleet_kt -- python $R/bench.py --workload mixed_fleet --no-cpu-baseline --steps 6 --warmup 2 > $O/prof_fleet_bench.json 2> $O/prof_fleet.err
python $R/tools/timeline.py $O/prof_fleet_kt 30 > $O/prof_fleet_timeline.txt
rm -rf $O/prof_fleet_kt
cd "$R"
timeout 900 python tools/all_configs.py > "$O/all_configs_timing.txt" 2>&1
tail -3 "$O/all_configs_timing.txt"
