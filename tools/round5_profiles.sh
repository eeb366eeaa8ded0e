#!/bin/bash
# Run ON THE GPU BOX: the round's PMC / kernel-trace summaries for EVERY record of the default bench line, then the
# default bench line itself.  Outputs under gpurun_out/ (pmc_<name>.json, prof_<name>_summary.txt, r05_bench_default.*).
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out
cd "$R"
for w in allegro_vector allegro_vector_f64 allegro_vector_cold shadow_dexpilot leap_position mixed_fleet general_kernel; do
  bash tools/profile_round.sh $w > /dev/null 2>&1
  grep "HBM traffic per step" "$O/prof_${w}_summary.txt" | sed "s/^/$w: /"
done
