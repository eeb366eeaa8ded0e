#!/bin/bash
# Run ON THE GPU BOX (gpurun): rocprofv3 kernel trace + separate PMC passes of the default bench command.
# Usage: bash tools/profile_round.sh [workload]   -> gpurun_out/prof_<workload>_summary.txt, gpurun_out/pmc_<workload>.json
set -u
W=${1:-allegro_vector}
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out
mkdir -p "$O"
cd /tmp && export TMPDIR=/tmp
rm -rf "$O"/prof_${W}_*
timeout 600 rocprofv3 --kernel-trace --stats -d "$O/prof_${W}_kt" -- python "$R/bench.py" --workload "$W" --no-cpu-baseline --headline-only --steps 20 --warmup 3 > "$O/prof_${W}_bench.json" 2> "$O/prof_${W}_kt.err"
timeout 600 rocprofv3 --pmc FETCH_SIZE -d "$O/prof_${W}_fetch" -- python "$R/bench.py" --workload "$W" --no-cpu-baseline --headline-only --steps 5 --warmup 1 > /dev/null 2> "$O/prof_${W}_fetch.err"
timeout 600 rocprofv3 --pmc WRITE_SIZE -d "$O/prof_${W}_write" -- python "$R/bench.py" --workload "$W" --no-cpu-baseline --headline-only --steps 5 --warmup 1 > /dev/null 2> "$O/prof_${W}_write.err"
timeout 600 rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_LDS SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY -d "$O/prof_${W}_sq" -- python "$R/bench.py" --workload "$W" --no-cpu-baseline --headline-only --steps 5 --warmup 1 > /dev/null 2> "$O/prof_${W}_sq.err"
python "$R/tools/prof_summary.py" "$W" "$O" > "$O/prof_${W}_summary.txt" 2>&1
# the raw databases are large: keep only the summaries
rm -rf "$O"/prof_${W}_kt "$O"/prof_${W}_fetch "$O"/prof_${W}_write "$O"/prof_${W}_sq
cat "$O/prof_${W}_summary.txt"
