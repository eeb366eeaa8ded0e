#!/bin/bash
# Run ON THE GPU BOX (gpurun): rocprofv3 kernel trace + separate PMC passes of the default bench command.
# Usage: bash tools/profile_round.sh [workload]   -> gpurun_out/prof_<workload>_summary.txt, gpurun_out/pmc_<workload>.json
set -u
W=${1:-allegro_vector}
R=${GRAFT_REPO_ROOT:-/root/repo}
# sub-records of the default line have no workload of their own: bench.py --probe times their loop alone
case $W in
  allegro_vector_f64) BA="--workload allegro_vector --probe f64" ;;
  allegro_vector_cold) BA="--workload allegro_vector --probe cold_start" ;;
  general_kernel) BA="--workload allegro_vector --probe general_kernel" ;;
  *) BA="--workload $W" ;;
esac
KS=20; PS=5
if [ $W = general_kernel ]; then KS=5; PS=3; fi
O=$R/gpurun_out
mkdir -p "$O"
cd /tmp && export TMPDIR=/tmp
rm -rf "$O"/prof_${W}_*
timeout 600 rocprofv3 --kernel-trace --stats -d "$O/prof_${W}_kt" -- python "$R/bench.py" $BA --no-cpu-baseline --headline-only --steps $KS --warmup 3 > "$O/prof_${W}_bench.json" 2> "$O/prof_${W}_kt.err"
timeout 600 rocprofv3 --pmc FETCH_SIZE -d "$O/prof_${W}_fetch" -- python "$R/bench.py" $BA --no-cpu-baseline --headline-only --steps $PS --warmup 1 > /dev/null 2> "$O/prof_${W}_fetch.err"
timeout 600 rocprofv3 --pmc WRITE_SIZE -d "$O/prof_${W}_write" -- python "$R/bench.py" $BA --no-cpu-baseline --headline-only --steps $PS --warmup 1 > /dev/null 2> "$O/prof_${W}_write.err"
timeout 600 rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_LDS SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY -d "$O/prof_${W}_sq" -- python "$R/bench.py" $BA --no-cpu-baseline --headline-only --steps $PS --warmup 1 > /dev/null 2> "$O/prof_${W}_sq.err"
python "$R/tools/prof_summary.py" "$W" "$O" $KS $PS > "$O/prof_${W}_summary.txt" 2>&1
# the raw databases are large: keep only the summaries
rm -rf "$O"/prof_${W}_kt "$O"/prof_${W}_fetch "$O"/prof_${W}_write "$O"/prof_${W}_sq
cat "$O/prof_${W}_summary.txt"
