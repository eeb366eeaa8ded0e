#!/bin/bash
# Run ON THE GPU BOX: rocprofv3 kernel trace + separate PMC passes around tools/prof_config.py.
# Usage: bash tools/prof_kernel.sh <tag> <config.yml> [kernel] [batch]  -> gpurun_out/kprof_<tag>.txt
set -u
TAG=$1; CFG=$2; KER=${3:-auto}; BATCH=${4:-65536}
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out
mkdir -p "$O"
cd /tmp && export TMPDIR=/tmp
rm -rf "$O"/kprof_${TAG}_*
{
echo "# $CFG kernel=$KER batch=$BATCH"
timeout 300 rocprofv3 --kernel-trace --stats -d "$O/kprof_${TAG}_kt" -- python "$R/tools/prof_config.py" "$CFG" "$KER" "$BATCH" 5 2>/dev/null | grep -v "^$" | tail -2
for P in "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_LDS" "SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR" "FETCH_SIZE" "WRITE_SIZE"; do
  D="$O/kprof_${TAG}_pmc_$(echo $P | tr ' ' '_' | cut -c1-24)"
  timeout 300 rocprofv3 --pmc $P -d "$D" -- python "$R/tools/prof_config.py" "$CFG" "$KER" "$BATCH" 3 > /dev/null 2>&1
done
python "$R/tools/pmc_summary.py" $(find "$O"/kprof_${TAG}_* -name "*.db") 2>/dev/null
} > "$O/kprof_${TAG}.txt" 2>&1
rm -rf "$O"/kprof_${TAG}_kt "$O"/kprof_${TAG}_pmc_*
cat "$O/kprof_${TAG}.txt"
