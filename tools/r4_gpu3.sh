#!/bin/bash
# Round 4, GPU call 3: test suite (XCD-local hand-out of the sixteen-lane kernel, general kernel's second-order sweep),
# general-kernel timings, headline block policy + priority-on-rejection variant, the N > 1 bench path with one rank,
# PMC traffic of the Shadow DexPilot launch.
R=${GRAFT_REPO_ROOT:-/root/repo}
cd "$R"; O=$R/gpurun_out; mkdir -p "$O"
P=$R/tools/_prof
( timeout 900 python -m pytest tests -m gpu -q -p no:cacheprovider 2>&1 | tail -30 ) > "$O/r4c_tests.txt"
python tools/gen_kernel_timing.py > "$O/r4c_gen_timing.txt" 2>&1
{
  echo "## default library (blocks of 8 waves from 4096 waves)"; python tools/small_latency.py 16384 65536 262144
  echo "## prio99 (priority for waves that hold a frame with a rejected step)"; DEXR_LIB=$P/libdexr_prio99.so python tools/small_latency.py 16384 65536
  echo "## default again"; python tools/small_latency.py 65536
} > "$O/r4c_small_latency.txt" 2>&1
timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29533 bench.py --gpus 1 --headline-only --no-cpu-baseline > "$O/r4c_bench_1rank.json" 2> "$O/r4c_bench_1rank.err"
bash tools/profile_round.sh shadow_dexpilot > "$O/r4c_profile_shadow.txt" 2>&1
tail -4 "$O/r4c_tests.txt"; cat "$O/r4c_gen_timing.txt" | grep -v "^/opt" | head -12; tail -3 "$O/r4c_profile_shadow.txt"
