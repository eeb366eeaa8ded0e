#!/bin/bash
# Round 4, GPU call 4: sixteen-lane kernel A/B inside one box -- current (LDS diet + XCD-local hand-out) vs the same without
# the XCD-local hand-out vs the round-3 header; the N > 1 bench path with one rank.
R=${GRAFT_REPO_ROOT:-/root/repo}
cd "$R"; O=$R/gpurun_out; mkdir -p "$O"
P=$R/tools/_prof
bash tools/ab_configs.sh "teleop/shadow_hand_right_dexpilot,offline/leap_hand_right,teleop/leap_hand_right_dexpilot,offline/shadow_hand_right" "$R/dex_retargeting_amd/libdexr.so" "$P/libdexr_wide_noxcd.so" "$P/libdexr_wide_r3.so" > "$O/r4d_wide_ab.txt" 2>&1
timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29533 bench.py --gpus 1 --headline-only --no-cpu-baseline > "$O/r4d_bench_1rank.json" 2> "$O/r4d_bench_1rank.err"
sort -k5,5 -k1,1 "$O/r4d_wide_ab.txt"
tail -c 300 "$O/r4d_bench_1rank.err"
