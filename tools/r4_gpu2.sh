#!/bin/bash
# Round 4, GPU call 2: VALU issue-cost microbenchmark, headline variants (scalar tip pass, block sizes), the test suite on the
# sixteen-lane kernel's new LDS layout, 24-row grid at three waves per SIMD.
R=${GRAFT_REPO_ROOT:-/root/repo}
cd "$R"; O=$R/gpurun_out; mkdir -p "$O"
P=$R/tools/_prof
$P/valu_rate > "$O/r4b_valu_rate.txt" 2>&1
( timeout 900 python -m pytest tests -m gpu -q -p no:cacheprovider 2>&1 | tail -30 ) > "$O/r4b_tests.txt"
{
  echo "## default library"; python tools/small_latency.py 16384 65536
  for v in tipscalar wpb1 wpb8 wpb16; do echo "## $v"; DEXR_LIB=$P/libdexr_$v.so python tools/small_latency.py 16384 65536; done
  echo "## default again"; python tools/small_latency.py 65536
} > "$O/r4b_small_latency.txt" 2>&1
bash tools/ab_configs.sh "teleop/shadow_hand_right_dexpilot,offline/leap_hand_right,teleop/shadow_hand_right.yml" "$R/dex_retargeting_amd/libdexr.so" "$P/libdexr_wide24w3.so" > "$O/r4b_wide24_ab.txt" 2>&1
tail -4 "$O/r4b_tests.txt"; cat "$O/r4b_wide24_ab.txt" | tail -20
