#!/bin/bash
# Round 4, GPU call 1: test suite, headline experiments (priority variants, queue knobs, float64 tip variants), wave trace,
# the full default bench line.  Everything lands in gpurun_out/r4a_*.
R=${GRAFT_REPO_ROOT:-/root/repo}
cd "$R"; O=$R/gpurun_out; mkdir -p "$O"
P=$R/tools/_prof
( timeout 900 python -m pytest tests -m gpu -q -p no:cacheprovider 2>&1 | tail -40 ) > "$O/r4a_tests.txt"
cp -f "$O/all_configs_parity.txt" "$O/r4a_all_configs_parity.txt" 2>/dev/null
cp -f "$O/parity_ceilings_measured.json" "$O/r4a_parity_ceilings_measured.json" 2>/dev/null
{
  echo "## default library"; python tools/small_latency.py 64 4096 16384 65536 262144
  for v in prio2 prio4; do echo "## $v"; DEXR_LIB=$P/libdexr_$v.so python tools/small_latency.py 16384 65536; done
  echo "## default again"; python tools/small_latency.py 65536
  for k in "persist_from=1,persist_occ=2,qchunk=64" "persist_from=1,persist_occ=3,qchunk=64" "persist_from=1,persist_occ=4,qchunk=64" "persist_from=1,persist_occ=2,qchunk=256"; do
    echo "## knobs $k"; DEXR_TOOL_KNOBS=$k python tools/small_latency.py 65536
  done
  echo "## float64: default (tip, 2 waves/SIMD)"; python tools/small_latency.py f64 64 16384 65536
  for v in tip64w1 tip64w3; do echo "## float64 $v"; DEXR_LIB=$P/libdexr_$v.so python tools/small_latency.py f64 16384 65536; done
  echo "## float64 generic kernel (chain=0)"; DEXR_TOOL_KNOBS=chain=0 python tools/small_latency.py f64 65536
} > "$O/r4a_small_latency.txt" 2>&1
{
  bash tools/wave_trace.sh run teleop/allegro_hand_right.yml 65536
  bash tools/wave_trace.sh run teleop/allegro_hand_right.yml 16384
  bash tools/wave_trace.sh run teleop/allegro_hand_right.yml 65536 f64
} > "$O/r4a_wave_trace.txt" 2>&1
timeout 900 python bench.py > "$O/r4a_bench.json" 2> "$O/r4a_bench.err"
tail -c 600 "$O/r4a_bench.err"
tail -5 "$O/r4a_tests.txt"
