#!/bin/bash
# Run ON THE GPU BOX: the sixteen-lane kernel's tests, its per-stage cycles and the three bench configs (headline only).
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out
T=${1:-r3w}
cd "$R"
timeout 1500 python -m pytest tests/test_gpu_parity.py tests/test_gpu_all_configs.py -m gpu -x -q -k "sixteen or lane or wide or all_configs or full_size or ragged or fleet or longest or two_streams or sequence or fused" > "$O/${T}_tests.txt" 2>&1
tail -3 "$O/${T}_tests.txt"
bash tools/prof_wide_stages.sh run 2>&1 | grep wprof > "$O/${T}_wprof.txt"; cat "$O/${T}_wprof.txt"
for w in shadow_dexpilot leap_position; do
  timeout 600 python bench.py --workload $w --headline-only --no-cpu-baseline > "$O/${T}_bench_$w.json" 2> "$O/${T}_bench_$w.err"
  python -c "
import json,sys
d=json.loads([l for l in open('$O/${T}_bench_$w.json') if l.startswith('{')][-1])
print('$w', '%.4f ms' % d['ms_per_step'], 'kernel', d['roofline']['kernel_ms'], d['solver']['iters_mean'], d['solver']['iters_max'], d['parity']['max_abs_dq_rad'], d['parity']['frac_within_1e-4'])
"
done
timeout 600 python tools/all_configs.py > "$O/${T}_all_configs.txt" 2>&1; grep -c yml "$O/${T}_all_configs.txt"
