#!/bin/bash
# What a small-component launch's duration is made of: every wave's start / first-load / per-pass / retire times on the
# 100 MHz wall clock (-DDEXR_WAVE_TRACE=1, see dexr_kernel.hpp).
#   bash tools/wave_trace.sh build              HERE: tracing copy of the library -> tools/_prof/libdexr_wtrace.so
#   bash tools/wave_trace.sh run [config] [B]   ON THE GPU BOX: one tracking launch, summary on stdout
set -u
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd "$R"
if [ "${1:-run}" = build ]; then
  export DEXR_BUILD_DIR=$R/build_wtrace DEXR_LIB_OUT=$R/tools/_prof/libdexr_wtrace.so DEXR_EXTRA_FLAGS="-DDEXR_WAVE_TRACE=1 ${WTRACE_EXTRA:-}"
  mkdir -p $DEXR_BUILD_DIR $R/tools/_prof
  # reuse the objects of the normal build for everything but the small-component solve kernels and the API
  for f in "$R"/build/*.o; do b=$(basename $f); case $b in dexr_inst_4_0_0.o|dexr_inst_8_0_0.o|dexr_inst_4_1_0.o|dexr_inst_8_1_0.o|dexr_inst_chain*|dexr_inst_ext*|dexr_inst_tip*|dexr_api.o) ;; *) cp -pu $f $DEXR_BUILD_DIR/ ;; esac; done
  python -m dex_retargeting_amd._build
  exit $?
fi
shift
export DEXR_LIB=${WTRACE_LIB:-$R/tools/_prof/libdexr_wtrace.so}
python - "$@" <<'PY'
import os, sys
import numpy as np
sys.path.insert(0, os.getcwd())
import bench_data
from dex_retargeting_amd import _lib
from dex_retargeting_amd.constants import DEFAULT_URDF_DIR
from dex_retargeting_amd.retargeting_config import RetargetingConfig
RetargetingConfig.set_default_urdf_dir(str(DEFAULT_URDF_DIR))
rel = sys.argv[1] if len(sys.argv) > 1 else "teleop/allegro_hand_right.yml"
B = int(sys.argv[2]) if len(sys.argv) > 2 else 65536
f64 = len(sys.argv) > 3 and sys.argv[3] == "f64"
seq = RetargetingConfig.load_from_file(os.path.join(bench_data.CONFIG_DIR, rel)).build()
model = seq.optimizer.device_model()
kp = bench_data.human_keypoints(B + 1)
mid = np.repeat(seq.joint_limits.mean(1)[None], B, 0).astype(np.float32)
last = model.retarget(np.ascontiguousarray(kp[:-1]), None, mid, keypoints=True)
opts = _lib.default_options(precision=1) if f64 else None
for rep in range(3):  # the trace file holds the last launch
    q, info = model.retarget(np.ascontiguousarray(kp[1:]), None, last, keypoints=True, want_info=True, opts=opts)
d = np.fromfile("/tmp/dexr_wave_trace.bin").reshape(-1, 24)
d = d[d[:, 0] > 0]
t0 = d[:, 0].min()
us = lambda v: (v - t0) / 100.0  # 100 MHz -> us
start, loaded, end, n = us(d[:, 0]), us(d[:, 1]), us(d[:, 2]), d[:, 3].astype(int)
print(f"# {rel} B={B} {'f64' if f64 else 'f32'} kernel {model.kernel()}: {len(d)} waves; iters mean {info['iters'].mean():.2f} max {info['iters'].max()}")
q_ = lambda a: " ".join(f"{np.percentile(a, p):7.2f}" for p in (0, 10, 50, 90, 99, 100))
print("percentiles                 min     p10     p50     p90     p99     max   [us]")
print("wave start              ", q_(start))
print("first frames loaded     ", q_(loaded))
print("  load phase            ", q_(loaded - start))
print("wave retired            ", q_(end))
print("  lifetime              ", q_(end - start))
print("passes per wave         ", q_(n))
P = np.where(d[:, 8:24] > 0, us(d[:, 8:24]), np.nan)
dur = np.diff(np.concatenate([loaded[:, None], P], 1), axis=1)
print("duration of pass k over the waves that ran it: k, waves, p10 / p50 / p90 [us], alive waves at its median end time")
for k in range(16):
    v = dur[:, k][~np.isnan(dur[:, k])]
    if len(v) == 0:
        break
    tmed = np.nanmedian(P[:, k])
    print(f"  pass {k:2d}  {len(v):5d}   {np.percentile(v, 10):6.2f} {np.percentile(v, 50):6.2f} {np.percentile(v, 90):6.2f}    alive {int(((start <= tmed) & (end >= tmed)).sum()):5d}")
edges = np.arange(0, end.max() + 2, 2.0)
alive = [int(((start <= t) & (end > t)).sum()) for t in edges]
print("alive waves every 2 us:", " ".join(str(a) for a in alive))
hw = d[:, 4].astype(np.int64)
xcc = d[:, 5].astype(np.int64) & 0xF
for x in sorted(set(xcc)):
    m = xcc == x
    print(f"  XCC {x}: {int(m.sum()):5d} waves, start p50 {np.median(start[m]):6.2f} us, retire max {end[m].max():6.2f} us")
slow = np.argsort(-end)[:8]
print("the 8 waves that retire last: wave, start, loaded, end, passes, pass durations")
for w in slow:
    print(f"  {w:5d} {start[w]:6.2f} {loaded[w]:6.2f} {end[w]:6.2f} {n[w]:3d}  " + " ".join(f"{x:.2f}" for x in dur[w][~np.isnan(dur[w])]))
PY
