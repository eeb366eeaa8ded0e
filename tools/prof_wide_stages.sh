#!/bin/bash
# Per-stage cycles of the sixteen-lane kernel's passes (wave 0 of the launch, s_memtime; -DDEXR_WIDE_PROF=1).
#   bash tools/prof_wide_stages.sh build            HERE: profiling copy of the library -> tools/_prof/libdexr_wprof.so
#   bash tools/prof_wide_stages.sh run [configs]    ON THE GPU BOX: a lone wave (4 frames) and a full launch per config
set -u
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd "$R"
if [ "${1:-run}" = build ]; then
  export DEXR_BUILD_DIR=$R/build_wprof DEXR_LIB_OUT=$R/tools/_prof/libdexr_wprof.so DEXR_EXTRA_FLAGS="-DDEXR_WIDE_PROF=1"
  mkdir -p $DEXR_BUILD_DIR $R/tools/_prof
  # reuse the objects of the normal build for everything that does not include dexr_wide.hpp
  for f in "$R"/build/*.o; do b=$(basename $f); case $b in dexr_wide*|dexr_api.o) ;; *) cp -pu $f $DEXR_BUILD_DIR/ ;; esac; done
  python -m dex_retargeting_amd._build
  exit $?
fi
shift
export DEXR_LIB=$R/tools/_prof/libdexr_wprof.so
python - "$@" <<'PY'
import os, sys
import numpy as np
sys.path.insert(0, os.getcwd())
import bench_data
from dex_retargeting_amd.constants import DEFAULT_URDF_DIR
from dex_retargeting_amd.retargeting_config import RetargetingConfig
RetargetingConfig.set_default_urdf_dir(str(DEFAULT_URDF_DIR))
rels = sys.argv[1:] or ["teleop/shadow_hand_right_dexpilot.yml", "offline/leap_hand_right.yml"]
for rel in rels:
    seq = RetargetingConfig.load_from_file(os.path.join(bench_data.CONFIG_DIR, rel)).build()
    opt = seq.optimizer
    for B in (4, 65536):
        kp = bench_data.human_keypoints(B + 1)
        mid = np.repeat(seq.joint_limits.mean(1)[None], B, 0).astype(np.float32)
        st = np.zeros(B, np.uint32) if opt.retargeting_type == "DEXPILOT" else None
        print(rel, "B =", B, "(cold start from the limit midpoint)", file=sys.stderr)
        last = opt.device_model().retarget(np.ascontiguousarray(kp[:-1]), None, mid, state=st, keypoints=True)
        print(rel, "B =", B, "(tracking)", file=sys.stderr)
        opt.device_model().retarget(np.ascontiguousarray(kp[1:]), None, last, state=st, keypoints=True)
PY
