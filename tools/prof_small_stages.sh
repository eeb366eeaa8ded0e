#!/bin/bash
# Per-stage cycles of the small-component (register) kernel's passes (wave 0 of the launch, s_memtime; -DDEXR_SMALL_PROF=1).
#   bash tools/prof_small_stages.sh build            HERE: profiling copy of the library -> tools/_prof/libdexr_sprof.so
#   bash tools/prof_small_stages.sh run [configs]    ON THE GPU BOX: a lone wave per component (64 frames) and a full launch
set -u
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd "$R"
if [ "${1:-run}" = build ]; then
  export DEXR_BUILD_DIR=$R/build_sprof DEXR_LIB_OUT=$R/tools/_prof/libdexr_sprof.so DEXR_EXTRA_FLAGS="-DDEXR_SMALL_PROF=1"
  mkdir -p $DEXR_BUILD_DIR $R/tools/_prof
  # reuse the objects of the normal build for everything but the small-component solve kernels and the API
  for f in "$R"/build/*.o; do b=$(basename $f); case $b in dexr_inst_4_0_0.o|dexr_inst_8_0_0.o|dexr_inst_4_1_0.o|dexr_inst_8_1_0.o|dexr_inst_chain*|dexr_inst_ext*|dexr_inst_tip*|dexr_api.o) ;; *) cp -pu $f $DEXR_BUILD_DIR/ ;; esac; done
  python -m dex_retargeting_amd._build
  exit $?
fi
shift
export DEXR_LIB=$R/tools/_prof/libdexr_sprof.so
python - "$@" <<'PY'
import os, sys
import numpy as np
sys.path.insert(0, os.getcwd())
import bench_data
from dex_retargeting_amd.constants import DEFAULT_URDF_DIR
from dex_retargeting_amd.retargeting_config import RetargetingConfig
RetargetingConfig.set_default_urdf_dir(str(DEFAULT_URDF_DIR))
rels = sys.argv[1:] or ["teleop/allegro_hand_right.yml"]
for rel in rels:
    seq = RetargetingConfig.load_from_file(os.path.join(bench_data.CONFIG_DIR, rel)).build()
    opt = seq.optimizer
    for B in (64, 4096, 65536):
        kp = bench_data.human_keypoints(B + 1)
        mid = np.repeat(seq.joint_limits.mean(1)[None], B, 0).astype(np.float32)
        st = np.zeros(B, np.uint32) if opt.retargeting_type == "DEXPILOT" else None
        print(rel, "B =", B, "(cold start from the limit midpoint)", file=sys.stderr)
        last = opt.device_model().retarget(np.ascontiguousarray(kp[:-1]), None, mid, state=st, keypoints=True)
        for rep in range(2):
            print(rel, "B =", B, "(tracking)", file=sys.stderr)
            opt.device_model().retarget(np.ascontiguousarray(kp[1:]), None, last, state=st, keypoints=True)
PY
