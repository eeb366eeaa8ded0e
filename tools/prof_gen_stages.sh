#!/bin/bash
# Per-stage cycles of the general kernel's passes (block 0, core clock; -DDEXR_GEN_PROF=1).
#   bash tools/prof_gen_stages.sh build     HERE: profiling copy of the library -> tools/_prof/libdexr_gprof.so
#   bash tools/prof_gen_stages.sh run       ON THE GPU BOX: a lone wave (B = 1) and a full launch of the arm + hand models
set -u
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd "$R"
if [ "${1:-run}" = build ]; then
  exec bash tools/build_variant.sh gprof "-DDEXR_GEN_PROF=1" "dexr_gen|dexr_api"
fi
export DEXR_LIB=$R/tools/_prof/libdexr_gprof.so
python - <<'PY'
import os, sys
import numpy as np
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "tests"))
import bench_data
from dex_retargeting_amd.constants import DEFAULT_URDF_DIR
from dex_retargeting_amd.retargeting_config import RetargetingConfig
from test_generic_tables import arm_hand_config
RetargetingConfig.set_default_urdf_dir(str(DEFAULT_URDF_DIR))
for kind in ("position", "vector"):
    seq = RetargetingConfig.from_dict(arm_hand_config(kind)).build()
    m = seq.optimizer.device_model()
    for B in (1, 65536):
        kp = bench_data.human_keypoints(B + 1)
        mid = np.repeat(seq.joint_limits.mean(1)[None], B, 0).astype(np.float32)
        last = m.retarget(np.ascontiguousarray(kp[:-1]), None, mid, keypoints=True)
        print(kind, "B =", B, "(tracking)", file=sys.stderr)
        m.retarget(np.ascontiguousarray(kp[1:]), None, last, keypoints=True)
PY
