#!/bin/bash
# What the slow frames of a launch spend their passes on (sixteen-lane kernel; -DDEXR_WIDE_DIAG=1: rejected steps, steps cut by
# the trust radius and failed factorisations counted per frame and returned in the upper bytes of the iteration count).
#   bash tools/pass_composition.sh build             HERE: diagnostic copy of the library -> tools/_prof/libdexr_wdiag.so
#   bash tools/pass_composition.sh run [configs]     ON THE GPU BOX
set -u
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd "$R"
if [ "${1:-run}" = build ]; then
  export DEXR_BUILD_DIR=$R/build_wdiag DEXR_LIB_OUT=$R/tools/_prof/libdexr_wdiag.so DEXR_EXTRA_FLAGS="-DDEXR_WIDE_DIAG=1"
  mkdir -p $DEXR_BUILD_DIR $R/tools/_prof
  for f in "$R"/build/*.o; do b=$(basename $f); case $b in dexr_wide*) ;; *) cp -pu $f $DEXR_BUILD_DIR/ ;; esac; done
  python -m dex_retargeting_amd._build
  exit $?
fi
shift
export DEXR_LIB=$R/tools/_prof/libdexr_wdiag.so
python - "$@" <<'PY'
import os, sys
sys.path.insert(0, os.getcwd())
import numpy as np
from dex_retargeting_amd.constants import DEFAULT_URDF_DIR
from dex_retargeting_amd.retargeting_config import RetargetingConfig
from oracle import cases  # (input recipes only)
RetargetingConfig.set_default_urdf_dir(str(DEFAULT_URDF_DIR))
rels = sys.argv[1:] or ["teleop/shadow_hand_right_dexpilot.yml", "offline/leap_hand_right.yml", "teleop/leap_hand_left_dexpilot.yml"]
for rel in rels:
    seq = RetargetingConfig.load_from_file(os.path.join(cases.CONFIG_DIR, rel)).build()
    prob = cases.problem_from_config(rel)
    model = seq.optimizer.device_model()
    B = 65536
    dex = prob.kind == "dexpilot"
    kp = cases.human_keypoints(B + 1, seed=cases.SEED)
    mid = np.repeat(prob.joint_limits.mean(1)[None], B, 0).astype(np.float32)
    st0 = np.zeros(B, np.uint32) if dex else None
    last = model.retarget(np.ascontiguousarray(kp[:-1]), None, mid, state=st0, keypoints=True)
    q, info = model.retarget(np.ascontiguousarray(kp[1:]), None, last, state=None if st0 is None else st0.copy(), keypoints=True, want_info=True)
    v = info["iters"].astype(np.int64)
    it, rej, cap, fail = v & 255, (v >> 8) & 255, (v >> 16) & 255, (v >> 24) & 255
    print(rel, "kernel", model.kernel(), "iterations mean %.2f max %d" % (it.mean(), it.max()))
    for lo, hi in ((0, 8), (8, 16), (16, 24), (24, 40), (40, 256)):
        m = (it >= lo) & (it < hi)
        if m.any():
            print(f"   [{lo:2d},{hi:3d}) iterations: {m.sum():6d} frames, {it[m].sum() / it.sum():.3f} of all iterations; rejected {rej[m].sum() / it[m].sum():.2f}"
                  f" (failed factorisation {fail[m].sum() / it[m].sum():.2f}), cut by the trust radius {cap[m].sum() / it[m].sum():.2f}")
PY
