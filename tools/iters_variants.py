#!/usr/bin/env python3
"""Developer tool (GPU): iteration histograms of one workload under kernel-selection / arithmetic variants."""
import os
import sys

import numpy as np

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
import torch  # noqa: E402,F401

from dex_retargeting_amd.constants import DEFAULT_URDF_DIR  # noqa: E402
from dex_retargeting_amd.retargeting_config import RetargetingConfig  # noqa: E402
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import _tune  # noqa: E402
from oracle import cases  # noqa: E402  (input recipes only)

rel = sys.argv[1] if len(sys.argv) > 1 else "offline/leap_hand_right.yml"
B = int(sys.argv[2]) if len(sys.argv) > 2 else 8192
RetargetingConfig.set_default_urdf_dir(str(DEFAULT_URDF_DIR))
prob = cases.problem_from_config(rel)
dexpilot = prob.kind == "dexpilot"
kp = cases.human_keypoints(B + 1, seed=cases.SEED)
ref = np.ascontiguousarray(cases.ref_from_keypoints(prob, kp), dtype=np.float32)
mid = np.repeat(prob.joint_limits.mean(1)[None], B, 0).astype(np.float32)


def model_for(knobs):
    seq = RetargetingConfig.load_from_file(os.path.join(cases.CONFIG_DIR, rel)).build()
    m = seq.optimizer.device_model()
    _, opts = _tune.apply(m, knobs)
    return m, opts


m0, _ = model_for({})
st = (lambda: np.zeros(B, np.uint32)) if dexpilot else (lambda: None)
last = m0.retarget(ref[:-1], None, mid, state=st())
q64, i64 = m0.retarget_f64(ref[1:], None, last, state=st(), want_info=True)
print(f"# {rel} B={B}")
print("float64 register kernel: iters mean %.2f" % i64["iters"].mean(), np.bincount(i64["iters"]).tolist())
for name, env in (("default", {}), ("register f32 + polish", {"kernel": "register"}),
                  ("force big", {"kernel": "lds"}), ("force quad", {"kernel": "quad"}),
                  ("default, gauss-newton", {"newton": 0})):
    try:
        m, opts = model_for(env)
        q, info = m.retarget(ref[1:], None, last, state=st(), opts=opts, want_info=True)
    except Exception as e:  # variant not available for this model
        print(f"{name}: {e}")
        continue
    dq = np.abs(q.astype(np.float64) - q64).max(1)
    print(f"{name}: iters mean {info['iters'].mean():.2f} max dq {dq.max():.2e} status!=0 {(info['status'] != 0).sum()}",
          np.bincount(info["iters"]).tolist())
