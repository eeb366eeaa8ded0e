#!/usr/bin/env python3
"""Developer tool (GPU): where does the LDS kernel's answer sit on mimic position models?  Oracle gradient at the answer."""
import os
import sys

import numpy as np

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
import torch  # noqa: E402,F401

from dex_retargeting_amd.constants import DEFAULT_URDF_DIR  # noqa: E402
from dex_retargeting_amd.retargeting_config import RetargetingConfig  # noqa: E402
from oracle import cases  # noqa: E402

rel = sys.argv[1] if len(sys.argv) > 1 else "offline/ability_hand_right.yml"
RetargetingConfig.set_default_urdf_dir(str(DEFAULT_URDF_DIR))
prob = cases.problem_from_config(rel)
B = 4096
kp = cases.human_keypoints(B + 1, seed=cases.SEED)
ref = np.ascontiguousarray(cases.ref_from_keypoints(prob, kp), dtype=np.float32)
mid = np.repeat(prob.joint_limits.mean(1)[None], B, 0).astype(np.float32)
model = RetargetingConfig.load_from_file(os.path.join(cases.CONFIG_DIR, rel)).build().optimizer.device_model()
last = model.retarget(ref[:-1], None, mid)
q32, info = model.retarget(ref[1:], None, last, want_info=True)
q64 = model.retarget_f64(ref[1:], None, last)
lo, hi = prob.bounds
last64 = last.astype(np.float64)


def pg(x):
    f, g, _ = prob.evaluate(x, ref[1:], None, last64)
    G = g + 2 * prob.norm_delta * (x - last64)
    act = ((x <= lo + 1e-9) & (G > 0)) | ((x >= hi - 1e-9) & (G < 0))
    return f + prob.norm_delta * ((x - last64) ** 2).sum(1), np.where(act, 0.0, G)


F32, G32 = pg(q32.astype(np.float64))
F64, G64 = pg(q64)
dq = np.abs(q32 - q64).max(1)
print(f"# {rel}: |PG|inf at float32 answer: median {np.median(np.abs(G32).max(1)):.2e} max {np.abs(G32).max():.2e};"
      f" at float64 answer: median {np.median(np.abs(G64).max(1)):.2e} max {np.abs(G64).max():.2e}")
names = prob.target_joint_names if hasattr(prob, "target_joint_names") else list(range(prob.n_opt))
worst = np.argsort(-dq)[:8]
for b in worst:
    j = int(np.abs(q32[b] - q64[b]).argmax())
    print(f"frame {b}: dq {dq[b]:.2e} (joint {j} {names[j] if j < len(names) else ''}) F32-F64 {F32[b] - F64[b]:+.2e} |PG32| {np.abs(G32[b]).max():.2e} "
          f"|PG64| {np.abs(G64[b]).max():.2e} iters {info['iters'][b]} status {info['status'][b]}")
    print("    dq per joint:", np.array2string(q32[b] - q64[b], precision=1, max_line_width=200))
    print("    PG32        :", np.array2string(G32[b], precision=1, max_line_width=200))
per_joint = np.abs(q32 - q64)
print("p99.9 |dq| per joint:", np.array2string(np.percentile(per_joint, 99.9, axis=0), precision=1, max_line_width=200))
