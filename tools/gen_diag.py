#!/usr/bin/env python3
"""Developer tool (GPU): per-frame iteration counts of the general kernel on the arm + hand models (tracking workload) next
to the oracle's LM, and the frames that take longest.   python tools/gen_diag.py [position|vector] [B]"""
import os
import sys

import numpy as np

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
sys.path.insert(0, os.path.join(REPO, "tests"))
import bench_data  # noqa: E402
from dex_retargeting_amd.constants import DEFAULT_URDF_DIR  # noqa: E402
from dex_retargeting_amd.retargeting_config import RetargetingConfig  # noqa: E402
from oracle import cases, solvers  # noqa: E402
from test_gpu_generic import arm_hand  # noqa: E402

RetargetingConfig.set_default_urdf_dir(str(DEFAULT_URDF_DIR))
kind = sys.argv[1] if len(sys.argv) > 1 else "position"
B = int(sys.argv[2]) if len(sys.argv) > 2 else 256
seq, prob = arm_hand(kind)
m = seq.optimizer.device_model()
kp = bench_data.human_keypoints(B + 1)
mid = np.repeat(seq.joint_limits.mean(1)[None], B, 0).astype(np.float32)
last = m.retarget(np.ascontiguousarray(kp[:-1]), None, mid, keypoints=True)
q, info = m.retarget(np.ascontiguousarray(kp[1:]), None, last, keypoints=True, want_info=True)
ref = np.ascontiguousarray(cases.ref_from_keypoints(prob, kp[1:]), dtype=np.float32)
want, oi = solvers.solve_lm_batched(prob, ref, None, last, newton=True, max_iter=100, return_info=True)
l64 = last.astype(np.float64)
Fg, Fo = prob.total(q.astype(np.float64), ref, None, l64), prob.total(want, ref, None, l64)
it = info["iters"]
print(f"# {kind}: B={B} gpu iters mean {it.mean():.2f} max {it.max()} | oracle iters mean {oi['iters'].mean():.2f} max {oi['iters'].max()}"
      f" | status {np.bincount(info['status'], minlength=3)} | max|dq| {np.abs(q - want).max():.2e}")
for b in np.argsort(-it)[:12]:
    lo, hi = prob.bounds
    nb = int(((q[b] <= lo + 1e-7) | (q[b] >= hi - 1e-7)).sum())
    print(f"  frame {b:4d} (fixture {b % 621:3d})  gpu it {it[b]:3d} status {info['status'][b]}  oracle it {oi['iters'][b]:3d}  F gpu {Fg[b]:.9e} oracle {Fo[b]:.9e}"
          f"  |dq| {np.abs(q[b] - want[b]).max():.2e}  at bounds {nb}  |q - last| {np.abs(q[b] - last[b]).max():.3f}")
