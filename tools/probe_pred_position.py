#!/usr/bin/env python3
"""Developer tool (GPU + oracle on the host): what predicts the slow frames of a POSITION model?  Candidates computed at the
start point from the oracle's kinematics: F(x0), the largest residual, the number of residual coordinates in the linear zone
of the SmoothL1 loss, the gradient norm.   python tools/probe_pred_position.py [config] [B]"""
import os
import sys

import numpy as np

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
from dex_retargeting_amd.constants import DEFAULT_URDF_DIR  # noqa: E402
from dex_retargeting_amd.retargeting_config import RetargetingConfig  # noqa: E402
from oracle import cases  # noqa: E402

RetargetingConfig.set_default_urdf_dir(str(DEFAULT_URDF_DIR))
rel = sys.argv[1] if len(sys.argv) > 1 else "offline/leap_hand_right.yml"
B = int(sys.argv[2]) if len(sys.argv) > 2 else 16384
prob = cases.problem_from_config(rel)
seq = RetargetingConfig.load_from_file(os.path.join(cases.CONFIG_DIR, rel)).build()
model = seq.optimizer.device_model()
kp = cases.human_keypoints(B + 1, seed=cases.SEED)
mid = np.repeat(prob.joint_limits.mean(1)[None], B, 0).astype(np.float32)
last = model.retarget(np.ascontiguousarray(kp[:-1]), None, mid, keypoints=True)
q, info = model.retarget(np.ascontiguousarray(kp[1:]), None, last, keypoints=True, want_info=True)
it = info["iters"]
ref = np.ascontiguousarray(cases.ref_from_keypoints(prob, kp[1:]), dtype=np.float32)
f0, g0, pos = prob.evaluate(last.astype(np.float64), ref, None, last)
res = pos[:, prob.target_idx] - ref.astype(np.float64) if hasattr(prob, "target_idx") else None
if res is None:  # the computed links are the target links, in order
    res = pos[:, : ref.shape[1]] - ref.astype(np.float64)
ares = np.abs(res).reshape(B, -1)
lo, hi = prob.joint_limits[:, 0], prob.joint_limits[:, 1]
at_bound = ((last <= lo + 2e-3) | (last >= hi - 2e-3)).sum(1).astype(float)
tips = kp[1:, [4, 8, 12, 16, 20]]
pair_min = np.min([np.linalg.norm(tips[:, i] - tips[:, j], axis=1) for i in range(5) for j in range(i)], axis=0)
cands = {"joints of last_qpos at a limit (free)": at_bound, "-(min fingertip distance) (free)": -pair_min,
         "hand openness: sum |tip| (free)": np.linalg.norm(tips, axis=2).sum(1), "-(hand openness) (free)": -np.linalg.norm(tips, axis=2).sum(1),
         "|last - mid| (free)": np.abs(last - prob.joint_limits.mean(1)).sum(1),
         "F(x0)": f0, "max |residual|": ares.max(1), "coords in the linear zone": (ares > prob.huber_delta).sum(1).astype(float),
         "|grad|": np.linalg.norm(g0, axis=1), "sum |residual|": ares.sum(1),
         "keypoint spread change": np.abs(np.linalg.norm(kp[1:], axis=2) - np.linalg.norm(kp[:-1], axis=2)).max(1)}
print(f"# {rel}: B={B} iters mean {it.mean():.2f} max {it.max()}; it>=10: {(it >= 10).mean():.4f}, it>=14: {(it >= 14).mean():.4f}, it>=18: {(it >= 18).mean():.4f}")
for name, v in cands.items():
    order = np.argsort(-v)
    line = f"  {name:40s}"
    for frac in (0.05, 0.15, 0.30):
        top = np.zeros(B, bool)
        top[order[: int(B * frac)]] = True
        line += f" | top {int(frac * 100):2d} %: " + " ".join(f"{((top & (it >= thr)).sum() / max(1, (it >= thr).sum())):.2f}" for thr in (10, 14, 18))
    print(line + "   (share of it>=10 / 14 / 18 caught)")

# ---- what an ordering would buy: list scheduling of the frames on the kernel's resident rows (B / 8 rows, as 65 536 frames have
# 8 192), a frame costs iters + 1 passes, a launch ends with its last row
import heapq


def makespan(order, rows):
    h = [0] * rows
    heapq.heapify(h)
    end = 0
    for b in order:
        t = heapq.heappop(h) + int(it[b]) + 1
        end = max(end, t)
        heapq.heappush(h, t)
    return end


rows = max(64, B // 8)
base = makespan(np.arange(B), rows)
print(f"# list scheduling on {rows} rows, passes: natural order {base}; lower bound {max(int(np.ceil((it + 1).sum() / rows)), int(it.max()) + 1)}")
for name in ("coords in the linear zone", "|grad|", "sum |residual|", "F(x0)"):
    v = cands[name]
    for frac in (0.05, 0.15):
        first = np.argsort(-v)[: int(B * frac)]
        rest = np.setdiff1d(np.arange(B), first, assume_unique=False)
        print(f"   {name:28s} top {int(frac * 100):2d} % first: {makespan(np.concatenate([np.sort(first), rest]), rows)} passes ({makespan(np.concatenate([np.sort(first), rest]), rows) / base:.3f} x)")
print(f"   perfect (longest first): {makespan(np.argsort(-it), rows)} passes")
