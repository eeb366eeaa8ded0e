#!/usr/bin/env python3
"""Where a mixed-fleet step spends its time (run on the GPU box): the whole call with / without forked per-model streams,
and every model's bucket alone (frames of the other models get id -1 = left untouched)."""
import os
import sys

import numpy as np
import torch

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
import bench_data  # noqa: E402
from bench import FLEET  # noqa: E402
from dex_retargeting_amd.constants import DEFAULT_URDF_DIR  # noqa: E402
from dex_retargeting_amd.fleet import MixedFleet  # noqa: E402
from dex_retargeting_amd.retargeting_config import RetargetingConfig  # noqa: E402

RetargetingConfig.set_default_urdf_dir(str(DEFAULT_URDF_DIR))
dev = torch.device("cuda:0")
B = int(sys.argv[1]) if len(sys.argv) > 1 else 131072
seqs = [RetargetingConfig.load_from_file(os.path.join(bench_data.CONFIG_DIR, r)).build() for r in FLEET]
fleet = MixedFleet([q.optimizer for q in seqs], device=str(dev))
rng = np.random.default_rng(0)
mid = rng.integers(0, len(FLEET), B).astype(np.int32)
kp = bench_data.human_keypoints(B + 1)
start = np.zeros((B, fleet.n_max), np.float32)
for m, sq in enumerate(seqs):
    start[mid == m, : sq.optimizer.opt_dof] = sq.joint_limits.mean(1).astype(np.float32)
t_mid = torch.from_numpy(mid).to(dev)
t_state = torch.zeros(B, dtype=torch.int32, device=dev)
t_last = fleet.retarget(t_mid, torch.from_numpy(np.ascontiguousarray(kp[:-1])).to(dev), torch.from_numpy(start).to(dev), t_state).clone()
t_kp = torch.from_numpy(np.ascontiguousarray(kp[1:])).to(dev)
out = torch.zeros((B, fleet.n_max), dtype=torch.float32, device=dev)


def timed(ids, steps=20):
    for _ in range(3):
        t_state.zero_()
        fleet.retarget(ids, t_kp, t_last, t_state, out=out)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(steps):
        t_state.zero_()
        fleet.retarget(ids, t_kp, t_last, t_state, out=out)
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / steps


for fork in (1, 0, 1, 0):
    fleet.models[0].tune(fork_streams=fork)
    print(f"fork_streams={fork}: {timed(t_mid):.4f} ms per step of {B} frames")
fleet.models[0].tune(fork_streams=1)
for m, rel in enumerate(FLEET):
    ids = torch.where(t_mid == m, t_mid, torch.full_like(t_mid, -1))
    print(f"only {rel}: {timed(ids):.4f} ms ({int((mid == m).sum())} frames)")
ids = torch.full_like(t_mid, -1)
print(f"no frames at all (bucketing + empty launches): {timed(ids):.4f} ms")
