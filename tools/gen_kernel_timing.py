#!/usr/bin/env python3
"""Timings of the general kernel (run on the GPU box): the arm + Shadow hand models only it serves, and -- forced onto
shipped robots -- next to the specialised kernels."""
import os
import sys

import numpy as np
import torch

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
sys.path.insert(0, os.path.join(REPO, "tests"))
import bench_data  # noqa: E402
from dex_retargeting_amd.constants import DEFAULT_URDF_DIR  # noqa: E402
from dex_retargeting_amd.retargeting_config import RetargetingConfig  # noqa: E402
from test_generic_tables import arm_hand_config  # noqa: E402

RetargetingConfig.set_default_urdf_dir(str(DEFAULT_URDF_DIR))
dev = torch.device("cuda:0")


def time_model(seq, B, label):
    opt = seq.optimizer
    m = opt.device_model()
    kp = bench_data.human_keypoints(B + 1)
    mid = np.repeat(seq.joint_limits.mean(1)[None], B, 0).astype(np.float32)
    st = np.zeros(B, np.uint32) if opt.retargeting_type == "DEXPILOT" else None
    last = m.retarget(np.ascontiguousarray(kp[:-1]), None, mid, state=st, keypoints=True)
    t_kp, t_last = torch.from_numpy(np.ascontiguousarray(kp[1:])).to(dev), torch.from_numpy(last).to(dev)
    out = torch.empty_like(t_last)
    it = torch.zeros(B, dtype=torch.int32, device=dev)
    t_st = torch.zeros(B, dtype=torch.int32, device=dev)
    s = torch.cuda.current_stream().cuda_stream
    for _ in range(2):
        m.retarget_dev(B, t_kp.data_ptr(), 0, t_last.data_ptr(), t_st.data_ptr() if st is not None else 0, out.data_ptr(),
                       iters_ptr=it.data_ptr(), stream=s, keypoints=True)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(3):
        m.retarget_dev(B, t_kp.data_ptr(), 0, t_last.data_ptr(), t_st.data_ptr() if st is not None else 0, out.data_ptr(),
                       stream=s, keypoints=True)
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 3
    print(f"{label:64s} kernel {m.kernel()}  B={B:6d}  {ms:9.3f} ms  {B / ms / 1e3:8.2f} Mframes/s  iters mean {float(it.float().mean()):.2f} max {int(it.max())}")


print("# general kernel (family 5) timings, tracking workload (keypoints of the human fixture, warm start = previous frame)")
for kind in ("position", "vector"):
    seq = RetargetingConfig.from_dict(arm_hand_config(kind)).build()
    for B in (64, 4096, 65536):
        time_model(seq, B, f"arm + Shadow hand, {kind} ({seq.optimizer.opt_dof} variables)")
for rel in ("teleop/allegro_hand_right.yml", "teleop/shadow_hand_right_dexpilot.yml", "offline/leap_hand_right.yml"):
    for forced in (False, True):
        seq = RetargetingConfig.load_from_file(os.path.join(bench_data.CONFIG_DIR, rel)).build()
        seq.optimizer.use_generic_tables = forced
        time_model(seq, 65536, rel + (" [generic tables]" if forced else ""))
