#!/usr/bin/env python3
"""Developer tool (GPU): the tail launch of large batches (dexr_tuning.tail_passes) against the single launch -- time, iteration
counts, answers, for pass caps P.

    python tools/tail_check.py [config.yml ...] [B]
"""
import os
import sys

import numpy as np

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
import torch  # noqa: E402

from dex_retargeting_amd.constants import DEFAULT_URDF_DIR  # noqa: E402
from dex_retargeting_amd.retargeting_config import RetargetingConfig  # noqa: E402
from oracle import cases  # noqa: E402  (input recipes only)

rels = [a for a in sys.argv[1:] if a.endswith(".yml")] or ["offline/leap_hand_right.yml", "teleop/shadow_hand_right_dexpilot.yml"]
B = ([int(a) for a in sys.argv[1:] if a.isdigit()] or [65536])[0]
RetargetingConfig.set_default_urdf_dir(str(DEFAULT_URDF_DIR))
dev = torch.device("cuda:0")
s = torch.cuda.current_stream()
for rel in rels:
    seq = RetargetingConfig.load_from_file(os.path.join(cases.CONFIG_DIR, rel)).build()
    model = seq.optimizer.device_model()
    prob = cases.problem_from_config(rel)
    dex = prob.kind == "dexpilot"
    kp = cases.human_keypoints(B + 1, seed=cases.SEED)
    mid = np.repeat(prob.joint_limits.mean(1)[None], B, 0).astype(np.float32)
    model.tune(tail_passes=0)
    st0 = np.zeros(B, np.uint32) if dex else None
    last = model.retarget(np.ascontiguousarray(kp[:-1]), None, mid, state=st0, keypoints=True)
    t_kp, t_last = torch.from_numpy(np.ascontiguousarray(kp[1:])).to(dev), torch.from_numpy(last).to(dev)
    t_q = torch.empty_like(t_last)
    t_st0 = torch.from_numpy(st0.astype(np.int32)).to(dev) if dex else None
    t_st = t_st0.clone() if dex else None
    t_it = torch.zeros(B, dtype=torch.int32, device=dev)
    t_status = torch.zeros(B, dtype=torch.int32, device=dev)
    model.reserve(B)
    base = None
    for label, P, lf in (("single launch (policy ordering)", 0, -1), ("single launch, natural order", 0, 0), ("tail P=6", 6, -1), ("tail P=8", 8, -1), ("tail P=10", 10, -1),
                         ("tail P=12", 12, -1), ("tail P=16", 16, -1)):
        model.tune(tail_passes=P, longest_first=lf)
        ts = []
        for i in range(8):
            if dex:
                t_st.copy_(t_st0)
            t_it.zero_(); t_status.zero_()
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record(s)
            model.retarget_dev(B, t_kp.data_ptr(), 0, t_last.data_ptr(), t_st.data_ptr() if dex else 0, t_q.data_ptr(), status_ptr=t_status.data_ptr(),
                               iters_ptr=t_it.data_ptr(), stream=s.cuda_stream, keypoints=True)
            b.record(s)
            torch.cuda.synchronize()
            if i >= 2:
                ts.append(a.elapsed_time(b))
        q, it, stt = t_q.cpu().numpy(), t_it.cpu().numpy(), t_status.cpu().numpy()
        if base is None:
            base = q
        dq = np.abs(q - base).max(1)
        print(f"{rel:42s} B={B} {label:32s} {np.median(ts):7.4f} ms  iters mean {it.mean():.2f} max {it.max()}  handed over {(it > P).mean() * 100 if P else 0:.2f} %  "
              f"status != 0: {int((stt != 0).sum())}  vs single launch: > 1e-4 in {int((dq > 1e-4).sum())}, p99.9 {np.percentile(dq, 99.9):.1e}")
    model.tune(tail_passes=-1, longest_first=-1)
