#!/usr/bin/env python3
"""(GPU box) PCIe-inclusive rate of the HOST-pointer entry point: dexr_retarget_kp on host arrays (pack into the pinned
block, one H2D, the solve, one D2H, scatter) for several batch sizes -- the figure DESIGN.md quotes next to the
HBM-resident headline."""
import os
import sys
import time

import numpy as np

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
import bench_data  # noqa: E402
from dex_retargeting_amd.constants import DEFAULT_URDF_DIR  # noqa: E402
from dex_retargeting_amd.retargeting_config import RetargetingConfig  # noqa: E402

RetargetingConfig.set_default_urdf_dir(str(DEFAULT_URDF_DIR))
seq = RetargetingConfig.load_from_file(os.path.join(bench_data.CONFIG_DIR, "teleop/allegro_hand_right.yml")).build()
m = seq.optimizer.device_model()
print("# Allegro vector, dexr_retarget_kp (host pointers): frames, ms per call, frames/s, bytes over PCIe per frame = 252 + 64 in, 64 out (+ 12 diagnostics)")
for B in (1, 64, 1024, 16384, 65536, 262144):
    kp = bench_data.human_keypoints(B + 1)
    mid = np.repeat(seq.joint_limits.mean(1)[None], B, 0).astype(np.float32)
    last = m.retarget(np.ascontiguousarray(kp[:-1]), None, mid, keypoints=True)
    x = np.ascontiguousarray(kp[1:])
    for _ in range(3):
        m.retarget(x, None, last, keypoints=True)
    n = 200 if B <= 1024 else 20
    t0 = time.perf_counter()
    for _ in range(n):
        m.retarget(x, None, last, keypoints=True)
    dt = (time.perf_counter() - t0) / n
    print(f"{B:8d} {dt * 1e3:10.4f} ms  {B / dt:14.0f} frames/s")
