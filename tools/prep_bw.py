#!/usr/bin/env python3
"""GPU: achieved HBM bandwidth of the keypoint pre-processing kernel (dexr_mano_keypoints_dev): 252 B in + 252 B out
(+ 36 B with the wrist rotation) per frame."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from dex_retargeting_amd import keypoints as kpmod  # noqa: E402

dev = torch.device("cuda:0")
s = torch.cuda.current_stream()
print("# dexr_mano_keypoints_dev, median of 20 launches")
for B in (65536, 1 << 20, 1 << 22, 1 << 23):
    raw = torch.randn((B, 21, 3), dtype=torch.float32, device=dev)
    out = torch.empty_like(raw)
    rot = torch.empty((B, 3, 3), dtype=torch.float32, device=dev)
    for with_rot in (False, True):
        def go():
            kpmod.mano_keypoints_dev(B, raw.data_ptr(), out.data_ptr(), rot.data_ptr() if with_rot else 0, "Right", s.cuda_stream)
        for _ in range(3):
            go()
        ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(20)]
        for a, b in ev:
            a.record(s)
            go()
            b.record(s)
        torch.cuda.synchronize()
        ms = float(np.median([a.elapsed_time(b) for a, b in ev]))
        nbytes = B * (504 + (36 if with_rot else 0))
        print(f"B={B:8d} rot={int(with_rot)}: {ms:8.4f} ms  {nbytes / ms / 1e6:8.1f} GB/s  ({nbytes / ms / 1e6 / 8000 * 100:.1f} % of 8 TB/s)")
