#!/usr/bin/env python3
"""Developer tool (GPU): time per solver iteration of the small-component kernel at 1..4 waves per SIMD.

Every frame is forced to run exactly max_iter iterations (tol = 0 is unreachable, the blind/stall exits are
disabled), so kernel time = launch overhead + iterations x per-iteration time; the slope over max_iter is the
per-iteration latency.  B = 16384 k Allegro frames = k waves per SIMD (4 components x 256 tiles x k = 1024 k waves).
"""
import os
import sys

import numpy as np

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
import torch  # noqa: E402

from dex_retargeting_amd import _lib  # noqa: E402
from dex_retargeting_amd.constants import DEFAULT_URDF_DIR  # noqa: E402
from dex_retargeting_amd.retargeting_config import RetargetingConfig  # noqa: E402
from oracle import cases  # noqa: E402  (input recipes only)

rel = sys.argv[1] if len(sys.argv) > 1 else "teleop/allegro_hand_right.yml"
RetargetingConfig.set_default_urdf_dir(str(DEFAULT_URDF_DIR))
seq = RetargetingConfig.load_from_file(os.path.join(cases.CONFIG_DIR, rel)).build()
model = seq.optimizer.device_model()
prob = cases.problem_from_config(rel)
dev = torch.device("cuda:0")
model.tune(persist_from=1000000, max_blind=1000000, stall_from=1000000, blind_tol_scale=0.0)  # tile mode, no early exits
s = torch.cuda.current_stream()
print(f"# {rel}: forced iteration counts, tile mode; ms (median of 15)")
for B in (1024, 4096, 16384, 32768, 49152, 65536):
    kp = cases.human_keypoints(B + 1, seed=cases.SEED)
    mid = np.repeat(prob.joint_limits.mean(1)[None], B, 0).astype(np.float32)
    t_kp = torch.from_numpy(np.ascontiguousarray(kp[1:])).to(dev)
    t_last = torch.from_numpy(mid).to(dev)
    t_q = torch.empty((B, prob.n_opt), dtype=torch.float32, device=dev)
    row = []
    for mi in (1, 2, 4, 8, 16, 32):
        opts = _lib.default_options(tol=1e-30, max_iter=mi)
        ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(15)]
        for i in range(18):
            if i >= 3:
                ev[i - 3][0].record(s)
            model.retarget_dev(B, t_kp.data_ptr(), 0, t_last.data_ptr(), 0, t_q.data_ptr(), stream=s.cuda_stream,
                               keypoints=True, opts=opts)
            if i >= 3:
                ev[i - 3][1].record(s)
        torch.cuda.synchronize()
        row.append(float(np.median([a.elapsed_time(b) for a, b in ev])))
    slope = (row[-1] - row[-2]) / 16.0 * 1e3
    print(f"B={B:6d} ({B / 16384:.2f} waves/SIMD): " + "  ".join(f"it{mi}={t:.4f}" for mi, t in zip((1, 2, 4, 8, 16, 32), row)) +
          f"  -> {slope:.2f} us/iteration, intercept {row[-1] - 32 * slope / 1e3:.4f} ms")
