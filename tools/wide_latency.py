#!/usr/bin/env python3
"""Developer tool (GPU): what a SMALL launch of a large-component model costs -- fixed part vs passes.

For B in (1, 4, 64, 1024) frames of a model served by the sixteen-lane kernel: (a) the natural solve (tracking frames) with
its iteration counts, (b) every frame forced to run exactly max_iter = 1, 2, 4, 8, 16 passes (tol unreachable, blind / stall
exits off): duration = fixed + passes x per-pass, so the slope is the pass latency of a lone wave and the intercept what a
call pays before and after its passes (launch, table set-up, first loads, write-back).  HIP events around the
device-pointer entry point on the current stream; median of 30.

    python tools/wide_latency.py [config.yml ...]
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

rels = [a for a in sys.argv[1:] if a.endswith(".yml")] or ["teleop/shadow_hand_right_dexpilot.yml", "offline/leap_hand_right.yml",
                                                             "teleop/shadow_hand_right.yml", "teleop/allegro_hand_right_dexpilot.yml"]
sizes = [int(a) for a in sys.argv[1:] if a.isdigit()] or [1, 4, 64, 1024]
RetargetingConfig.set_default_urdf_dir(str(DEFAULT_URDF_DIR))
dev = torch.device("cuda:0")
s = torch.cuda.current_stream()


def timed(fn, n=30):
    for _ in range(3):
        fn()
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(n)]
    for a, b in ev:
        a.record(s)
        fn()
        b.record(s)
    torch.cuda.synchronize()
    return float(np.median([a.elapsed_time(b) for a, b in ev])) * 1e3


for rel in rels:
    seq = RetargetingConfig.load_from_file(os.path.join(cases.CONFIG_DIR, rel)).build()
    model = seq.optimizer.device_model()
    prob = cases.problem_from_config(rel)
    dexpilot = prob.kind == "dexpilot"
    print(f"# {rel}: kernel {model.kernel()}  n_opt {prob.n_opt}  lib {os.environ.get('DEXR_LIB', 'default')}")
    for B in sizes:
        kp = cases.human_keypoints(B + 1, seed=cases.SEED)
        mid = np.repeat(prob.joint_limits.mean(1)[None], B, 0).astype(np.float32)
        st0 = np.zeros(B, np.uint32) if dexpilot else None
        last = model.retarget(np.ascontiguousarray(kp[:-1]), None, mid, state=st0, keypoints=True)
        t_kp, t_last = torch.from_numpy(np.ascontiguousarray(kp[1:])).to(dev), torch.from_numpy(last).to(dev)
        t_q = torch.empty((B, prob.n_opt), dtype=torch.float32, device=dev)
        t_st = torch.zeros(B, dtype=torch.int32, device=dev)
        t_it = torch.zeros(B, dtype=torch.int32, device=dev)

        def go(opts=None, diag=False):
            model.retarget_dev(B, t_kp.data_ptr(), 0, t_last.data_ptr(), t_st.data_ptr() if dexpilot else 0, t_q.data_ptr(),
                               iters_ptr=t_it.data_ptr() if diag else 0, stream=s.cuda_stream, keypoints=True, opts=opts)

        model.tune()  # defaults
        nat = timed(lambda: go())
        t_it.zero_()
        go(diag=True)
        torch.cuda.synchronize()
        it = t_it.cpu().numpy()
        saved = model.tune()
        model.tune(max_blind=1000000, stall_from=1000000, blind_tol_scale=0.0)
        row = []
        for mi in (1, 2, 4, 8, 16):
            o = _lib.default_options(tol=1e-30, max_iter=mi)
            row.append(timed(lambda: go(o)))
        model.tune(max_blind=saved.max_blind, stall_from=saved.stall_from, blind_tol_scale=saved.blind_tol_scale)
        slope = (row[-1] - row[-2]) / 8.0
        print(f"B={B:5d}  natural {nat:7.1f} us (iters mean {it.mean():.2f} max {it.max()})   forced " +
              "  ".join(f"it{mi}={t:.1f}" for mi, t in zip((1, 2, 4, 8, 16), row)) +
              f"   -> {slope:.2f} us/pass, fixed {row[-1] - 16 * slope:.1f} us")
