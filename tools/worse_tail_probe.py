#!/usr/bin/env python3
"""Developer tool (GPU, diagnostic library): could the frames that end in a WORSE local minimum than the float64 oracle's be
caught by "one extra solve for frames that saw a rejected step or a trust-radius cut" ?

For every config with such frames in the all-configs parity run (tools/data/far_frames_r05.json: indices of the frames >= 1e-4
rad from the oracle in the 4 096-frame workload of tests/test_gpu_all_configs.py, and which of them have the higher F) the
diagnostic build of the sixteen-lane kernel (-DDEXR_WIDE_DIAG=1, tools/pass_composition.sh build) reports per frame the passes
that were rejected steps / steps cut by the trust radius / failed factorisations.  Printed per config: the share of ALL frames
that satisfy each predicate, how many of the worse-minimum frames it covers, and what a second solve of the flagged frames
would cost in passes (their passes / all passes).

    bash tools/pass_composition.sh build        (here)
    DEXR_LIB=tools/_prof/libdexr_wdiag.so python tools/worse_tail_probe.py      (GPU box)
"""
import json
import os
import sys

import numpy as np

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
from dex_retargeting_amd import _lib  # noqa: E402
from dex_retargeting_amd.constants import DEFAULT_URDF_DIR  # noqa: E402
from dex_retargeting_amd.retargeting_config import RetargetingConfig  # noqa: E402
from oracle import cases  # noqa: E402  (input recipes only)

RetargetingConfig.set_default_urdf_dir(str(DEFAULT_URDF_DIR))
far = json.load(open(os.path.join(REPO, "tools", "data", "far_frames_r05.json")))
B = 4096
tot_w = tot_cov = {}
print(f"# {B} tracking frames per config (the workload of tests/test_gpu_all_configs.py), diagnostic library {os.environ.get('DEXR_LIB')}")
print(f"{'config':44s} {'kernel':>11s} {'worse':>5s} | predicate: share of all frames, worse frames covered, passes of a second solve / all passes")
agg = {}
for rel, rec in far.items():
    nworse = int(sum(rec["worse"]))
    if nworse == 0:
        continue
    seq = RetargetingConfig.load_from_file(os.path.join(cases.CONFIG_DIR, rel)).build()
    prob = cases.problem_from_config(rel)
    model = seq.optimizer.device_model()
    if model.kernel()[0] != _lib.KERNEL_WIDE:
        print(f"{rel:44s} {str(model.kernel()):>11s} {nworse:5d} | (not on the sixteen-lane kernel: no per-pass diagnostics)")
        continue
    dex = prob.kind == "dexpilot"
    kp = cases.human_keypoints(B + 1, seed=cases.SEED)
    mid = np.repeat(prob.joint_limits.mean(1)[None], B, 0).astype(np.float32)
    st = np.zeros(B, np.uint32) if dex else None
    last = model.retarget(np.ascontiguousarray(kp[:-1]), None, mid, state=st, keypoints=True)
    q, info = model.retarget(np.ascontiguousarray(kp[1:]), None, last, state=st, keypoints=True, want_info=True)
    v = info["iters"].astype(np.int64)
    it, rej, cap, fail = v & 255, (v >> 8) & 255, (v >> 16) & 255, (v >> 24) & 255
    widx = np.array([i for i, w in zip(rec["far_idx"], rec["worse"]) if w])
    line = f"{rel:44s} {str(model.kernel()):>11s} {nworse:5d} |"
    for name, m in (("rejected step", rej > 0), ("trust-radius cut", cap > 0), ("either", (rej > 0) | (cap > 0)),
                    (">= 2 rejections", rej >= 2)):
        cov = int(m[widx].sum())
        line += f" {name}: {m.mean():.3f}, {cov}/{nworse}, {it[m].sum() / it.sum():.3f};"
        a = agg.setdefault(name, [0, 0, 0, 0])
        a[0] += cov; a[1] += nworse; a[2] += int(it[m].sum()); a[3] += int(it.sum())
    print(line)
    print(f"{'':62s} passes of the worse frames: {it[widx].tolist()}  rejected {rej[widx].tolist()}  cut {cap[widx].tolist()}")
print("# all configs above:")
for name, a in agg.items():
    print(f"#   {name:18s} covers {a[0]} of {a[1]} worse-minimum frames; a second solve of the flagged frames = {a[2] / a[3]:.3f} of all passes")
