#!/usr/bin/env python3
"""GPU: every shipped config on the bench workload (65 536 human-keypoint frames, warm start = previous frame):
kernel used, ms per launch, frames/s, iteration statistics, and the float32 answer's distance from the float64 kernel's.

    python tools/all_configs.py > gpurun_out/all_configs.txt
"""
import glob
import os
import sys

import numpy as np

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
sys.path.insert(0, os.path.join(REPO, "tools"))
import torch  # noqa: E402

from dex_retargeting_amd.constants import DEFAULT_URDF_DIR  # noqa: E402
from dex_retargeting_amd.retargeting_config import RetargetingConfig  # noqa: E402
from oracle import cases  # noqa: E402  (input recipes only)

RetargetingConfig.set_default_urdf_dir(str(DEFAULT_URDF_DIR))
B = int(sys.argv[1]) if len(sys.argv) > 1 else 65536
ONLY = sys.argv[2].split(",") if len(sys.argv) > 2 and sys.argv[2] != "all" else None  # substrings of config paths to run
KERNEL = sys.argv[3] if len(sys.argv) > 3 else "auto"  # force a kernel family (auto|register|quad|lds|reduced|wide) where it applies
dev = torch.device("cuda:0")
kp = cases.human_keypoints(B + 1, seed=cases.SEED)
t_kp = torch.from_numpy(np.ascontiguousarray(kp[1:])).to(dev)
s = torch.cuda.current_stream()
print(f"# {B} frames per launch, one MI355X; dq = max_j |q_f32 - q_f64| per frame over all frames")
print(f"# kernel = {KERNEL}; kernel column: (family 0 register / 1 quad / 2 LDS / 3 reduced / 4 sixteen-lane, joint bucket, chain)")
print(f"{'config':44s} {'kernel':>14s} {'n_opt':>5s} {'comps':>5s} {'ms':>8s} {'Mframes/s':>9s} {'it mean':>7s} {'it max':>6s} {'conv':>6s} {'p99.9 dq':>9s} {'>1e-4':>6s} {'it p99':>6s} {'tile max':>8s}")
for path in sorted(glob.glob(os.path.join(cases.CONFIG_DIR, "*", "*.yml"))):
    rel = os.path.relpath(path, cases.CONFIG_DIR)
    if ONLY is not None and not any(o in rel for o in ONLY):
        continue
    prob = cases.problem_from_config(rel)
    seq = RetargetingConfig.load_from_file(path).build()
    model = seq.optimizer.device_model()
    if KERNEL != "auto":
        from dex_retargeting_amd import _lib
        model.tune(kernel={"register": _lib.KERNEL_REGISTER, "quad": _lib.KERNEL_QUAD, "lds": _lib.KERNEL_LDS,
                           "reduced": _lib.KERNEL_REDUCED, "wide": _lib.KERNEL_WIDE}[KERNEL])
    if os.environ.get("DEXR_TOOL_KNOBS"):  # e.g. DEXR_TOOL_KNOBS="blind_tol_scale=100,step_cap=0.2": developer knobs (tools/_tune.py)
        import _tune
        _, OPTS = _tune.apply(model, dict(kv.split("=") for kv in os.environ["DEXR_TOOL_KNOBS"].split(",")))
    else:
        OPTS = None
    dex = prob.kind == "dexpilot"
    mid = np.repeat(prob.joint_limits.mean(1)[None], B, 0).astype(np.float32)
    st = (lambda: np.zeros(B, np.uint32)) if dex else (lambda: None)
    last = model.retarget(np.ascontiguousarray(kp[:-1]), None, mid, state=st(), keypoints=True)
    ref = np.ascontiguousarray(cases.ref_from_keypoints(prob, kp[1:]), dtype=np.float32)
    q64 = model.retarget_f64(ref, None, last, state=st())
    t_last = torch.from_numpy(last).to(dev)
    t_q = torch.empty((B, prob.n_opt), dtype=torch.float32, device=dev)
    t_st = torch.zeros(B, dtype=torch.int32, device=dev)
    t_it = torch.zeros(B, dtype=torch.int32, device=dev)
    t_status = torch.zeros(B, dtype=torch.int32, device=dev)

    def go(diag=False):
        if dex:
            t_st.zero_()
        model.retarget_dev(B, t_kp.data_ptr(), 0, t_last.data_ptr(), t_st.data_ptr() if dex else 0, t_q.data_ptr(),
                           status_ptr=t_status.data_ptr() if diag else 0, iters_ptr=t_it.data_ptr() if diag else 0,
                           stream=s.cuda_stream, keypoints=True, opts=OPTS)

    for _ in range(2):
        go()
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(7)]
    for a, b in ev:
        a.record(s)
        go()
        b.record(s)
    torch.cuda.synchronize()
    ms = float(np.median([a.elapsed_time(b) for a, b in ev]))
    go(diag=True)
    torch.cuda.synchronize()
    it = t_it.cpu().numpy()
    dq = np.abs(t_q.cpu().numpy().astype(np.float64) - q64).max(1)
    ncomp = len(seq.optimizer.compiled_model().comps)
    print(f"{rel:44s} {str(model.kernel()):>14s} {prob.n_opt:5d} {ncomp:5d} {ms:8.3f} {B / ms / 1e3:9.2f} {it.mean():7.2f} {it.max():6d} "
          f"{float((t_status == 0).float().mean()):6.4f} {np.percentile(dq, 99.9):9.1e} {int((dq > 1e-4).sum()):6d} {np.percentile(it, 99):6.0f} "
          f"{it[: B // 64 * 64].reshape(-1, 64).max(1).mean():8.2f}", flush=True)  # hist99
