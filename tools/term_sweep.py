#!/usr/bin/env python3
"""Developer tool (GPU): cost/accuracy of the float32 termination rules on a bench workload.

For each setting of the dexr_tuning fields (max_blind / stall_*) solve the full 65 536-frame workload in
float32 and compare with the float64 kernel's answer (dexr_retarget_f64, tight) on every frame.

    python tools/term_sweep.py [config.yml] > gpurun_out/term_sweep.txt
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

rel = sys.argv[1] if len(sys.argv) > 1 else "teleop/allegro_hand_right.yml"
B = int(sys.argv[2]) if len(sys.argv) > 2 else 65536
RetargetingConfig.set_default_urdf_dir(str(DEFAULT_URDF_DIR))
seq = RetargetingConfig.load_from_file(os.path.join(cases.CONFIG_DIR, rel)).build()
model = seq.optimizer.device_model()
prob = cases.problem_from_config(rel)
dexpilot = prob.kind == "dexpilot"
dev = torch.device("cuda:0")
T0 = model.get_tuning()
DEFAULT = {k: getattr(T0, k) for k in ("max_blind", "stall_from", "stall_ratio", "stall_cap", "lam_jump", "lam_fastdec",
                                       "floor_scale", "step_cap", "blind_tol_scale")}  # this model's dexr_tuning defaults

kp = cases.human_keypoints(B + 1, seed=cases.SEED)
mid = np.repeat(prob.joint_limits.mean(1)[None], B, 0).astype(np.float32)
st0 = np.zeros(B, np.uint32) if dexpilot else None
last = model.retarget(np.ascontiguousarray(kp[:-1]), None, mid, state=st0, keypoints=True)
kp_now = np.ascontiguousarray(kp[1:])
ref_now = np.ascontiguousarray(cases.ref_from_keypoints(prob, kp_now), dtype=np.float32)
q64 = model.retarget_f64(ref_now, None, last, state=None if not dexpilot else np.zeros(B, np.uint32))

t_kp, t_last = torch.from_numpy(kp_now).to(dev), torch.from_numpy(last).to(dev)
t_q = torch.empty((B, prob.n_opt), dtype=torch.float32, device=dev)
t_st = torch.zeros(B, dtype=torch.int32, device=dev)
t_it = torch.zeros(B, dtype=torch.int32, device=dev)
s = torch.cuda.current_stream()


def go(diag=False, opts=None):
    if dexpilot:
        t_st.zero_()
    model.retarget_dev(B, t_kp.data_ptr(), 0, t_last.data_ptr(), t_st.data_ptr() if dexpilot else 0, t_q.data_ptr(),
                       iters_ptr=t_it.data_ptr() if diag else 0, stream=s.cuda_stream, keypoints=True, opts=opts)


def measure(env, tol=None):
    from dex_retargeting_amd import _lib

    model.tune(**{**DEFAULT, **env})
    opts = _lib.default_options(tol=tol) if tol else None
    for _ in range(3):
        go(opts=opts)
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(15)]
    for a, b in ev:
        a.record(s)
        go(opts=opts)
        b.record(s)
    torch.cuda.synchronize()
    ms = float(np.median([a.elapsed_time(b) for a, b in ev]))
    go(diag=True, opts=opts)
    torch.cuda.synchronize()
    it = t_it.cpu().numpy()
    dq = np.abs(t_q.cpu().numpy().astype(np.float64) - q64).max(1)
    return ms, it, dq


print(f"# {rel} B={B}: float32 kernel vs float64 kernel on all frames")
print(f"{'setting':44s} {'ms':>8s} {'it mean':>8s} {'tile max':>8s} {'max dq':>10s} {'p99.99':>10s} {'>1e-5':>7s} {'>1e-4':>7s}")
SETTINGS = [("default", {}, None),
            ("round-1 first kernel: jump=0 fastdec=0 floor=7e-15 cap=0 blind=0", dict(lam_jump=0.0, lam_fastdec=0.0, floor_scale=7.1e-15, step_cap=0.0, blind_tol_scale=0.0), None)]
for bt in (0, 2e-6, 2e-5, 1e-4):
    SETTINGS.append((f"blind_tol={bt:g}", dict(blind_tol_scale=bt / 2e-6), None))
for cap in (0, 0.2, 0.5):
    SETTINGS.append((f"step_cap={cap}", dict(step_cap=float(cap)), None))
for jump, dec in ((0.3, 0), (1.0, 0), (1.0, 0.1)):
    SETTINGS.append((f"jump={jump} fastdec={dec}", dict(lam_jump=float(jump), lam_fastdec=float(dec)), None))
if "--precision" in sys.argv:  # which knob limits the float32 answer's distance from the float64 one?
    SETTINGS = [("default", {}, None), ("floor=7e-15", dict(floor_scale=7.1e-15), None), ("tol=5e-7", {}, 5e-7),
                ("tol=5e-7 floor=7e-15", dict(floor_scale=7.1e-15), 5e-7), ("max_blind=0 blind_tol=0", dict(max_blind=1000, blind_tol_scale=0.0, stall_from=1000), None),
                ("tol=5e-7 no blind exits", dict(max_blind=1000, blind_tol_scale=0.0, stall_from=1000, floor_scale=7.1e-15), 5e-7)]
for name, env, tol in SETTINGS:
    ms, it, dq = measure(env, tol)
    wm = it[: B // 64 * 64].reshape(-1, 64).max(1)
    print(f"{name:44s} {ms:8.4f} {it.mean():8.2f} {wm.mean():8.2f} {dq.max():10.2e} {np.percentile(dq, 99.99):10.2e} "
          f"{int((dq > 1e-5).sum()):7d} {int((dq > 1e-4).sum()):7d}")
    if name == "default":
        print("   iters histogram:", np.bincount(it).tolist())
