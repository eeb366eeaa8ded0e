#!/usr/bin/env python3
"""GPU: launch one config's solve kernel a few times on the bench workload (for rocprofv3 to wrap).

    python tools/prof_config.py <config.yml> [kernel family: auto|register|quad|lds|reduced] [batch] [launches]
"""
import os
import sys

import numpy as np

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
sys.path.insert(0, os.path.join(REPO, "tools"))
import torch  # noqa: E402

import bench_data  # noqa: E402
from dex_retargeting_amd import _lib  # noqa: E402
from dex_retargeting_amd.constants import DEFAULT_URDF_DIR  # noqa: E402
from dex_retargeting_amd.retargeting_config import RetargetingConfig  # noqa: E402

rel = sys.argv[1]
kernel = sys.argv[2] if len(sys.argv) > 2 else "auto"
B = int(sys.argv[3]) if len(sys.argv) > 3 else 65536
n = int(sys.argv[4]) if len(sys.argv) > 4 else 5
RetargetingConfig.set_default_urdf_dir(str(DEFAULT_URDF_DIR))
seq = RetargetingConfig.load_from_file(os.path.join(bench_data.CONFIG_DIR, rel)).build()
model = seq.optimizer.device_model()
if kernel != "auto":
    model.tune(kernel={"register": _lib.KERNEL_REGISTER, "quad": _lib.KERNEL_QUAD, "lds": _lib.KERNEL_LDS,
                       "reduced": _lib.KERNEL_REDUCED, "wide": _lib.KERNEL_WIDE}[kernel])
if os.environ.get("DEXR_TOOL_KNOBS"):  # e.g. DEXR_TOOL_KNOBS="pivot_rule=1,lam_jump=0.3" (tools only; the library reads no environment)
    import _tune

    _tune.apply(model, dict(kv.split("=") for kv in os.environ["DEXR_TOOL_KNOBS"].split(",")))
dex = seq.optimizer.retargeting_type == "DEXPILOT"
kp = bench_data.human_keypoints(B + 1)
mid = np.repeat(seq.joint_limits.mean(1)[None], B, 0).astype(np.float32)
st = np.zeros(B, np.uint32) if dex else None
last = model.retarget(np.ascontiguousarray(kp[:-1]), None, mid, state=st, keypoints=True)
dev = torch.device("cuda:0")
t_kp = torch.from_numpy(np.ascontiguousarray(kp[1:])).to(dev)
t_last = torch.from_numpy(last).to(dev)
t_q = torch.empty_like(t_last)
t_st0 = torch.from_numpy(st.astype(np.int32)).to(dev) if dex else None
t_st = t_st0.clone() if dex else None
t_it = torch.zeros(B, dtype=torch.int32, device=dev)
s = torch.cuda.current_stream()
ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(n)]
for a, b in ev:
    if dex:
        t_st.copy_(t_st0)
    a.record(s)
    model.retarget_dev(B, t_kp.data_ptr(), 0, t_last.data_ptr(), t_st.data_ptr() if dex else 0, t_q.data_ptr(),
                       iters_ptr=t_it.data_ptr(), stream=s.cuda_stream, keypoints=True)
    b.record(s)
torch.cuda.synchronize()
it = t_it.cpu().numpy()
slow = np.nonzero(it >= max(30, int(np.percentile(it, 99.99))))[0][:32]
if len(slow) and len(sys.argv) > 5:  # dump the slowest frames for off-line analysis
    os.makedirs(os.path.join(REPO, "gpurun_out"), exist_ok=True)
    np.savez(os.path.join(REPO, "gpurun_out", sys.argv[5]), idx=slow, kp=kp[1:][slow], last=last[slow], iters=it[slow],
             state=(st[slow] if dex else np.zeros(len(slow), np.uint32)), q=t_q.cpu().numpy()[slow])
if len(sys.argv) > 6:  # the first N frames with their inputs and iteration counts (tools/lm_lab.py replays them)
    nh = int(sys.argv[6])
    np.savez(os.path.join(REPO, "gpurun_out", "head_" + sys.argv[5]), kp=kp[1:][:nh], last=last[:nh], iters=it[:nh],
             state=(t_st0.cpu().numpy()[:nh].astype(np.uint32) if dex else np.zeros(nh, np.uint32)), q=t_q.cpu().numpy()[:nh])
if os.environ.get("DEXR_TOOL_LPT"):  # what would longest-first ordering buy?  re-run with the frames sorted by the
    # iteration counts just measured (a perfect predictor), or with only the top fraction moved to the front
    frac = float(os.environ["DEXR_TOOL_LPT"])
    order = np.argsort(-it, kind="stable")
    nfront = int(frac * B)
    front = np.sort(order[:nfront])
    rest = np.setdiff1d(np.arange(B), front)
    pm = torch.from_numpy(np.concatenate([front, rest])).to(dev)
    t_kp2, t_last2 = t_kp[pm].contiguous(), t_last[pm].contiguous()
    t_st2 = t_st0[pm].contiguous() if dex else None
    ts = []
    for _ in range(n):
        if dex:
            t_st.copy_(t_st2)
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(s)
        model.retarget_dev(B, t_kp2.data_ptr(), 0, t_last2.data_ptr(), t_st.data_ptr() if dex else 0, t_q.data_ptr(),
                           stream=s.cuda_stream, keypoints=True)
        b.record(s)
        torch.cuda.synchronize()
        ts.append(a.elapsed_time(b))
    print(f"  hardest {frac:.0%} of the frames first: ms {np.median(ts):.3f}")
print(f"{rel} kernel={model.kernel()} B={B}: ms {np.median([a.elapsed_time(b) for a, b in ev]):.3f}; iters mean {it.mean():.2f} "
      f"p99 {np.percentile(it, 99):.0f} max {it.max()}; hist {np.bincount(it).tolist()}")
