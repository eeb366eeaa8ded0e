#!/usr/bin/env python3
"""Developer tool (GPU): launch duration of a small-component model against the batch size -- the floor (a few waves: the
slowest frame's passes at lone-wave latency) and the full launch.  DEXR_LIB selects a library variant,
DEXR_TOOL_KNOBS="persist_from=1,persist_occ=2,qchunk=64" developer knobs (tools/_tune.py), "f64" the float64 launch.

    python tools/small_latency.py [config] [f64] [B ...]
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

args = [a for a in sys.argv[1:] if a != "f64"]
F64 = "f64" in sys.argv[1:]
rel = args[0] if args and not args[0].isdigit() else "teleop/allegro_hand_right.yml"
sizes = [int(a) for a in args if a.isdigit()] or [64, 4096, 16384, 65536, 262144]
RetargetingConfig.set_default_urdf_dir(str(DEFAULT_URDF_DIR))
seq = RetargetingConfig.load_from_file(os.path.join(cases.CONFIG_DIR, rel)).build()
model = seq.optimizer.device_model()
opts = None
if os.environ.get("DEXR_TOOL_KNOBS"):
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    import _tune

    _, opts = _tune.apply(model, dict(kv.split("=") for kv in os.environ["DEXR_TOOL_KNOBS"].split(",")))
if F64:
    from dex_retargeting_amd import _lib

    opts = _lib.default_options(precision=1)
prob = cases.problem_from_config(rel)
dexpilot = prob.kind == "dexpilot"
dev = torch.device("cuda:0")
print(f"# {rel}: kernel {model.kernel()}  lib {os.environ.get('DEXR_LIB', 'default')}  knobs {os.environ.get('DEXR_TOOL_KNOBS', '-')}  {'f64' if F64 else 'f32'}")
for B in sizes:
    kp = cases.human_keypoints(B + 1, seed=cases.SEED)
    mid = np.repeat(prob.joint_limits.mean(1)[None], B, 0).astype(np.float32)
    st0 = np.zeros(B, np.uint32) if dexpilot else None
    last = model.retarget(np.ascontiguousarray(kp[:-1]), None, mid, state=st0, keypoints=True)
    t_kp, t_last = torch.from_numpy(np.ascontiguousarray(kp[1:])).to(dev), torch.from_numpy(last).to(dev)
    t_q = torch.empty((B, prob.n_opt), dtype=torch.float32, device=dev)
    t_st = torch.zeros(B, dtype=torch.int32, device=dev)
    t_it = torch.zeros(B, dtype=torch.int32, device=dev)
    s = torch.cuda.current_stream()

    def go(diag=False):
        if dexpilot:
            t_st.zero_()
        model.retarget_dev(B, t_kp.data_ptr(), 0, t_last.data_ptr(), t_st.data_ptr() if dexpilot else 0, t_q.data_ptr(),
                           iters_ptr=t_it.data_ptr() if diag else 0, stream=s.cuda_stream, keypoints=True, opts=opts)

    for _ in range(3):
        go()
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(30)]
    for a, b in ev:
        a.record(s)
        go()
        b.record(s)
    torch.cuda.synchronize()
    ms = np.array([a.elapsed_time(b) for a, b in ev])
    go(diag=True)
    torch.cuda.synchronize()
    it = t_it.cpu().numpy()
    print(f"B={B:7d}  median {np.median(ms) * 1e3:8.1f} us  min {ms.min() * 1e3:8.1f} us   iters mean {it.mean():.2f} max {it.max()}"
          f"   checksum {float(t_q.double().sum()):.6f}")
