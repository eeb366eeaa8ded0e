#!/usr/bin/env python3
"""Developer tool (GPU): iteration-count distribution of the headline workload and launch-mode timing sweep.

    python tools/iter_hist.py [workload] > gpurun_out/iter_hist.txt
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
NO_SWEEP = "--no-sweep" in sys.argv
RetargetingConfig.set_default_urdf_dir(str(DEFAULT_URDF_DIR))
seq = RetargetingConfig.load_from_file(os.path.join(cases.CONFIG_DIR, rel)).build()
model = seq.optimizer.device_model()
prob = cases.problem_from_config(rel)
dexpilot = prob.kind == "dexpilot"
dev = torch.device("cuda:0")


def workload(B):
    kp = cases.human_keypoints(B + 1, seed=cases.SEED)
    mid = np.repeat(prob.joint_limits.mean(1)[None], B, 0).astype(np.float32)
    st0 = np.zeros(B, np.uint32) if dexpilot else None
    last = model.retarget(np.ascontiguousarray(kp[:-1]), None, mid, state=st0, keypoints=True)
    return np.ascontiguousarray(kp[1:]), last


def run(B, kp_now, last, reps=20):
    t_kp, t_last = torch.from_numpy(kp_now).to(dev), torch.from_numpy(last).to(dev)
    t_q = torch.empty((B, prob.n_opt), dtype=torch.float32, device=dev)
    t_st = torch.zeros(B, dtype=torch.int32, device=dev)
    t_it = torch.zeros(B, dtype=torch.int32, device=dev)
    s = torch.cuda.current_stream()

    def go(diag=False):
        if dexpilot:
            t_st.zero_()
        model.retarget_dev(B, t_kp.data_ptr(), 0, t_last.data_ptr(), t_st.data_ptr() if dexpilot else 0, t_q.data_ptr(),
                           iters_ptr=t_it.data_ptr() if diag else 0, stream=s.cuda_stream, keypoints=True)

    for _ in range(3):
        go()
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(reps)]
    for a, b in ev:
        a.record(s)
        go()
        b.record(s)
    torch.cuda.synchronize()
    ms = float(np.median([a.elapsed_time(b) for a, b in ev]))
    go(diag=True)
    torch.cuda.synchronize()
    return ms, t_it.cpu().numpy()


B = 65536
kp_now, last = workload(B)
ms, it = run(B, kp_now, last)
print(f"# {rel} B={B}: {ms:.4f} ms, iters mean {it.mean():.2f} max {it.max()}")
print("iters histogram:", np.bincount(it).tolist())
wm = it.reshape(-1, 64).max(1)
print("64-frame tile max histogram:", np.bincount(wm).tolist(), "mean", wm.mean())
slow = np.nonzero(it >= 10)[0]
print(f"frames with >= 10 iterations: {len(slow)}; of those at a fixture wrap (b % 621 == 620): "
      f"{int(((slow % 621) == 620).sum())}")
for thr in (8, 10, 12, 16):
    print(f"  frac of frames with iters > {thr}: {(it > thr).mean():.5f}; tiles containing one: {(wm > thr).mean():.4f}")

if NO_SWEEP:
    sys.exit(0)
print("\n# launch-mode sweep (median ms of 20); persist = resident waves + queue, occN = waves per SIMD, cM = queue chunk")
DEFAULT = dict(persist_from=8, persist_occ=0, qchunk=256)  # dexr_tuning defaults
MODES = [("tile", dict(persist_from=1000000)), ("default", {})]
for occ in (2, 3, 4, 6):
    for ch in (16, 64, 256):
        MODES.append((f"occ{occ}c{ch}", dict(persist_from=0, persist_occ=occ, qchunk=ch)))
for Bs in (4096, 16384, 65536, 131072, 262144, 1048576):
    kpn, la = workload(Bs)
    res = []
    for name, env in MODES:
        model.tune(**{**DEFAULT, **env})
        ms, _ = run(Bs, kpn, la, reps=12)
        res.append((ms, name))
    best = min(res)
    print(f"B={Bs:8d}  " + "  ".join(f"{n} {m:.4f}" for m, n in res) + f"  -> best {best[1]} {Bs / best[0] / 1e3:.0f} kframes/ms")
model.tune(**DEFAULT)
