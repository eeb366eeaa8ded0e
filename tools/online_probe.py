#!/usr/bin/env python3
"""Developer tool (GPU): the one-frame-per-call regime (profile_online_retargeting.py:18-36) taken apart -- per fixture frame
the time of the bare C-ABI host-pointer call, the frame's iteration count, and the time of the same solve through the
device-pointer entry point (HIP events, inputs resident), so that "host path" and "kernel" costs separate.

    python tools/online_probe.py [config.yml ...]
"""
import os
import sys
import time

import numpy as np

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
import torch  # noqa: E402

import bench_data  # noqa: E402
from dex_retargeting_amd.constants import DEFAULT_URDF_DIR  # noqa: E402
from dex_retargeting_amd.retargeting_config import RetargetingConfig  # noqa: E402

rels = [a for a in sys.argv[1:] if a.endswith(".yml")] or ["teleop/shadow_hand_right_dexpilot.yml", "teleop/shadow_hand_right.yml",
                                                             "teleop/allegro_hand_right_dexpilot.yml", "offline/leap_hand_right.yml"]
data = np.load(bench_data.HUMAN_FIXTURE)
RetargetingConfig.set_default_urdf_dir(str(DEFAULT_URDF_DIR))
dev = torch.device("cuda:0")
s = torch.cuda.current_stream()
for rel in rels:
    seq = RetargetingConfig.load_from_file(os.path.join(bench_data.CONFIG_DIR, rel)).build()
    opt = seq.optimizer
    model = opt.device_model()
    idx = opt.target_link_human_indices
    position = opt.retargeting_type == "POSITION"
    dexpilot = opt.retargeting_type == "DEXPILOT"
    lo, hi = seq.joint_limits[:, 0], seq.joint_limits[:, 1]
    last = np.clip(seq.last_qpos, lo, hi).astype(np.float32)[None]
    st = np.zeros(1, np.uint32) if dexpilot else None
    n_fixed = len(getattr(opt, "idx_pin2fixed", []))
    fixed = np.zeros((1, n_fixed), np.float32) if n_fixed else None
    dt, its, refs, lasts, states = [], [], [], [], []
    for jp in data:
        ref = (jp[idx, :] if position else jp[idx[1, :], :] - jp[idx[0, :], :]).astype(np.float32)[None]
        refs.append(ref[0])
        lasts.append(last[0].copy())
        states.append(0 if st is None else int(st[0]))
        tic = time.perf_counter()
        q, info = model.retarget(ref, fixed, last, state=st, want_info=True)
        dt.append(time.perf_counter() - tic)
        its.append(int(info["iters"][0]))
        last = np.clip(q, lo, hi).astype(np.float32)
    dt, its = np.array(dt[5:]) * 1e6, np.array(its[5:])
    A = np.stack([np.ones_like(its, float), its.astype(float)], 1)
    coef = np.linalg.lstsq(A, dt, rcond=None)[0]
    # the same solves through the device-pointer entry point, one launch per frame, HIP events
    t_ref = torch.from_numpy(np.array(refs)).to(dev)
    t_last = torch.from_numpy(np.array(lasts)).to(dev)
    t_st = torch.from_numpy(np.array(states, dtype=np.int32)).to(dev)
    t_fixed = torch.zeros((len(refs), max(n_fixed, 1)), dtype=torch.float32, device=dev)
    t_q = torch.empty_like(t_last)
    n_opt, n_ref = t_last.shape[1], t_ref.shape[1]
    ev = []
    for b in range(len(refs)):
        a, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(s)
        model.retarget_dev(1, t_ref.data_ptr() + b * n_ref * 12, (t_fixed.data_ptr() + b * n_fixed * 4) if n_fixed else 0,
                           t_last.data_ptr() + b * n_opt * 4, (t_st.data_ptr() + b * 4) if dexpilot else 0,
                           t_q.data_ptr() + b * n_opt * 4, stream=s.cuda_stream)
        e.record(s)
        ev.append((a, e))
    torch.cuda.synchronize()
    kdt = np.array([a.elapsed_time(e) for a, e in ev][5:]) * 1e3
    kcoef = np.linalg.lstsq(A, kdt, rcond=None)[0]
    # the same (ref, last, state) pairs as ONE batch: per-frame iteration counts must not depend on the batch size
    qb, infob = model.retarget(np.array(refs), None if fixed is None else np.repeat(fixed, len(refs), 0), np.array(lasts),
                               state=np.array(states, np.uint32) if dexpilot else None, want_info=True)
    itb = infob["iters"][5:]
    print(f"   as one batch of {len(refs)}: iters mean {itb.mean():.2f} max {itb.max()}  frames whose count differs from the one-frame call: "
          f"{int((itb != its).sum())}  hist one-frame {np.bincount(its).tolist()}")
    if os.environ.get("DEXR_PROBE_SAVE"):
        np.savez(os.path.join(REPO, "gpurun_out", "online_pairs_" + rel.split("/")[-1][:-4] + ".npz"), refs=np.array(refs), lasts=np.array(lasts),
                 states=np.array(states, np.uint32), iters=infob["iters"], q=qb)
    print(f"{rel}: kernel {model.kernel()}  host call mean {dt.mean():.1f} us p50 {np.median(dt):.1f} p99 {np.percentile(dt, 99):.1f}  "
          f"iters mean {its.mean():.2f} p99 {np.percentile(its, 99):.0f} max {its.max()}  fit {coef[0]:.1f} + {coef[1]:.2f} x iters | "
          f"device-pointer launch mean {kdt.mean():.1f} us p50 {np.median(kdt):.1f}  fit {kcoef[0]:.1f} + {kcoef[1]:.2f} x iters")
