#!/usr/bin/env python3
"""Developer tool (GPU): the one-frame-per-wave launch shape of the sixteen-lane kernel (dexr_tuning.sprint_max_batch) against
the four-frames-per-wave shape on the same frames -- answers, iteration counts, DexPilot state, launch time.

    python tools/sprint_check.py [config.yml ...] [B ...]
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

rels = [a for a in sys.argv[1:] if a.endswith(".yml")] or ["teleop/shadow_hand_right_dexpilot.yml", "offline/leap_hand_right.yml",
                                                             "teleop/shadow_hand_right.yml", "teleop/allegro_hand_right_dexpilot.yml",
                                                             "offline/shadow_hand_right.yml"]
sizes = [int(a) for a in sys.argv[1:] if a.isdigit()] or [1, 5, 64, 700]
RetargetingConfig.set_default_urdf_dir(str(DEFAULT_URDF_DIR))
dev = torch.device("cuda:0")
s = torch.cuda.current_stream()
for rel in rels:
    seq = RetargetingConfig.load_from_file(os.path.join(cases.CONFIG_DIR, rel)).build()
    model = seq.optimizer.device_model()
    prob = cases.problem_from_config(rel)
    dex = prob.kind == "dexpilot"
    for B in sizes:
        kp = cases.human_keypoints(B + 1, seed=cases.SEED)
        mid = np.repeat(prob.joint_limits.mean(1)[None], B, 0).astype(np.float32)
        res = {}
        for mode, smax, lad in (("four per wave", 0, 0), ("one per wave", 1 << 20, 0), ("ladder", 1 << 20, 1)):
            model.tune(sprint_max_batch=smax, sprint_ladder=lad)
            st = np.zeros(B, np.uint32) if dex else None
            last_m = model.retarget(np.ascontiguousarray(kp[:-1]), None, mid, state=st, keypoints=True)  # (cold start from the limit midpoint)
            if mode == "four per wave":
                last, st_first = last_m, (None if st is None else st.copy())
            st = None if st_first is None else st_first.copy()  # every mode tracks from the SAME previous answers / state
            st_in = None if st is None else st.copy()
            q, info = model.retarget(np.ascontiguousarray(kp[1:]), None, last, state=st, keypoints=True, want_info=True)
            t_kp, t_last = torch.from_numpy(np.ascontiguousarray(kp[1:])).to(dev), torch.from_numpy(last).to(dev)
            t_q = torch.empty_like(t_last)
            t_st = torch.zeros(B, dtype=torch.int32, device=dev)
            ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(20)]
            for i in range(23):
                if dex:
                    t_st.copy_(torch.from_numpy(st_in.astype(np.int32)))
                if i >= 3:
                    ev[i - 3][0].record(s)
                model.retarget_dev(B, t_kp.data_ptr(), 0, t_last.data_ptr(), t_st.data_ptr() if dex else 0, t_q.data_ptr(),
                                   stream=s.cuda_stream, keypoints=True)
                if i >= 3:
                    ev[i - 3][1].record(s)
            torch.cuda.synchronize()
            res[mode] = dict(q=q, last=last_m, it=info["iters"], status=info["status"], st=st, us=float(np.median([a.elapsed_time(b) for a, b in ev])) * 1e3)
        model.tune(sprint_max_batch=-1, sprint_ladder=-1)
        a, b, c = res["four per wave"], res["one per wave"], res["ladder"]
        if os.environ.get("DEXR_SPRINT_DUMP") and B >= 256:
            np.savez(os.path.join(REPO, "gpurun_out", f"sprint_dump_{rel.split('/')[-1][:-4]}_{B}.npz"), kp=kp[1:], last=a["last"], kp_prev=kp[:-1], last_ladder=c["last"],
                     q_four=a["q"], q_ladder=c["q"], it_four=a["it"], it_ladder=c["it"], st_in=np.zeros(B, np.uint32) if not dex else st_in)
        dl = np.abs(a["last"] - b["last"]).max()
        dlc = np.abs(a["last"] - c["last"]).max(1)
        dq = np.abs(a["q"] - b["q"]).max(1)
        dqc = np.abs(a["q"] - c["q"]).max(1)
        print(f"{rel:44s} B={B:4d}  {a['us']:7.1f} -> {b['us']:7.1f} -> ladder {c['us']:7.1f} us   iters {a['it'].mean():.2f} / {b['it'].mean():.2f} / {c['it'].mean():.2f} "
              f"(max {a['it'].max()} / {b['it'].max()} / {c['it'].max()}; differ in {int((a['it'] != b['it']).sum())} frames)  "
              f"max |dq| {dq.max():.2e} (> 1e-5: {int((dq > 1e-5).sum())})  ladder: > 1e-4 in {int((dqc > 1e-4).sum())}, p99 {np.percentile(dqc, 99):.1e}  "
              f"cold-start solve max |dq| {dl:.2e} (ladder: > 1e-4 in {int((dlc > 1e-4).sum())})  status ok {bool((b['status'] == 0).all())} {bool((c['status'] == 0).all())}"
              + (f"  state equal {bool(np.array_equal(a['st'], b['st']))} {bool(np.array_equal(a['st'], c['st']))}" if dex else ""))
