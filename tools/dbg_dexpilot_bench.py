#!/usr/bin/env python3
"""Developer tool (GPU): the Shadow DexPilot bench frames whose GPU answer differs from the oracle LM's."""
import os
import sys

import numpy as np

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
import torch  # noqa: E402,F401

from dex_retargeting_amd.constants import DEFAULT_URDF_DIR  # noqa: E402
from dex_retargeting_amd.retargeting_config import RetargetingConfig  # noqa: E402
from oracle import cases, solvers  # noqa: E402

rel = "teleop/shadow_hand_right_dexpilot.yml"
RetargetingConfig.set_default_urdf_dir(str(DEFAULT_URDF_DIR))
prob = cases.problem_from_config(rel)
model = RetargetingConfig.load_from_file(os.path.join(cases.CONFIG_DIR, rel)).build().optimizer.device_model()
B = 512
kp = cases.human_keypoints(B + 1, seed=cases.SEED)
ref_all = cases.ref_from_keypoints(prob, kp).astype(np.float32)
mid = np.repeat(prob.joint_limits.mean(1)[None], B, 0).astype(np.float32)
last = model.retarget(np.ascontiguousarray(kp[:-1]), None, mid, state=np.zeros(B, np.uint32), keypoints=True)
ref = ref_all[1:]
st = np.zeros(B, np.uint32)
q, info = model.retarget(np.ascontiguousarray(kp[1:]), None, last, state=st, keypoints=True, want_info=True)
q64 = model.retarget_f64(ref, None, last, state=np.zeros(B, np.uint32))
w, rv, st_o = prob.dexpilot_preamble(ref, np.zeros((B, prob.n_pair), bool))
bits = (st_o.astype(np.uint32) << np.arange(st_o.shape[1], dtype=np.uint32)).sum(1).astype(np.uint32)
print("state bits equal to the oracle preamble:", np.array_equal(bits, st), "mismatches", int((bits != st).sum()))
kw = dict(weights=w, dexpilot_ref=rv)
want, oi = solvers.solve_lm_batched(prob, ref, None, last, newton=True, max_iter=100, return_info=True, **kw)
last64 = last.astype(np.float64)
Fg = prob.total(q.astype(np.float64), ref, None, last64, **kw)
F64 = prob.total(q64, ref, None, last64, **kw)
Fo = prob.total(want, ref, None, last64, **kw)
dq = np.abs(q - want).max(1)
off = np.nonzero(dq >= 1e-4)[0]
print("frames off:", len(off))
# does a tight scipy solve from the GPU answer stay there (certified local minimum)?  and from the oracle answer?
for b in off[:12]:
    kb = {k: v[b:b + 1] for k, v in kw.items()}
    pol_g = solvers.solve_tight(prob, ref[b:b + 1], None, last[b:b + 1], x0=q[b:b + 1].astype(np.float64), **kb)
    Fpg = prob.total(pol_g, ref[b:b + 1], None, last64[b:b + 1], **kb)[0]
    print(f"frame {b}: dq {dq[b]:.3f} F_gpu32 {Fg[b]:.6f} F_gpu64 {F64[b]:.6f} F_oracle {Fo[b]:.6f} | polish from gpu: moved {np.abs(pol_g - q[b]).max():.2e} F {Fpg:.6f}"
          f" | iters gpu {info['iters'][b]} oracle {oi['iters'][b]} | start F {prob.total(last64[b:b+1], ref[b:b+1], None, last64[b:b+1], **kb)[0]:.6f}")
