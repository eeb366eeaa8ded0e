#!/usr/bin/env python3
"""Developer tool (GPU): Shadow DexPilot tracking set, quad kernel vs register kernel vs oracle, per-frame detail."""
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
B = 512
d = cases.reachable_set(prob, B, 0.05)
w, rv, _ = prob.dexpilot_preamble(d["ref"], np.zeros((B, prob.n_pair), bool))
kw = dict(weights=w, dexpilot_ref=rv)
want, info = solvers.solve_lm_batched(prob, d["ref"], d["fixed"], d["last"], newton=True, max_iter=100, return_info=True, **kw)
last64 = d["last"].astype(np.float64)
Fo = prob.total(want, d["ref"], d["fixed"], last64, **kw)


def model_for(env):
    for k in ("DEXR_NO_QUAD", "DEXR_FORCE_BIG", "DEXR_LAM_JUMP", "DEXR_LAM_FASTDEC", "DEXR_FLOOR", "DEXR_MAX_ITER", "DEXR_STEP_CAP"):
        os.environ.pop(k, None)
    os.environ.update(env)
    return RetargetingConfig.load_from_file(os.path.join(cases.CONFIG_DIR, rel)).build().optimizer.device_model()


OLD = {"DEXR_LAM_JUMP": "0", "DEXR_LAM_FASTDEC": "0", "DEXR_FLOOR": "7.1e-15"}
res = {}
for name, env in (("quad new defaults", {}), ("quad new defaults (again)", {}), ("quad old dynamics", OLD),
                  ("quad old dynamics, max_iter 300", dict(OLD, DEXR_MAX_ITER="300")),
                  ("quad jump=0.3 floor=7e-15", {"DEXR_FLOOR": "7.1e-15"}), ("quad jump=0 floor=1e-12", {"DEXR_LAM_JUMP": "0"}),
                  ("register new defaults", {"DEXR_NO_QUAD": "1"}), ("register old dynamics", dict(OLD, DEXR_NO_QUAD="1")),
                  ("big new defaults", {"DEXR_NO_QUAD": "1", "DEXR_FORCE_BIG": "1"})):
    m = model_for(env)
    q, gi = m.retarget(d["ref"], d["fixed"], d["last"], state=np.zeros(B, np.uint32), want_info=True)
    q = q.astype(np.float64)
    dx = np.abs(q - want).max(1)
    Fg = prob.total(q, d["ref"], d["fixed"], last64, **kw)
    bad = np.nonzero(dx >= 1e-4)[0]
    print(f"## {name}: iters mean {gi['iters'].mean():.2f} max {gi['iters'].max()}  status!=0: {(gi['status'] != 0).sum()}  "
          f"frames off by >1e-4: {len(bad)}  worse than oracle: {int((Fg[bad] > Fo[bad] + 1e-7).sum())}")
    for b in bad:
        print(f"   frame {b}: dx {dx[b]:.3e} F_gpu {Fg[b]:.6e} F_oracle {Fo[b]:.6e} iters {gi['iters'][b]} status {gi['status'][b]} oracle_iters {info['iters'][b]}")
    res[name] = q
print("deterministic:", np.array_equal(res["quad new defaults"], res["quad new defaults (again)"]))
