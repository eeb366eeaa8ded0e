#!/usr/bin/env python3
"""Developer tool (GPU): where the quad kernel and the register kernel disagree on a DexPilot model, compare objectives."""
import os
import sys

import numpy as np

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
import torch  # noqa: E402,F401

from dex_retargeting_amd import _lib  # noqa: E402
from dex_retargeting_amd.constants import DEFAULT_URDF_DIR  # noqa: E402
from dex_retargeting_amd.retargeting_config import RetargetingConfig  # noqa: E402
from oracle import cases  # noqa: E402

rel = sys.argv[1] if len(sys.argv) > 1 else "teleop/leap_hand_right_dexpilot.yml"
B = 16384
RetargetingConfig.set_default_urdf_dir(str(DEFAULT_URDF_DIR))
prob = cases.problem_from_config(rel)
kp = cases.human_keypoints(B + 1, seed=cases.SEED)
ref = np.ascontiguousarray(cases.ref_from_keypoints(prob, kp), dtype=np.float32)
mid = np.repeat(prob.joint_limits.mean(1)[None], B, 0).astype(np.float32)


def model_for(kernel):
    m = RetargetingConfig.load_from_file(os.path.join(cases.CONFIG_DIR, rel)).build().optimizer.device_model()
    m.tune(kernel=kernel)  # dexr_tuning.kernel: per-handle kernel family
    return m


m_reg, m_quad = model_for(_lib.KERNEL_REGISTER), model_for(_lib.KERNEL_QUAD)
st0 = np.zeros(B, np.uint32)
last = m_reg.retarget(ref[:-1], None, mid, state=st0)
s1, s2 = st0.copy(), st0.copy()
q_reg = m_reg.retarget(ref[1:], None, last, state=s1).astype(np.float64)
q_quad = m_quad.retarget(ref[1:], None, last, state=s2).astype(np.float64)
proj = ((st0[:, None] >> np.arange(prob.n_pair, dtype=np.uint32)) & 1).astype(bool)
w, rv, _ = prob.dexpilot_preamble(ref[1:], proj)
kw = dict(weights=w, dexpilot_ref=rv)
last64 = last.astype(np.float64)
dq = np.abs(q_reg - q_quad).max(1)
off = np.nonzero(dq > 1e-4)[0]
if not len(off):
    print(f"# {rel}: no frame of {B} differs by > 1e-4 rad between the register (+polish) and the quad kernel")
    sys.exit(0)
F_reg = prob.total(q_reg[off], ref[1:][off], None, last64[off], **{k: v[off] for k, v in kw.items()})
F_quad = prob.total(q_quad[off], ref[1:][off], None, last64[off], **{k: v[off] for k, v in kw.items()})
print(f"# {rel}: {len(off)} of {B} frames differ by > 1e-4 rad between the register (+polish) and the quad kernel")
if len(off):
    d = F_quad - F_reg
    print(f"F_quad - F_reg: median {np.median(d):+.3e}, quad lower in {int((d < -1e-9).sum())}, equal in {int((np.abs(d) <= 1e-9).sum())}, "
          f"register lower in {int((d > 1e-9).sum())}; mean {d.mean():+.3e}; typical F {np.median(F_reg):.3e}")
