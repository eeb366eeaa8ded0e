#!/usr/bin/env python3
"""GPU probe: re-solve the frames a parity table flagged (gpurun_out/all_configs_far_frames*.npz: inputs + both answers) one at
a time under variants of the small-batch launch policy, and print where each variant ends.

    python tools/ladder_probe.py gpurun_out/all_configs_far_frames_b2048.npz [config-substring ...]
"""
import os
import sys

import numpy as np

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
from dex_retargeting_amd.constants import DEFAULT_URDF_DIR  # noqa: E402
from dex_retargeting_amd.retargeting_config import RetargetingConfig  # noqa: E402
from oracle import cases, solvers  # noqa: E402
from oracle.jobs import _kw  # noqa: E402

RetargetingConfig.set_default_urdf_dir(str(DEFAULT_URDF_DIR))
VARIANTS = [("ladder (default)", {}), ("copies", dict(sprint_ladder=0)), ("four per wave", dict(sprint_max_batch=0)),
            ("ladder, every step verified", dict(blind_tol_scale=0.0)), ("ladder, max_blind 64", dict(max_blind=64)),
            ("ladder, stall_from 99", dict(stall_from=99))]


def main():
    d = np.load(sys.argv[1])
    want = sys.argv[2:]
    keys = sorted({k.rsplit("__", 1)[0] for k in d.files})
    for k in keys:
        if want and not any(w in k for w in want):
            continue
        rel = k.replace("__", "/") + ".yml"
        seq = RetargetingConfig.load_from_file(os.path.join(cases.CONFIG_DIR, rel)).build()
        prob = cases.problem_from_config(rel)
        model = seq.optimizer.device_model()
        base = model.get_tuning()
        ref, last, qo = d[k + "__ref"], d[k + "__last"], d[k + "__q_oracle"]
        st_in = d[k + "__state_in"] if k + "__state_in" in d.files else None
        kw = _kw(prob, ref, st_in)
        tight = solvers.solve_tight(prob, ref, None, last, x0=d[k + "__q_gpu"].astype(np.float64), **kw)
        last64 = last.astype(np.float64)
        print(f"== {rel} kernel {model.kernel()}  {len(ref)} frames")
        for i in range(len(ref)):
            print(f" frame {int(d[k + '__idx'][i])}: recorded |q_gpu - q_oracle| {np.abs(d[k + '__q_gpu'][i] - qo[i]).max():.2e}  tight-from-gpu moved "
                  f"{np.abs(tight[i] - d[k + '__q_gpu'][i]).max():.2e}")
            for name, tk in VARIANTS:
                model.tune(**{f: getattr(base, f) for f in ("sprint_ladder", "sprint_max_batch", "blind_tol_scale", "max_blind", "stall_from")})
                model.tune(**tk)
                st = None if st_in is None else st_in[i:i + 1].copy()
                q, info = model.retarget(ref[i:i + 1], None, last[i:i + 1], state=st, want_info=True)
                q = q.astype(np.float64)
                kwi = {a: b[i:i + 1] for a, b in kw.items()}
                F = prob.total(q, ref[i:i + 1], None, last64[i:i + 1], **kwi)[0]
                Fo = prob.total(qo[i:i + 1], ref[i:i + 1], None, last64[i:i + 1], **kwi)[0]
                Ft = prob.total(tight[i:i + 1], ref[i:i + 1], None, last64[i:i + 1], **kwi)[0]
                print(f"   {name:30s} iters {int(info['iters'][0]):3d} status {int(info['status'][0])}  |q - q_oracle| {np.abs(q[0] - qo[i]).max():.2e}  "
                      f"|q - tight| {np.abs(q[0] - tight[i]).max():.2e}  F - F_oracle {F - Fo:+.2e}  F - F_tight {F - Ft:+.2e}")
        model.tune(**{f: getattr(base, f) for f in ("sprint_ladder", "sprint_max_batch", "blind_tol_scale", "max_blind", "stall_from")})


if __name__ == "__main__":
    main()
