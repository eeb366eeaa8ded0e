#!/usr/bin/env python3
"""GPU: trajectory of the reduced-variable kernel on a few dumped frames: answer after max_iter = 1, 2, ... iterations
(each from the same start), for the default kernel, the Gauss-Newton model, the float64 register kernel and the
register float32 + polish path.  Output: gpurun_out/red_trace.npz."""
import os
import sys

import numpy as np

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
from dex_retargeting_amd import _lib  # noqa: E402
from dex_retargeting_amd.constants import DEFAULT_URDF_DIR  # noqa: E402
from dex_retargeting_amd.retargeting_config import RetargetingConfig  # noqa: E402

RetargetingConfig.set_default_urdf_dir(str(DEFAULT_URDF_DIR))
rel = "teleop/ability_hand_right_dexpilot.yml"
d = np.load(os.path.join(REPO, "tools", "data", "red_dbg.npz"))
seq = RetargetingConfig.load_from_file(os.path.join(REPO, "dex_retargeting_amd", "configs", rel)).build()
model = seq.optimizer.device_model()
out = {}
for name, tune, base in (("red", dict(kernel=_lib.KERNEL_REDUCED), {}), ("red_gn", dict(kernel=_lib.KERNEL_REDUCED), dict(newton=0)),
                         ("reg", dict(kernel=_lib.KERNEL_REGISTER), dict(polish=0)), ("f64", dict(kernel=_lib.KERNEL_REGISTER), dict(precision=1))):
    model.tune(**tune)
    tr, its, sts = [], [], []
    for k in list(range(1, 13)) + [64]:
        st = d["state"].copy()
        q, info = model.retarget(d["ref"], None, d["last"], state=st, opts=_lib.default_options(max_iter=k, **base), want_info=True)
        tr.append(q)
        its.append(info["iters"])
        sts.append(info["status"])
    out[name] = np.array(tr)
    out[name + "_iters"] = np.array(its)
    out[name + "_status"] = np.array(sts)
    print(name, "final iters", its[-1], "status", sts[-1])
os.makedirs(os.path.join(REPO, "gpurun_out"), exist_ok=True)
np.savez(os.path.join(REPO, "gpurun_out", "red_trace.npz"), **out)
