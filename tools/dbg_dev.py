import sys, os
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from dex_retargeting_amd import _lib
from dex_retargeting_amd.constants import DEFAULT_URDF_DIR
from dex_retargeting_amd.retargeting_config import RetargetingConfig
from oracle import cases
RetargetingConfig.set_default_urdf_dir(str(DEFAULT_URDF_DIR))
for rel in ["offline/leap_hand_right.yml", "teleop/allegro_hand_right.yml", "teleop/shadow_hand_right.yml"]:
    seq = RetargetingConfig.load_from_file(os.path.join(cases.CONFIG_DIR, rel)).build()
    prob = cases.problem_from_config(rel)
    model = seq.optimizer.device_model()
    B = 4096
    d = cases.reachable_set(prob, B, 0.05)
    for polish in (0, -1):
        o = _lib.default_options(polish=polish)
        want = model.retarget(d["ref"], d["fixed"], d["last"], opts=o)
        dev = torch.device("cuda:0")
        ref, last = torch.from_numpy(d["ref"]).to(dev), torch.from_numpy(d["last"]).to(dev)
        out = torch.empty_like(last)
        torch.cuda.synchronize()
        model.retarget_dev(B, ref.data_ptr(), 0, last.data_ptr(), 0, out.data_ptr(), opts=o, stream=torch.cuda.current_stream().cuda_stream)
        torch.cuda.synchronize()
        got = out.cpu().numpy()
        print(rel, "polish", polish, "max diff", np.abs(got - want).max(), "rows differing", (np.abs(got - want).max(1) > 0).sum())
print("got[0]", got[0][:6]); print("want[0]", want[0][:6]); print("last[0]", d["last"][0][:6])
# find whether got rows are a permutation / other rows of want
i = np.argmin(np.abs(want - got[0][None]).max(1)); print("closest want row to got[0]:", i, np.abs(want[i]-got[0]).max())
import ctypes
print("torch hip lib:", [l.split()[-1] for l in open("/proc/self/maps") if "libamdhip64" in l][:1], set(l.split()[-1] for l in open("/proc/self/maps") if "libamdhip64" in l or "libhsa-runtime" in l))
