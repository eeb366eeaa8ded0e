#!/usr/bin/env python3
"""Developer probe (GPU): does ONE batch finish sooner as two half-batches on two streams (fork / join with events) than as
one launch?  The dispatch ramp of the third and fourth wave per SIMD (tools/wave_trace.sh) suggests two queues might place
their workgroups in parallel.   python tools/split_launch_probe.py [config] [B]"""
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
B = int(sys.argv[2]) if len(sys.argv) > 2 else 65536
RetargetingConfig.set_default_urdf_dir(str(DEFAULT_URDF_DIR))
seq = RetargetingConfig.load_from_file(os.path.join(cases.CONFIG_DIR, rel)).build()
model = seq.optimizer.device_model()
prob = cases.problem_from_config(rel)
dev = torch.device("cuda:0")
kp = cases.human_keypoints(B + 1, seed=cases.SEED)
mid = np.repeat(prob.joint_limits.mean(1)[None], B, 0).astype(np.float32)
last = model.retarget(np.ascontiguousarray(kp[:-1]), None, mid, keypoints=True)
t_kp, t_last = torch.from_numpy(np.ascontiguousarray(kp[1:])).to(dev), torch.from_numpy(last).to(dev)
t_q = torch.empty((B, prob.n_opt), dtype=torch.float32, device=dev)
s0, s1 = torch.cuda.current_stream(), torch.cuda.Stream(device=dev)
n = prob.n_opt


def whole():
    model.retarget_dev(B, t_kp.data_ptr(), 0, t_last.data_ptr(), 0, t_q.data_ptr(), stream=s0.cuda_stream, keypoints=True)


def split(parts):
    fork = torch.cuda.Event()
    fork.record(s0)
    s1.wait_event(fork)
    step = B // parts
    for p in range(parts):
        st = s0 if p % 2 == 0 else s1
        lo = p * step
        model.retarget_dev(step, t_kp.data_ptr() + lo * 21 * 3 * 4, 0, t_last.data_ptr() + lo * n * 4, 0, t_q.data_ptr() + lo * n * 4,
                           stream=st.cuda_stream, keypoints=True)
    join = torch.cuda.Event()
    join.record(s1)
    s0.wait_event(join)


def timeit(fn, reps=40):
    for _ in range(5):
        fn()
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(reps)]
    for a, b in ev:
        a.record(s0)
        fn()
        b.record(s0)
    torch.cuda.synchronize()
    ms = np.array([a.elapsed_time(b) for a, b in ev])
    return np.median(ms) * 1e3, ms.min() * 1e3


whole()
torch.cuda.synchronize()
ref = t_q.clone()
print(f"# {rel} B={B} kernel {model.kernel()}")
print("one launch               median %.1f us  min %.1f us" % timeit(whole))
for parts in (2, 4):
    r = timeit(lambda: split(parts))
    torch.cuda.synchronize()
    print(f"{parts} launches on 2 streams   median %.1f us  min %.1f us   equal answers: {bool(torch.equal(ref, t_q))}" % r)
