#!/usr/bin/env python3
"""GPU: same-box A/B of two builds of libdexr.so on the bench workload (65 536 human-tracking frames per launch, warm start = the
previous frame's solution): launches alternate between the libraries (one process per measurement, DEXR_LIB selects the library),
HIP events around 20 launches, median of the rounds; answers of the two builds compared.

    python tools/ab_same_box.py <libA.so> <libB.so> [config.yml ...] [--batch N] [--rounds R]
"""
import json
import os
import subprocess
import sys

import numpy as np

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CHILD = r'''
import os, sys, json
import numpy as np
sys.path.insert(0, %(repo)r)
import torch
from dex_retargeting_amd.constants import DEFAULT_URDF_DIR
from dex_retargeting_amd.retargeting_config import RetargetingConfig
from oracle import cases  # (input recipes only)
RetargetingConfig.set_default_urdf_dir(str(DEFAULT_URDF_DIR))
rel, B = sys.argv[1], int(sys.argv[2])
dev = torch.device("cuda:0")
seq = RetargetingConfig.load_from_file(os.path.join(cases.CONFIG_DIR, rel)).build()
prob = cases.problem_from_config(rel)
model = seq.optimizer.device_model()
kp = cases.human_keypoints(B + 1, seed=cases.SEED)
mid = np.repeat(prob.joint_limits.mean(1)[None], B, 0).astype(np.float32)
dex = prob.kind == "dexpilot"
st = np.zeros(B, np.uint32) if dex else None
last = model.retarget(np.ascontiguousarray(kp[:-1]), None, mid, state=st, keypoints=True)
t_kp = torch.from_numpy(np.ascontiguousarray(kp[1:])).to(dev)
t_last = torch.from_numpy(last).to(dev)
t_st0 = torch.from_numpy(st.astype(np.int32)).to(dev) if dex else None
t_st = torch.zeros(B, dtype=torch.int32, device=dev) if dex else None
t_q = torch.empty((B, last.shape[1]), dtype=torch.float32, device=dev)
t_it = torch.zeros(B, dtype=torch.int32, device=dev)
s = torch.cuda.current_stream()
def go(diag=False):
    if dex: t_st.copy_(t_st0)
    model.retarget_dev(B, t_kp.data_ptr(), 0, t_last.data_ptr(), t_st.data_ptr() if dex else 0, t_q.data_ptr(),
                       iters_ptr=t_it.data_ptr() if diag else 0, stream=s.cuda_stream, keypoints=True)
for _ in range(3): go()
torch.cuda.synchronize()
ms = []
for _ in range(5):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(s)
    for _ in range(10): go()
    e1.record(s)
    torch.cuda.synchronize()
    ms.append(e0.elapsed_time(e1) / 10)
go(True)
torch.cuda.synchronize()
it = t_it.cpu().numpy()
q = t_q.cpu().numpy()
np.save(sys.argv[3], q)
print("RESULT " + json.dumps({"ms": float(np.median(ms)), "ms_min": float(min(ms)), "iters_mean": float(it.mean()), "iters_max": int(it.max())}))
'''


def main():
    args = [a for a in sys.argv[1:] if not a.startswith("--")]
    B = int(sys.argv[sys.argv.index("--batch") + 1]) if "--batch" in sys.argv else 65536
    R = int(sys.argv[sys.argv.index("--rounds") + 1]) if "--rounds" in sys.argv else 3
    args = [a for a in args if not a.isdigit()]
    libs = args[:2]
    rels = args[2:] or ["teleop/shadow_hand_right_dexpilot.yml", "offline/leap_hand_right.yml"]
    os.makedirs(os.path.join(REPO, "gpurun_out"), exist_ok=True)
    print(f"# same-box A/B, {B} tracking frames per launch, median of {R} rounds x 5 x 10 launches (HIP events); A = {libs[0]}, B = {libs[1]}")
    for rel in rels:
        res = {0: [], 1: []}
        for r in range(R):
            for i, lib in enumerate(libs):
                env = dict(os.environ, DEXR_LIB=os.path.abspath(lib))
                out = os.path.join(REPO, "gpurun_out", f"_ab_{i}.npy")
                p = subprocess.run([sys.executable, "-c", CHILD % {"repo": REPO}, rel, str(B), out], env=env, capture_output=True, text=True)
                line = [l for l in p.stdout.splitlines() if l.startswith("RESULT ")]
                if not line:
                    print(rel, lib, "FAILED", p.stderr[-500:])
                    continue
                res[i].append(json.loads(line[0][7:]))
        if res[0] and res[1]:
            qa, qb = np.load(os.path.join(REPO, "gpurun_out", "_ab_0.npy")), np.load(os.path.join(REPO, "gpurun_out", "_ab_1.npy"))
            a, b = np.median([x["ms"] for x in res[0]]), np.median([x["ms"] for x in res[1]])
            print(f"{rel:44s} A {a:.4f} ms (iters {res[0][0]['iters_mean']:.3f} / {res[0][0]['iters_max']})   B {b:.4f} ms (iters {res[1][0]['iters_mean']:.3f} / "
                  f"{res[1][0]['iters_max']})   B/A {b / a:.4f}   max |q_A - q_B| {np.abs(qa - qb).max():.2e}  rounds A {[round(x['ms'], 4) for x in res[0]]} B {[round(x['ms'], 4) for x in res[1]]}")


if __name__ == "__main__":
    main()
