import os, sys, time
sys.path.insert(0, os.getcwd())
import numpy as np, torch
from dex_retargeting_amd.constants import DEFAULT_URDF_DIR
from dex_retargeting_amd.retargeting_config import RetargetingConfig
from oracle import cases
RetargetingConfig.set_default_urdf_dir(str(DEFAULT_URDF_DIR))
seq = RetargetingConfig.load_from_file(os.path.join(cases.CONFIG_DIR, "teleop/allegro_hand_right.yml")).build()
prob = cases.problem_from_config("teleop/allegro_hand_right.yml")
model = seq.optimizer.device_model()
d = cases.reachable_set(prob, 1, 0.05)
for _ in range(3): model.retarget(d["ref"], None, d["last"])
t=time.perf_counter(); model.retarget(d["ref"], None, d["last"]); print("idle call ms", (time.perf_counter()-t)*1e3)
side = torch.cuda.Stream()
big = torch.empty(1 << 28, dtype=torch.float32, device="cuda:0")
torch.cuda.synchronize()
done = torch.cuda.Event()
t0=time.perf_counter()
with torch.cuda.stream(side):
    for _ in range(100): big.mul_(1.0001)
    done.record(side)
t1=time.perf_counter()
q = model.retarget(d["ref"], None, d["last"])
t2=time.perf_counter()
print("enqueue side ms", (t1-t0)*1e3, "call under load ms", (t2-t1)*1e3, "side still running:", not done.query())
torch.cuda.synchronize(); print("side total ms", (time.perf_counter()-t0)*1e3)
