import os, sys, time
sys.path.insert(0, os.getcwd())
import numpy as np, torch
import bench_data
from dex_retargeting_amd import _lib
from dex_retargeting_amd.constants import DEFAULT_URDF_DIR
from dex_retargeting_amd.retargeting_config import RetargetingConfig
from oracle import cases, solvers
RetargetingConfig.set_default_urdf_dir(str(DEFAULT_URDF_DIR))
rel = "offline/panda_gripper.yml"
seq = RetargetingConfig.load_from_file(os.path.join(cases.CONFIG_DIR, rel)).build()
prob = cases.problem_from_config(rel)
opt = seq.optimizer
B = 65536
kp = bench_data.human_keypoints(B + 1)
mid = np.repeat(seq.joint_limits.mean(1)[None], B, 0).astype(np.float32)
m = opt.device_model()
dev = torch.device("cuda:0")
for fam in (_lib.KERNEL_AUTO, _lib.KERNEL_REGISTER, _lib.KERNEL_WIDE):
    m.tune(kernel=fam)
    last = m.retarget(np.ascontiguousarray(kp[:-1]), None, mid, keypoints=True)
    t_kp = torch.from_numpy(np.ascontiguousarray(kp[1:])).to(dev); t_last = torch.from_numpy(last).to(dev)
    out = torch.empty_like(t_last); it = torch.zeros(B, dtype=torch.int32, device=dev)
    st = torch.cuda.current_stream().cuda_stream
    for _ in range(3): m.retarget_dev(B, t_kp.data_ptr(), 0, t_last.data_ptr(), 0, out.data_ptr(), iters_ptr=it.data_ptr(), stream=st, keypoints=True)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10): m.retarget_dev(B, t_kp.data_ptr(), 0, t_last.data_ptr(), 0, out.data_ptr(), stream=st, keypoints=True)
    e1.record(); torch.cuda.synchronize()
    q = out.cpu().numpy()
    ref = cases.ref_from_keypoints(prob, kp[1:513]).astype(np.float32)
    want = solvers.solve_lm_batched(prob, ref, None, last[:512], newton=True, max_iter=100)
    dq = np.abs(q[:512] - want).max(1)
    print("family", fam, "->", m.kernel(), "ms", e0.elapsed_time(e1) / 10, "it mean/max", float(it.float().mean()), int(it.max()), "frac<1e-4", float((dq < 1e-4).mean()), "max dq", float(dq.max()))
