"""(GPU box) Launch time / iteration tail / parity of the small-component kernel for several values of one damping knob
(a dexr_tuning field, default lam_recover) on six vector models: how damping changes are measured before they become defaults.

    python tools/damping_probe.py [knob v1 v2 ...]"""
import os, sys
sys.path.insert(0, os.getcwd())
import numpy as np, torch
import bench_data
from dex_retargeting_amd.constants import DEFAULT_URDF_DIR
from dex_retargeting_amd.retargeting_config import RetargetingConfig
from oracle import cases, solvers
RetargetingConfig.set_default_urdf_dir(str(DEFAULT_URDF_DIR))
dev = torch.device("cuda:0")
for rel in ("teleop/allegro_hand_right.yml", "teleop/leap_hand_right.yml", "teleop/inspire_hand_right.yml", "teleop/ability_hand_right.yml", "teleop/allegro_hand_left.yml", "teleop/leap_hand_left.yml"):
    seq = RetargetingConfig.load_from_file(os.path.join(cases.CONFIG_DIR, rel)).build()
    prob = cases.problem_from_config(rel)
    m = seq.optimizer.device_model()
    B = 65536
    kp = bench_data.human_keypoints(B + 1)
    mid = np.repeat(seq.joint_limits.mean(1)[None], B, 0).astype(np.float32)
    last = m.retarget(np.ascontiguousarray(kp[:-1]), None, mid, keypoints=True)
    t_kp, t_last = torch.from_numpy(np.ascontiguousarray(kp[1:])).to(dev), torch.from_numpy(last).to(dev)
    out = torch.empty_like(t_last); it = torch.zeros(B, dtype=torch.int32, device=dev)
    ref = cases.ref_from_keypoints(prob, kp[1:2049]).astype(np.float32)
    want = solvers.solve_lm_batched(prob, ref, None, last[:2048], newton=True, max_iter=100)
    s = torch.cuda.current_stream().cuda_stream
    knob = sys.argv[1] if len(sys.argv) > 1 else "lam_recover"
    base = getattr(m.get_tuning(), knob)
    vals = [type(base)(v) for v in sys.argv[2:]] or [0.0, 0.03, 0.01, 0.003, 0.001]
    for fd in [base] + vals:
        m.tune(**{knob: fd})
        for _ in range(3): m.retarget_dev(B, t_kp.data_ptr(), 0, t_last.data_ptr(), 0, out.data_ptr(), iters_ptr=it.data_ptr(), stream=s, keypoints=True)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20): m.retarget_dev(B, t_kp.data_ptr(), 0, t_last.data_ptr(), 0, out.data_ptr(), stream=s, keypoints=True)
        e1.record(); torch.cuda.synchronize()
        dq = np.abs(out[:2048].cpu().numpy() - want).max(1)
        itn = it.cpu().numpy()
        if fd == vals[-1]:
            m.tune(**{knob: base})
        print(f"{rel:34s} {knob} {fd:8.4g}: {e0.elapsed_time(e1)/20*1e3:7.1f} us  it mean {itn.mean():.3f} max {itn.max()}  >=8: {(itn>=8).sum()}  max dq {dq.max():.2e} far {(dq>1e-4).sum()}")
