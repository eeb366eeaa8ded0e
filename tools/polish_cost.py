import os, sys, numpy as np
sys.path.insert(0, os.getcwd())
import torch
from dex_retargeting_amd import _lib
from dex_retargeting_amd.constants import DEFAULT_URDF_DIR
from dex_retargeting_amd.retargeting_config import RetargetingConfig
from oracle import cases
RetargetingConfig.set_default_urdf_dir(str(DEFAULT_URDF_DIR))
B=65536
kp = cases.human_keypoints(B + 1, seed=cases.SEED)
dev=torch.device("cuda:0"); s=torch.cuda.current_stream()
t_kp=torch.from_numpy(np.ascontiguousarray(kp[1:])).to(dev)
for rel in ["teleop/schunk_svh_hand_right_dexpilot.yml","teleop/inspire_hand_right_dexpilot.yml","teleop/allegro_hand_right_dexpilot.yml","teleop/schunk_svh_hand_right.yml"]:
    prob=cases.problem_from_config(rel)
    model=RetargetingConfig.load_from_file(os.path.join(cases.CONFIG_DIR, rel)).build().optimizer.device_model()
    dex=prob.kind=="dexpilot"
    mid=np.repeat(prob.joint_limits.mean(1)[None],B,0).astype(np.float32)
    last=model.retarget(np.ascontiguousarray(kp[:-1]),None,mid,state=np.zeros(B,np.uint32) if dex else None,keypoints=True)
    t_last=torch.from_numpy(last).to(dev); t_q=torch.empty((B,prob.n_opt),dtype=torch.float32,device=dev); t_st=torch.zeros(B,dtype=torch.int32,device=dev)
    for pol in (1,0):
        opts=_lib.default_options(polish=pol)
        def go():
            if dex: t_st.zero_()
            model.retarget_dev(B,t_kp.data_ptr(),0,t_last.data_ptr(),t_st.data_ptr() if dex else 0,t_q.data_ptr(),stream=s.cuda_stream,keypoints=True,opts=opts)
        for _ in range(2): go()
        ev=[(torch.cuda.Event(enable_timing=True),torch.cuda.Event(enable_timing=True)) for _ in range(5)]
        for a,b in ev:
            a.record(s); go(); b.record(s)
        torch.cuda.synchronize()
        print(rel, "polish",pol, "%.3f ms"%np.median([a.elapsed_time(b) for a,b in ev]))
