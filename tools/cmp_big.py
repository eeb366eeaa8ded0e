import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from dex_retargeting_amd import _lib
from dex_retargeting_amd.constants import DEFAULT_URDF_DIR
from dex_retargeting_amd.retargeting_config import RetargetingConfig
from oracle import cases, solvers
RetargetingConfig.set_default_urdf_dir(str(DEFAULT_URDF_DIR))
B = 65536
# usage: cmp_big.py [--kernel auto|register|quad|lds] config.yml ...
KERNEL = "auto"
args = sys.argv[1:]
if args and args[0] == "--kernel":
    KERNEL, args = args[1], args[2:]
for rel in args:
    seq = RetargetingConfig.load_from_file(os.path.join(cases.CONFIG_DIR, rel)).build()
    prob = cases.problem_from_config(rel)
    m = seq.optimizer.device_model()
    m.tune(kernel={"auto": _lib.KERNEL_AUTO, "register": _lib.KERNEL_REGISTER, "quad": _lib.KERNEL_QUAD, "lds": _lib.KERNEL_LDS}[KERNEL])
    d = cases.reachable_set(prob, B, 0.05)
    dev = torch.device("cuda:0")
    ref, last = torch.from_numpy(d["ref"]).to(dev), torch.from_numpy(d["last"]).to(dev)
    out = torch.empty_like(last); st = torch.zeros(B, dtype=torch.int32, device=dev); it = torch.zeros(B, dtype=torch.int32, device=dev)
    s = torch.cuda.current_stream().cuda_stream
    def run(diag=False):
        st.zero_()
        m.retarget_dev(B, ref.data_ptr(), 0, last.data_ptr(), st.data_ptr() if prob.kind == "dexpilot" else 0, out.data_ptr(), iters_ptr=it.data_ptr() if diag else 0, stream=s)
    run(True); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); 
    for _ in range(3): run()
    e1.record(); torch.cuda.synchronize()
    n = 256
    kw = {}
    if prob.kind == "dexpilot":
        w, rv, _ = prob.dexpilot_preamble(d["ref"][:n], np.zeros((n, prob.n_pair), bool)); kw = dict(weights=w, dexpilot_ref=rv)
    want = solvers.solve_lm_batched(prob, d["ref"][:n], d["fixed"][:n], d["last"][:n], newton=True, max_iter=100, **kw)
    dq = np.abs(out[:n].cpu().numpy() - want).max(1)
    print(f"{rel:44s} kernel={KERNEL}/{m.kernel()} ms={e0.elapsed_time(e1)/3:8.3f} iters={it.float().mean().item():.1f} p99dq={np.percentile(dq,99):.1e} within1e-4={np.mean(dq<1e-4):.3f}")
