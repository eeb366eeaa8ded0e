"""Developer probe (GPU): how often the float32 and float64 solves of a far-start batch end in different minima, and which of
the two has the lower objective there (numbers quoted in tests/test_gpu_parity.py::test_full_size_batches_of_the_persistent_kernels)."""
import sys, numpy as np
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tests")
from oracle import cases
import test_gpu_parity as T
for rel in ["teleop/shadow_hand_right_dexpilot.yml", "offline/leap_hand_right.yml", "offline/inspire_hand_right.yml"]:
    seq, prob = T.build(rel)
    model = seq.optimizer.device_model()
    m = 4096
    d = cases.human_set(prob, 65536, seed=11, sigma=0.1)
    dex = prob.kind == "dexpilot"
    st = (lambda n: np.zeros(n, np.uint32)) if dex else (lambda n: None)
    q1 = model.retarget(d["ref"][:m], d["fixed"][:m], d["last"][:m], state=st(m))
    q64 = model.retarget_f64(d["ref"][:m], d["fixed"][:m], d["last"][:m], state=st(m))
    dq = np.abs(q1.astype(np.float64) - q64).max(1)
    f32, g32 = model.eval(d["ref"][:m], d["fixed"][:m], d["last"][:m], q1.astype(np.float64), state=st(m))
    f64, g64 = model.eval(d["ref"][:m], d["fixed"][:m], d["last"][:m], q64, state=st(m))
    # objective incl. regulariser
    nd = prob.norm_delta if hasattr(prob, "norm_delta") else 4e-3
    F32 = f32 + nd * ((q1 - d["last"][:m]) ** 2).sum(1)
    F64 = f64 + nd * ((q64 - d["last"][:m]) ** 2).sum(1)
    far = dq >= 1e-4
    print(rel, "within 1e-4:", (~far).mean(), "n far", far.sum())
    if far.any():
        diff = F32[far] - F64[far]
        print("   F32-F64 on far frames: median %.3e  frac(F32<=F64) %.3f  frac(|diff|<1e-9) %.3f  p10 %.2e p90 %.2e" % (np.median(diff), (diff <= 0).mean(), (np.abs(diff) < 1e-9).mean(), np.percentile(diff, 10), np.percentile(diff, 90)))
        lo, hi = prob.bounds
        def pg(q, g):
            g = g.copy()
            g[(q <= lo + 1e-7) & (g > 0)] = 0
            g[(q >= hi - 1e-7) & (g < 0)] = 0
            return np.abs(g).max(1)
        print("   projected gradient (far): f32 median %.2e p99 %.2e | f64 median %.2e p99 %.2e" % (np.median(pg(q1[far].astype(np.float64), g32[far])), np.percentile(pg(q1[far].astype(np.float64), g32[far]), 99), np.median(pg(q64[far], g64[far])), np.percentile(pg(q64[far], g64[far]), 99)))
