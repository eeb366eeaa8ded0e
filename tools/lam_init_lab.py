#!/usr/bin/env python3
"""CPU lab: initial damping and decay rule of the sixteen-lane kernel's iteration, on recorded ONLINE pairs and tracking batches.

tools/online_probe.py (GPU, DEXR_PROBE_SAVE=1) records the (ref, last, state) pairs of the one-frame-per-call loop with the
GPU's per-frame iteration counts; tools/lm_lab.kernel_lm reproduces those counts (Shadow vector: the same histogram).  Most
online frames start at an INDEFINITE Newton model: with the solve's first damping at lam0 = 1e-4 the first passes are capped or
rejected, the rejection raises lambda to the curvature scale, and at x 0.1 per accepted step four or five over-damped passes
follow.  This script compares: the kernel's rules; a larger first damping `lam_init` (termination thresholds still use lam0);
a faster decay after a step the model predicted almost exactly.

    python tools/lam_init_lab.py online <config.yml> [frames]      (needs gpurun_out/online_pairs_<name>.npz)
    python tools/lam_init_lab.py track <config.yml> [n_sequences]  (fixture tracking frames + 2 mm noise, as bench.py)
"""
import os
import sys

import numpy as np

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
sys.path.insert(0, os.path.join(REPO, "tools"))
import lm_lab as L  # noqa: E402
from oracle import cases, solvers  # noqa: E402

mode, rel = sys.argv[1], sys.argv[2]
prob = cases.problem_from_config(rel)
kw = {}
if mode == "online":
    name = rel.split("/")[-1][:-4]
    d = np.load(os.path.join(REPO, "gpurun_out", f"online_pairs_{name}.npz"))
    n = min(int(sys.argv[3]) if len(sys.argv) > 3 else 300, len(d["iters"]))
    ref, last, git = d["refs"][:n], d["lasts"][:n], d["iters"][:n]
    if prob.kind == "dexpilot":
        pj = ((d["states"][:n, None] >> np.arange(prob.n_pair, dtype=np.uint32)) & 1).astype(bool)
        w, rv, _ = prob.dexpilot_preamble(ref, pj)
        kw = dict(weights=w, dexpilot_ref=rv)
    print(f"{rel} online pairs: {n} frames, GPU iterations mean {git.mean():.2f} max {git.max()}")
else:
    n = 621 * (int(sys.argv[3]) if len(sys.argv) > 3 else 1)
    kp = cases.human_keypoints(n + 1, seed=3)
    ref_all = cases.ref_from_keypoints(prob, kp).astype(np.float32)
    mid = np.repeat(prob.joint_limits.mean(1)[None], n, 0).astype(np.float32)
    kwp = {}
    if prob.kind == "dexpilot":
        w0, rv0, pj0 = prob.dexpilot_preamble(ref_all[:-1], np.zeros((n, prob.n_pair), bool))
        kwp = dict(weights=w0, dexpilot_ref=rv0)
        w1, rv1, _ = prob.dexpilot_preamble(ref_all[1:], pj0)
        kw = dict(weights=w1, dexpilot_ref=rv1)
    last = solvers.solve_lm_batched(prob, ref_all[:-1], None, mid, newton=True, max_iter=100, **kwp).astype(np.float32)
    ref = ref_all[1:]
    print(f"{rel} tracking batch: {n} frames")
cm = L.WholeModel(prob, ref, last, **kw)
want = solvers.solve_lm_batched(prob, ref, None, last, newton=True, max_iter=100, tol=1e-13, **kw)
Fw = prob.total(want, ref, None, last.astype(np.float64), **kw) if hasattr(prob, "total") else None
Q = dict(lam_jump=1.0, lam_fastdec=0.1, blind_tol_scale=10, jump_mode="keff", noise_scale=1e-12, blind_contract=0.1)


def mk(th, f):
    def rule(rho, shrink, below):
        return np.where((rho > th) & ~below, f, shrink)
    return rule


def run(label, **e):
    x, it = L.kernel_lm(cm, **{**Q, **e})
    fr = it[:, 0]
    xf = cm.full(x)
    dq = np.abs(xf - want).max(1)
    worse = 0
    if Fw is not None:
        Fx = prob.total(xf, ref, None, last.astype(np.float64), **kw)
        worse = int(((dq > 1e-4) & (Fx > Fw + 1e-9)).sum())
    print(f"{label:44s} mean {fr.mean():5.2f} p99 {np.percentile(fr, 99):3.0f} max {fr.max():3d} far {(dq > 1e-4).sum():3d} worse {worse:2d}  "
          f"{np.bincount(fr).tolist()[:16]}")


run("kernel rules")
if os.environ.get("LAB_SET", "first") == "init":
    for li in (0.03, 0.1, 0.3):
        run(f"lam_init {li}", lam_start=li)
        for th, f in ((0.99, 0.01), (0.95, 0.01), (0.9, 0.03)):
            run(f"lam_init {li}, rho > {th} -> x {f}", lam_start=li, dec_rule=mk(th, f))
else:  # damped start only for the frames whose first model is indefinite or proposes a step far beyond the trust radius
    for kf in (1.5, 3.0):
        for lb in (0.03, 0.1, 0.3):
            run(f"first step > {kf} cap or indefinite -> lam {lb}", first_jump=(kf, lb))
    run("first > 3 cap -> 0.1, rho > 0.9 -> x 0.03", first_jump=(3.0, 0.1), dec_rule=mk(0.9, 0.03))
    run("first > 1.5 cap -> 0.1, rho > 0.9 -> x 0.03", first_jump=(1.5, 0.1), dec_rule=mk(0.9, 0.03))
