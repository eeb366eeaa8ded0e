#!/usr/bin/env python3
"""CPU: replay the inputs of a GPU launch in the float64 host emulation of the kernels' damping loop (tools/lm_lab.py).

    (GPU box)  python tools/prof_config.py <config.yml> auto 65536 3 slowq_<name>.npz 4096
               -> gpurun_out/head_slowq_<name>.npz: keypoints, start points, DexPilot bits, per-frame iteration counts
    (here)     python tools/lm_replay.py <config.yml> [frames] [predictors]

Prints the GPU's iteration histogram next to the emulation's (same inputs, the quad / sixteen-lane kernels' rules: Rayleigh
quotient damping jump, plain Cholesky), their per-frame agreement, a few rule variants, and -- with a third argument --
how well quantities known at the start point predict the slow frames (longest-first scheduling).
The emulation only reproduces the GPU when it uses the kernels' float32 joint box: a bound that differs in the 8th
digit decides whether a variable is active at the start point.
"""
import os
import sys

import numpy as np

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
sys.path.insert(0, os.path.join(REPO, "tools"))
import lm_lab as L  # noqa: E402
from oracle import cases, solvers  # noqa: E402

rel = sys.argv[1]
n = int(sys.argv[2]) if len(sys.argv) > 2 else 2048
name = rel.split("/")[-1][:-4]
path = os.path.join(REPO, "gpurun_out", f"head_slowq_{name}.npz")
if not os.path.exists(path):  # a committed sample: the first 1 024 frames of a Shadow DexPilot launch (round 2)
    path = os.path.join(REPO, "tools", "data", f"head_{name}.npz")
d = np.load(path)
n = min(n, len(d["iters"]))
prob = cases.problem_from_config(rel)
ref = cases.ref_from_keypoints(prob, d["kp"][:n]).astype(np.float32)
last = d["last"][:n, : prob.n_opt]
kw = {}
if prob.kind == "dexpilot":
    pj = ((d["state"][:n, None] >> np.arange(prob.n_pair, dtype=np.uint32)) & 1).astype(bool)
    w, rv, _ = prob.dexpilot_preamble(ref, pj)
    kw = dict(weights=w, dexpilot_ref=rv)
cm = L.WholeModel(prob, ref, last, **kw)
git = d["iters"][:n]
print("GPU                        mean %.2f p99 %.0f max %d hist %s" % (git.mean(), np.percentile(git, 99), git.max(),
                                                                         np.bincount(git).tolist()[:26]))
Q = dict(lam_jump=1.0, lam_fastdec=0.0, blind_tol_scale=10, jump_mode="keff", noise_scale=1e-12)
want = solvers.solve_lm_batched(prob, ref, None, last, newton=True, max_iter=100, tol=1e-13, **kw)


def run(label, **extra):
    x, it = L.kernel_lm(cm, **{**Q, **extra})
    fr = it[:, 0]
    dq = np.abs(cm.full(x) - want).max(1)
    print(f"{label:26s} mean {fr.mean():.2f} p99 {np.percentile(fr, 99):.0f} max {fr.max()} far {(dq > 1e-4).sum()} "
          f"hist {np.bincount(fr).tolist()[:26]}")
    return fr


fr = run("emulation (kernel rules)")
print("  per-frame agreement with the GPU: correlation %.3f, within one pass %.1f %%" % (
    np.corrcoef(fr, git)[0, 1], 100 * (np.abs(fr - git) <= 1).mean()))
run("fast decay 0.1", lam_fastdec=0.1)
run("accelerated decay", accel=1e-3)
run("modified Cholesky", pivot_rule="kernel")
run("GN after indefinite model", gn_mode=True, gn_lam=1e-2, accel=1e-3)

if len(sys.argv) > 3:
    x0 = np.clip(cm.last, cm.lo[0], cm.hi[0])
    F0, g0, H0 = solvers._model(prob, x0, ref, None, cm.last, kw, newton=True)
    act = ((x0 <= cm.lo[0]) & (g0 > 0)) | ((x0 >= cm.hi[0]) & (g0 < 0))
    dmax, mineig, gfree = np.zeros(n), np.zeros(n), np.zeros(n)
    for b in range(n):
        f = ~act[b]
        Hb = H0[b][np.ix_(f, f)] + 1e-4 * np.eye(f.sum())
        mineig[b] = np.linalg.eigvalsh(Hb)[0]
        dmax[b] = min(np.abs(np.linalg.solve(Hb, g0[b][f])).max(), 10.0)
        gfree[b] = np.abs(g0[b][f]).max()
    hard = np.nonzero(git >= 15)[0]
    print(f"frames with >= 15 iterations: {len(hard)} of {n}; share of them in the top x % by ...")
    for label, key in (("F(x0)", F0), ("|g| free", gfree), ("first Newton step", dmax), ("-min eig", -mineig)):
        order = np.argsort(-key)
        print(f"  {label:18s}", "  ".join(
            f"{fq:.0%}: {np.isin(hard, order[:int(fq * n)]).mean():.2f} (max iters in the rest {git[order[int(fq * n):]].max()})"
            for fq in (0.02, 0.05, 0.1, 0.2)))
