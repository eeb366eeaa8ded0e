#!/usr/bin/env python3
"""CPU lab: how many passes do the SLOW frames of a sixteen-lane-kernel launch need when a wave gives one frame all four
of its rows and every row tries ITS OWN damping value in the pass ("sprint" mode, dexr_wide.hpp)?

    python tools/sprint_lab.py <config.yml> [n_sequences] [P]

The ordinary iteration (tools/lm_lab.kernel_lm with the sixteen-lane kernel's rules) runs for P passes; frames that are
not finished by then are handed to the sprint iteration at their accepted point with fresh damping state -- exactly what
the library does (the first launch stops at max_iter = P, the second starts from its rows of qpos_out).  Printed: the pass
histogram of the ordinary iteration, and for each ladder of damping multipliers the passes the handed-over frames need
in sprint mode next to what they needed in the ordinary iteration, plus the distance of the answers to the tight oracle.

Model functions come from oracle/ (this is a development tool, not product code).
"""
import os
import sys

import numpy as np

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
sys.path.insert(0, os.path.join(REPO, "tools"))
import lm_lab as L  # noqa: E402
from oracle import solvers  # noqa: E402


def sprint_lm(cm, x0, mults, *, lam0=1e-4, tol=2e-6, max_iter=64, step_cap=0.3, lam_jump=1.0, lam_fastdec=0.1,
              blind_tol_scale=10.0, max_blind=8, stall_from=2, stall_ratio=0.9, stall_cap=20.0, noise_scale=1e-12,
              pick="lowest", trace=None):
    """One frame per wave, K = len(mults) rows: row k steps from the accepted point with damping lam * mults[k].
    Returns (x (B, m), passes (B,)).  cm: L.WholeModel.  x0 (B, m): start (accepted) points."""
    B, m = cm.B, cm.m
    K = len(mults)
    mu = np.asarray(mults, float)[None]                     # (1, K)
    delta = cm.prob.norm_delta
    lo, hi = cm.lo[0][None], cm.hi[0][None]
    x = np.clip(x0, lo, hi)
    F, g, H = (a[:, 0] for a in cm(x[:, None], True))
    lam = np.full(B, lam0)
    nu = np.full(B, 2.0)
    sprev = np.full(B, 1e30)
    blind = np.zeros(B, int)
    iters = np.zeros(B, int)
    done = np.zeros(B, bool)
    eye = np.eye(m)[None]
    lam_ok = max(2 * delta, 10 * lam0)
    for it in range(max_iter + 2):
        if done.all():
            break
        act = ((x <= lo) & (g > 0)) | ((x >= hi) & (g < 0))
        free = ~act
        ff = free[:, :, None] & free[:, None, :]
        Hf = np.where(ff, H, 0.0)
        gm = np.where(free, g, 0.0)
        lamk = lam[:, None] * mu                             # (B, K)
        Hm = Hf[:, None] + np.where(free[:, None, :], 2 * delta + lamk[..., None], 1.0)[..., None] * eye[None]
        ok = np.linalg.eigvalsh(Hm).min(-1) > 1e-30          # (B, K)
        Hsafe = np.where(ok[..., None, None], Hm, eye[None])
        d = -np.linalg.solve(Hsafe, np.broadcast_to(gm[:, None, :, None], (B, K, m, 1)))[..., 0]
        dmax = np.abs(d).max(-1)
        gd = -(gm[:, None] * d).sum(-1)
        dd = (d ** 2).sum(-1)
        alpha = np.where((step_cap > 0) & (dmax > step_cap), step_cap / np.maximum(dmax, 1e-300), 1.0)
        pred = alpha * (1 - 0.5 * alpha) * gd + 0.5 * alpha * alpha * lamk * dd
        xt = np.clip(x[:, None] + alpha[..., None] * d, lo[:, None], hi[:, None])
        smax = np.abs(xt - x[:, None]).max(-1)
        Ft = np.empty((B, K))
        gt = np.empty((B, K, m))
        Ht = np.empty((B, K, m, m))
        for k in range(K):
            a, b_, c = cm(np.where(done[:, None], x, xt[:, k])[:, None], True)
            Ft[:, k], gt[:, k], Ht[:, k] = a[:, 0], b_[:, 0], c[:, 0]
        noise = noise_scale * np.abs(F)[:, None]
        finite = np.isfinite(Ft)
        below = ok & finite & (pred <= noise) & (smax < 1e-2)
        accept = ok & finite & ((Ft <= F[:, None]) | below)
        # the blind last step (verified undamped model, tiny step): any row that qualifies ends the solve
        last_step = ok & (smax < blind_tol_scale * tol) & (lamk <= lam0) & ((smax < 10 * tol) | (smax < 0.1 * sprev[:, None]))
        live = ~done
        iters = iters + live
        # the winner: the accepted row with the lowest value (ties: the smallest damping)
        score = np.where(accept, Ft, np.inf)
        if pick == "lowest":
            w = np.argmin(score, 1)
        else:  # the least damped accepted row
            w = np.argmax(accept, 1)
        anyacc = accept.any(1)
        bi = np.arange(B)
        take_last = (last_step & finite).any(1) & live
        wl = np.argmax(last_step & finite, 1)
        w = np.where(take_last, wl, w)
        Fw, predw, smaxw, lamw, beloww = Ft[bi, w], pred[bi, w], smax[bi, w], lamk[bi, w], below[bi, w]
        acc = live & anyacc & ~take_last
        rho = (F - Fw) / np.maximum(predw, 1e-30)
        tt = 2 * rho - 1
        shrink = np.where(beloww, 1 / 3, np.maximum(1 / 3, 1 - tt ** 3))
        if lam_fastdec > 0:
            shrink = np.where(rho > 0.9, lam_fastdec, shrink)
        stalled = beloww & (blind >= stall_from) & (smaxw > stall_ratio * sprev) & (smaxw < stall_cap * tol)
        newblind = np.where(beloww, blind + 1, 0)
        fin_acc = acc & (((smaxw < tol) & (lamw <= lam_ok)) | stalled | (newblind >= max_blind))
        overdamped = acc & ~fin_acc & (smaxw < tol)
        rej = live & ~anyacc & ~take_last
        # every row failed: the next ladder starts above the largest damping tried
        kmax = np.argmax(lamk, 1)
        lam_rej = np.maximum(lamk[bi, kmax], 1e-6) * nu
        if lam_jump > 0:
            lam_rej = np.maximum(lam_rej, lam_jump * gd[bi, kmax] / np.maximum(dd[bi, kmax], 1e-30))
        if trace is not None and live[trace]:
            b = trace
            print(f"pass {it:2d} lam={lam[b]:.2e} F={F[b]:.5e} " + " ".join(
                f"[{'A' if accept[b, k] else ('x' if ok[b, k] else 'f')} {Ft[b, k]:.5e} s={smax[b, k]:.1e}]" for k in range(K)) +
                f" -> {w[b] if anyacc[b] else '-'}")
        upd = acc | take_last
        x = np.where(upd[:, None], xt[bi, w], x)
        F = np.where(upd, Fw, F)
        g = np.where(upd[:, None], gt[bi, w], g)
        H = np.where(upd[:, None, None], Ht[bi, w], H)
        lam = np.where(acc, np.maximum(lamw * shrink, 1e-9), np.where(rej, lam_rej, lam))
        lam = np.where(overdamped, np.maximum(0.1 * lam, 0.5 * lam_ok), lam)
        nu = np.where(acc, 2.0, np.where(rej, nu * 2, nu))
        blind = np.where(acc, newblind, blind)
        sprev = np.where(acc, smaxw, sprev)
        rej_small = rej & (ok & finite & (smax < tol)).all(1)
        done = done | take_last | fin_acc | (rej & (lam > 1e10)) | rej_small | (live & (iters >= max_iter))
    return x, iters


def main():
    rel = sys.argv[1] if len(sys.argv) > 1 else "offline/leap_hand_right.yml"
    n_seq = int(sys.argv[2]) if len(sys.argv) > 2 else 4
    P = int(sys.argv[3]) if len(sys.argv) > 3 else 10
    from oracle import cases
    prob = cases.problem_from_config(rel)
    n = 621 * n_seq
    kp = cases.human_keypoints(n + 1, seed=3)
    ref_all = cases.ref_from_keypoints(prob, kp).astype(np.float32)
    mid = np.repeat(prob.joint_limits.mean(1)[None], n, 0).astype(np.float32)
    kw_prev, kw = {}, {}
    if prob.kind == "dexpilot":  # projection bits of frame b from a cold state; frame b + 1 from those (optimizer.py:466-476)
        w0, rv0, pj0 = prob.dexpilot_preamble(ref_all[:-1], np.zeros((n, prob.n_pair), bool))
        kw_prev = dict(weights=w0, dexpilot_ref=rv0)
        w1, rv1, _ = prob.dexpilot_preamble(ref_all[1:], pj0)
        kw = dict(weights=w1, dexpilot_ref=rv1)
    last = solvers.solve_lm_batched(prob, ref_all[:-1], None, mid, newton=True, max_iter=100, **kw_prev).astype(np.float32)
    ref = ref_all[1:]
    cm = L.WholeModel(prob, ref, last, **kw)
    Q = dict(lam_jump=1.0, lam_fastdec=0.1, blind_tol_scale=10, jump_mode="keff", noise_scale=1e-12, blind_contract=0.1)
    want = solvers.solve_lm_batched(prob, ref, None, last, newton=True, max_iter=100, tol=1e-13, **kw)
    x_full, it_full = L.kernel_lm(cm, **Q)
    fr = it_full[:, 0]
    print(f"{rel}: {cm.B} frames; ordinary iteration: mean {fr.mean():.2f} p99 {np.percentile(fr, 99):.0f} max {fr.max()} "
          f"hist {np.bincount(fr).tolist()}")
    xP, itP = L.kernel_lm(cm, max_iter=P, **Q)
    hard = np.nonzero(itP[:, 0] >= P)[0]
    # (frames that finish exactly at pass P are handed over too: the library cannot tell them apart either)
    print(f"handed over after {P} passes: {len(hard)} frames ({100 * len(hard) / cm.B:.1f} %); they need "
          f"{np.sort(fr[hard] - P)[::-1][:20].tolist()} ... more ordinary passes (mean {np.mean(fr[hard] - P):.1f})")
    sub = L.WholeModel(prob, ref[hard], last[hard], **{k: v[hard] for k, v in kw.items()})
    dq0 = np.abs(x_full[hard, 0] - want[hard]).max(1)
    print(f"  ordinary answers of those frames: far from the tight oracle (> 1e-4): {(dq0 > 1e-4).sum()}")
    for name, mults, pick in (("1 row (restart only)", (1.0,), "lowest"),
                              ("x (1, 10, 100, 1000)", (1, 10, 100, 1000), "lowest"),
                              ("x (.1, 1, 10, 100)", (0.1, 1, 10, 100), "lowest"),
                              ("x (.03, .3, 3, 30)", (0.03, 0.3, 3, 30), "lowest"),
                              ("x (.3, 1, 3, 10)", (0.3, 1, 3, 10), "lowest"),
                              ("x (.1, 1, 10, 100) least damped", (0.1, 1, 10, 100), "first"),
                              ("x (.01, 1, 100, 1e4)", (0.01, 1, 100, 1e4), "lowest")):
        xs, its = sprint_lm(sub, xP[hard, 0], mults, pick=pick)
        dq = np.abs(xs - want[hard]).max(1)
        print(f"  sprint {name:34s} passes mean {its.mean():5.2f} max {its.max():3d} top {np.sort(its)[::-1][:12].tolist()}  "
              f"far {(dq > 1e-4).sum()}")


if __name__ == "__main__":
    main()
