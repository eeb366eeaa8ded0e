#!/usr/bin/env python3
"""CPU lab for the solver's damping / step dynamics (no GPU needed).

A float64 numpy restatement of the per-lane loop of the small-component solve kernel
(dex_retargeting_amd/csrc/dexr_kernel.hpp, "persistent lanes": one FK + one fused value/gradient/Hessian per pass,
accepted model kept, a rejected step only re-solves with more damping), for vector models whose components are their
terms (Allegro / LEAP / Ability / Inspire VectorOptimizer: one finger per term).  It reproduces the ITERATION COUNTS of
the kernel (the quantity that sets the critical path of a launch: a frame that needs 17 passes keeps its wave alive
for 17 x 2.7 us) so that damping rules can be compared on the host before a GPU-minute is spent.

    python tools/lm_lab.py [config.yml] [n_sequences]

Model functions come from oracle/ (this is a development tool, not product code).
"""
import os
import sys

import numpy as np

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
from oracle import cases, solvers  # noqa: E402
from oracle.objectives import smooth_l1  # noqa: E402


class CompModel:
    """Per-component value / gradient / (Newton) Hessian of F for a vector problem whose terms do not share joints."""

    def __init__(self, prob, ref, last):
        self.prob, self.ref, self.last = prob, ref, last.astype(np.float32).astype(np.float64)
        B = ref.shape[0]
        x0 = np.clip(self.last, *prob.bounds)
        _, Jt, _, _ = solvers._terms(prob, x0[:4], ref[:4], None, {})
        T = Jt.shape[1]
        nz = (np.abs(Jt).max(axis=(0, 2)) > 0)  # (T, n)
        assert nz.sum(0).max() <= 1, "terms share joints: not a per-term component model"
        m = int(nz.sum(1).max())
        self.vars = np.zeros((T, m), int)
        self.vmask = np.zeros((T, m), bool)
        for t in range(T):
            idx = np.nonzero(nz[t])[0]
            self.vars[t, : len(idx)] = idx
            self.vmask[t, : len(idx)] = True
        self.T, self.m, self.B = T, m, B
        self.lo, self.hi = (b.astype(np.float32).astype(np.float64)[self.vars] for b in prob.bounds)  # the kernels' f32 box

    def full(self, xc):
        x = self.last.copy()
        B = xc.shape[0]
        x[np.arange(B)[:, None, None], self.vars[None]] = np.where(self.vmask[None], xc, x[np.arange(B)[:, None, None], self.vars[None]])
        return x

    def __call__(self, xc, newton=True):
        """xc (B,T,m) -> F (B,T) incl. regulariser, g (B,T,m) incl. regulariser, H (B,T,m,m) data term only."""
        prob = self.prob
        x = self.full(xc)
        r, Jt, w, _ = solvers._terms(prob, x, self.ref, None, {})
        beta = prob.huber_delta
        d = np.linalg.norm(r, axis=2)
        val, _ = smooth_l1(d, beta)
        quad = d < beta
        psi = np.where(quad, 1.0 / beta, 1.0 / np.maximum(d, 1e-30))
        u = np.einsum("btc,btcn->btn", r, Jt)
        k = np.where(quad, 0.0, 1.0 / np.maximum(d, 1e-30) ** 3)
        B = x.shape[0]
        F = val * w
        g = (psi * w)[..., None] * u
        H = np.einsum("bt,btcn,btcm->btnm", psi * w, Jt, Jt) - np.einsum("bt,btn,btm->btnm", k * w, u, u)
        if newton:
            for t in range(self.T):
                f = np.zeros_like(r)
                f[:, t] = (psi * w)[:, t, None] * r[:, t]
                H[:, t] += solvers._second_order(prob, x, None, f)
        bi = np.arange(B)[:, None, None]
        gc = g[bi, np.arange(self.T)[None, :, None], self.vars[None]]
        Hc = H[bi[..., None], np.arange(self.T)[None, :, None, None], self.vars[None, :, :, None], self.vars[None, :, None, :]]
        dx = np.where(self.vmask[None], xc - self.last[bi, self.vars[None]], 0.0)
        F = F + prob.norm_delta * (dx * dx).sum(2)
        gc = np.where(self.vmask[None], gc + 2 * prob.norm_delta * dx, 0.0)
        Hc = np.where(self.vmask[None, :, :, None] & self.vmask[None, :, None, :], Hc, 0.0)
        return F, gc, Hc


class WholeModel:
    """The whole problem as ONE component (DexPilot, position, Shadow vector): same interface as CompModel."""

    def __init__(self, prob, ref, last, **kw):
        self.prob, self.ref, self.kw = prob, ref, kw
        self.last = last.astype(np.float32).astype(np.float64)
        self.B, self.T, self.m = ref.shape[0], 1, prob.n_opt
        self.vars = np.arange(prob.n_opt)[None]
        self.vmask = np.ones((1, prob.n_opt), bool)
        self.lo, self.hi = (b.astype(np.float32).astype(np.float64)[None] for b in prob.bounds)  # the kernels' f32 box

    def full(self, xc):
        return xc[:, 0]

    def __call__(self, xc, newton=True):
        F, g, H = solvers._model(self.prob, xc[:, 0], self.ref, None, self.last, self.kw, newton=newton)
        H = H - 2 * self.prob.norm_delta * np.eye(self.m)[None]
        return F[:, None], g[:, None], H[:, None]


def kernel_lm(cm, *, lam0=1e-4, tol=2e-6, max_iter=64, step_cap=0.3, lam_jump=1.0, lam_fastdec=0.1, blind_tol_scale=10.0,
              max_blind=8, stall_from=2, stall_ratio=0.9, stall_cap=20.0, newton=True, eps=5.96e-8, gn_when=None,
              cap_mode="inf", lam_floor=1e-9, trace=None, inner_retry=0, cap_grow=0.0, cap_max=1.2, gersh_at=99, retry_mult=None, pivot_rule=None, pivot_floor=1e-3, blind_contract=0.0, neg_boost=0.0, jump_mode="hd", noise_scale=None,
              dec_rule=None, accel=0.0, accel_rho=0.9, gn_mode=None, gn_exit=0.05, gn_lam=None, tr_retry=0, tr_factor=2.0, tr_pow=1.0,
              lam_start=None, first_jump=None, fail_floor=None, blind_guard=None):
    """Returns (x (B,T,m), iters (B,T)).  `eps`: rounding unit of the kernel's arithmetic (float32) for the
    below-the-floor logic.  gn_when: optional callable(F, lam, it) -> bool mask selecting Gauss-Newton models."""
    B, T, m = cm.B, cm.T, cm.m
    delta = cm.prob.norm_delta
    bi = np.arange(B)[:, None, None]
    x = np.clip(cm.last[bi, cm.vars[None]], cm.lo[None], cm.hi[None])
    F, gs, Hs = cm(x, newton)
    if gn_mode:  # per-frame model switch: Gauss-Newton (PSD) Hessian after an indefinite Newton model, Newton again near the end
        Hn, Hg = Hs, cm(x, False)[2]
        gn = np.zeros((B, T), bool)
    lam = np.full((B, T), lam0 if lam_start is None else lam_start)
    nu = np.full((B, T), 2.0)
    sprev = np.full((B, T), 1e30)
    blind = np.zeros((B, T), int)
    iters = np.zeros((B, T), int)
    done = np.zeros((B, T), bool)
    eye = np.eye(m)[None, None]
    vm = cm.vmask[None]
    lfail = np.zeros((B, T))  # fail_floor = (mult, relax): the largest damping at which the factorisation failed, relaxed per accepted step
    for _ in range(max_iter + 2):
        if done.all():
            break
        if gn_mode:
            Hs = np.where(gn[..., None, None], Hg, Hn)
        act = ((x <= cm.lo[None]) & (gs > 0)) | ((x >= cm.hi[None]) & (gs < 0))
        free = vm & ~act
        ff = free[..., :, None] & free[..., None, :]
        Hm = np.where(ff, Hs, 0.0) + np.where(free, 2 * delta + lam[..., None], 1.0)[..., None] * eye
        Hm = np.where(ff | (eye > 0), Hm, 0.0)
        gm = np.where(free, gs, 0.0)
        ok = np.linalg.eigvalsh(Hm).min(-1) > 1e-30
        for _r in range(inner_retry):  # indefinite damped model: raise lambda and factor again within the same pass
            bad = ~ok & ~done
            if not bad.any():
                break
            hd0 = np.where(vm, np.einsum("btii->bti", Hs), 0.0).sum(-1) / vm.sum(-1)
            if _r >= gersh_at:  # Gershgorin: lambda that makes the free block diagonally dominant
                Hf = np.where(ff, Hs, 0.0)
                offsum = np.abs(Hf).sum(-1) - np.abs(np.einsum("btii->bti", Hf))
                need = (offsum - np.einsum("btii->bti", Hf) - 2 * delta)
                need = np.where(free, need, -1e30).max(-1)
                lam = np.where(bad, np.maximum(lam, 1.05 * need + 1e-6), lam)
            elif retry_mult:
                lam = np.where(bad, np.maximum(np.maximum(lam, 1e-6) * retry_mult, lam_jump * np.abs(hd0)), lam)
            else:
                lam = np.where(bad, np.maximum(np.maximum(lam, 1e-6) * nu, lam_jump * np.abs(hd0)), lam)
                nu = np.where(bad, nu * 2, nu)
            Hm = np.where(ff, Hs, 0.0) + np.where(free, 2 * delta + lam[..., None], 1.0)[..., None] * eye
            Hm = np.where(ff | (eye > 0), Hm, 0.0)
            ok = np.linalg.eigvalsh(Hm).min(-1) > 1e-30
        if pivot_rule is None:
            Hsafe = np.where(ok[..., None, None], Hm, eye)
            d = -np.linalg.solve(Hsafe, gm[..., None])[..., 0]
        else:  # modified Cholesky: a non-positive pivot is replaced (no retry, no failed pass)
            Lm = np.zeros_like(Hm)
            A = Hm.copy()
            scale = np.abs(np.einsum("btii->bti", Hm)).max(-1) + 1e-30
            for j in range(m):
                dj = A[..., j, j] - (Lm[..., j, :j] ** 2).sum(-1)
                bad = ~(dj > 1e-30 * 0 + pivot_floor * scale * (pivot_rule == "floor"))
                if pivot_rule == "abs":
                    bad = ~(dj > 1e-12 * scale)
                    dj = np.where(bad, np.maximum(np.abs(dj), pivot_floor * scale), dj)
                elif pivot_rule == "kernel":
                    bad = ~(dj > 1e-30)
                    modified = modified | bad if j else bad
                    dj = np.where(bad, np.maximum(np.abs(dj), np.maximum(2 * delta + lam, pivot_floor * scale)), dj)
                else:
                    dj = np.where(bad, pivot_floor * scale, dj)
                Lm[..., j, j] = np.sqrt(dj)
                for i in range(j + 1, m):
                    Lm[..., i, j] = (A[..., i, j] - (Lm[..., i, :j] * Lm[..., j, :j]).sum(-1)) / Lm[..., j, j]
            y = np.linalg.solve(Lm, -gm[..., None])
            d = np.linalg.solve(np.swapaxes(Lm, -1, -2), y)[..., 0]
            ok = np.ones_like(ok)
            if pivot_rule == "kernel":
                ok = ~modified  # gates last_step / below_floor only (see accept below)
        dmax = np.abs(np.where(vm, d, 0)).max(-1)
        if first_jump is not None and _ == 0:  # the solve's FIRST model is indefinite or proposes a wild step: start damped
            kf, lbig = first_jump
            wild = ~done & (~ok | (dmax > kf * step_cap))
            lam = np.where(wild, np.maximum(lam, lbig), lam)
            Hm = np.where(ff, Hs, 0.0) + np.where(free, 2 * delta + lam[..., None], 1.0)[..., None] * eye
            Hm = np.where(ff | (eye > 0), Hm, 0.0)
            ok2 = np.linalg.eigvalsh(Hm).min(-1) > 1e-30
            Hsafe = np.where(ok2[..., None, None], Hm, eye)
            d2 = -np.linalg.solve(Hsafe, gm[..., None])[..., 0]
            d = np.where(wild[..., None], d2, d)
            ok = np.where(wild, ok2, ok)
            dmax = np.abs(np.where(vm, d, 0)).max(-1)
        for _r in range(tr_retry):  # trust region by damping: a step far beyond the radius is re-solved with more damping
            big = ok & ~done & (dmax > tr_factor * step_cap) & (step_cap > 0)
            if not big.any():
                break
            hd0 = np.where(vm, np.einsum("btii->bti", Hs), 0.0).sum(-1) / vm.sum(-1)
            lam = np.where(big, np.maximum(lam, 1e-3 * np.abs(hd0)) * (dmax / step_cap) ** tr_pow, lam)
            Hm = np.where(ff, Hs, 0.0) + np.where(free, 2 * delta + lam[..., None], 1.0)[..., None] * eye
            Hm = np.where(ff | (eye > 0), Hm, 0.0)
            ok2 = np.linalg.eigvalsh(Hm).min(-1) > 1e-30
            Hsafe = np.where(ok2[..., None, None], Hm, eye)
            d2 = -np.linalg.solve(Hsafe, gm[..., None])[..., 0]
            d = np.where(big[..., None], d2, d)
            ok = np.where(big, ok2, ok)
            dmax = np.abs(np.where(vm, d, 0)).max(-1)
        gd = -(gm * d).sum(-1)
        dd = (np.where(vm, d, 0) ** 2).sum(-1)
        if cap_grow > 0 and _ == 0:
            cap = np.full((B, T), step_cap)
        capv = cap if cap_grow > 0 else step_cap
        alpha = np.where((step_cap > 0) & (dmax > capv), capv / np.maximum(dmax, 1e-300), 1.0)
        if neg_boost > 0 and pivot_rule is not None:  # indefinite model (modified pivot): go to the trust radius
            alpha = np.where(~ok & (step_cap > 0), np.minimum(capv / np.maximum(dmax, 1e-300), neg_boost), alpha)
        pred = alpha * (1 - 0.5 * alpha) * gd + 0.5 * alpha * alpha * lam * dd
        xt = np.where(vm, np.clip(x + alpha[..., None] * d, cm.lo[None], cm.hi[None]), x)
        smax = np.abs(xt - x).max(-1)
        last_step = ok & (smax < blind_tol_scale * tol) & (lam <= lam0)
        if blind_contract > 0:  # only when the previous step shows the fast (quadratic) contraction of a clean Newton tail
            last_step = last_step & ((smax < 10 * tol) | (smax < blind_contract * sprev))
        if blind_guard is not None:  # callable(act, free, gs, Hs, step, smax) -> (B, T) bool: may the step be taken unverified?
            last_step = last_step & blind_guard(act & vm, free, gs, Hs, xt - x, smax)
        Ft, gt, Ht = cm(np.where(done[..., None], x, xt), newton)
        noise = 16 * eps * np.abs(F) if noise_scale is None else noise_scale * np.abs(F)
        finite = np.isfinite(Ft)
        below = ok & finite & (pred <= noise) & (smax < 1e-2)
        accept = ok & finite & ((Ft <= F) | below)
        if pivot_rule == "kernel":
            accept = finite & ((Ft <= F) | below)
        take_last = last_step & finite & ~done
        live = ~done
        iters = iters + live
        # accepted (regular)
        acc = live & accept & ~take_last
        rho = (F - Ft) / np.maximum(pred, 1e-30)
        tt = 2 * rho - 1
        shrink = np.where(below, 1 / 3, np.maximum(1 / 3, 1 - tt ** 3))
        if lam_fastdec > 0:
            shrink = np.where(rho > 0.9, lam_fastdec, shrink)
        if dec_rule is not None:
            shrink = dec_rule(rho, shrink, below)
        if accel > 0:  # consecutive accurate steps: the shrink factor itself shrinks geometrically
            if _ == 0:
                streak = np.zeros((B, T), int)
            good = accept & (np.abs(rho - 1) < 1 - accel_rho)
            streak_new = np.where(good, streak + 1, 0)
            shrink = np.where(good, np.maximum((1 / 3) ** streak_new, accel), shrink)
        stalled = below & (blind >= stall_from) & (smax > stall_ratio * sprev) & (smax < stall_cap * tol)
        newblind = np.where(below, blind + 1, 0)
        lam_ok = max(2 * delta, 10 * lam0)
        fin_acc = acc & (((smax < tol) & (lam <= lam_ok)) | stalled | (newblind >= max_blind))
        overdamped = acc & ~fin_acc & (smax < tol)  # tiny step because of the damping: shrink it, go on
        rej = live & ~accept & ~take_last
        lam_rej = np.maximum(lam, 1e-6) * nu
        if lam_jump > 0:
            if jump_mode == "keff":  # quad kernel: curvature of the damped model along the failed step
                lam_rej = np.maximum(lam_rej, lam_jump * gd / np.maximum(dd, 1e-30))
            else:
                hd = np.where(vm, np.einsum("btii->bti", Hs), 0.0).sum(-1) / vm.sum(-1)
                lam_rej = np.maximum(lam_rej, lam_jump * hd)
        if cap_grow > 0:
            capped = alpha < 1.0
            cap = np.where(acc & capped & (rho > 0.5), np.minimum(cap * cap_grow, cap_max), np.where(rej, np.maximum(cap * 0.5, 0.1), cap))
        if trace is not None:
            b_, t_ = trace
            print(f"pass {_:2d} ok={bool(ok[b_, t_])} lam={lam[b_, t_]:.2e} alpha={alpha[b_, t_]:.3f} dmax={dmax[b_, t_]:.3f} "
                  f"pred={pred[b_, t_]:.3e} F={F[b_, t_]:.4e} Ft={Ft[b_, t_]:.4e} rho={rho[b_, t_]:.2f} acc={bool(acc[b_, t_])} "
                  f"minEig={np.linalg.eigvalsh(np.where(ff, Hs, 0.0)[b_, t_]).min():.2e} done={bool(done[b_, t_])}")
        upd = acc | take_last
        if gn_mode:
            Hgt = cm(np.where(done[..., None], x, xt), False)[2]
            Hn = np.where(upd[..., None, None], Ht, Hn)
            Hg = np.where(upd[..., None, None], Hgt, Hg)
            enter = live & ~ok & ~gn
            leave = acc & gn & (smax < gn_exit)
            if gn_lam is not None:  # the PSD model needs far less damping than the jump that followed the failure
                lam_rej = np.where(enter, np.maximum(lam, gn_lam), lam_rej)
            gn = (gn | enter) & ~leave
        x = np.where(upd[..., None], xt, x)
        F = np.where(upd, Ft, F)
        gs = np.where(upd[..., None], gt, gs)
        Hs = np.where(upd[..., None, None], Ht, Hs)
        if fail_floor is not None:
            lfail = np.where(live & ~ok, np.maximum(lfail, lam), lfail)
        lam = np.where(acc, np.maximum(lam * shrink, lam_floor), np.where(rej, lam_rej, lam))
        if fail_floor is not None:
            lam = np.where(acc, np.maximum(lam, fail_floor[0] * lfail), lam)
            lfail = np.where(acc, lfail * fail_floor[1], lfail)
        lam = np.where(overdamped, np.maximum(0.1 * lam, 0.5 * lam_ok), lam)
        nu = np.where(acc, 2.0, np.where(rej, nu * 2, nu))
        blind = np.where(acc, newblind, blind)
        if accel > 0:
            streak = np.where(acc, streak_new, np.where(rej, 0, streak))
        sprev = np.where(acc, smax, sprev)
        done = done | take_last | fin_acc | (rej & (lam > 1e10)) | (rej & finite & (smax < tol)) | (live & (iters >= max_iter))
    return x, iters


def dataset(rel, n_seq=4, seed=3):
    """Tracking frames incl. the fixture wrap (frame 620 -> 0): last = tight oracle solution of frame b, target b+1."""
    prob = cases.problem_from_config(rel)
    n = 621 * n_seq
    kp = cases.human_keypoints(n + 1, seed=seed)
    ref = cases.ref_from_keypoints(prob, kp).astype(np.float32)
    mid = np.repeat(prob.joint_limits.mean(1)[None], n, 0).astype(np.float32)
    last = solvers.solve_lm_batched(prob, ref[:-1], None, mid, newton=True, max_iter=100).astype(np.float32)
    return prob, ref[1:], last


def report(name, iters):
    fr = iters.max(1)
    tile = fr[: len(fr) // 64 * 64].reshape(-1, 64).max(1)
    h = np.bincount(fr)
    print(f"{name:58s} mean {fr.mean():5.2f}  tile-max mean {tile.mean():5.2f}  max {fr.max():3d}  >=10: {(fr >= 10).sum():4d}  "
          f">=13: {(fr >= 13).sum():4d}   hist {h.tolist()}")


if __name__ == "__main__":
    rel = sys.argv[1] if len(sys.argv) > 1 else "teleop/allegro_hand_right.yml"
    n_seq = int(sys.argv[2]) if len(sys.argv) > 2 else 4
    prob, ref, last = dataset(rel, n_seq)
    cm = CompModel(prob, ref, last)
    want = solvers.solve_lm_batched(prob, ref, None, last, newton=True, max_iter=100)
    bi = np.arange(cm.B)[:, None, None]

    def run(name, **kw):
        x, it = kernel_lm(cm, **kw)
        dq = np.abs(cm.full(x) - want).max(1)
        report(name + f"  [>1e-4: {(dq > 1e-4).sum()}]", it)

    run("kernel defaults")
    for cap in (0.0, 0.5, 1.0):
        run(f"step_cap={cap}", step_cap=cap)
    run("gauss-newton", newton=False)
    for l0 in (1e-3, 1e-2):
        run(f"lam0={l0}", lam0=l0)
    run("jump=0.3", lam_jump=0.3)
    run("jump=3", lam_jump=3.0)
    run("fastdec=0.3", lam_fastdec=0.3)
