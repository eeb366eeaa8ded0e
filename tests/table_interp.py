"""Test-only numpy interpreter of the compiled kinematic tables (include/dexr_tables.h).

Lets the CPU test-suite check dex_retargeting_amd/model_compiler.py against the oracle without a GPU.
It is NOT part of the product: the product evaluates these tables only in csrc/ (HIP).
"""
import numpy as np

SRC_OPT, SRC_FIXED, SRC_MIMIC, SRC_DIRECT = 0, 1, 2, 3


def joint_values(comp, x=None, fixed=None, q_full=None):
    nj = int(comp["n_joint"])
    B = (x if x is not None else q_full).shape[0]
    q = np.zeros((B, nj))
    for k in range(nj):
        kind, idx = int(comp["src_kind"][k]), int(comp["src_idx"][k])
        if kind == SRC_OPT:
            q[:, k] = x[:, idx]
        elif kind == SRC_FIXED:
            q[:, k] = comp["mult"][k] * fixed[:, idx] + comp["off"][k]
        elif kind == SRC_DIRECT:
            q[:, k] = q_full[:, idx]
    for k in range(nj):
        if int(comp["src_kind"][k]) == SRC_MIMIC:
            q[:, k] = float(comp["mult"][k]) * q[:, int(comp["src_idx"][k])] + float(comp["off"][k])
    return q


def frame_positions(comp, q):
    """q (B,nj) joint values in table order -> (B,n_frame,3) world positions, plus world axes/origins."""
    B = q.shape[0]
    nj, nf = int(comp["n_joint"]), int(comp["n_frame"])
    P = np.zeros((B, nf, 3))
    for f in range(int(comp["n_base_frame"])):
        P[:, f] = comp["frame_off"][f]
    R = np.broadcast_to(np.eye(3), (B, 3, 3)).copy()
    p = np.zeros((B, 3))
    slots = {}
    axes, orgs = [], []
    for k in range(nj):
        rs = int(comp["restore"][k])
        if rs == -2:
            R = np.broadcast_to(np.eye(3), (B, 3, 3)).copy()
            p = np.zeros((B, 3))
        elif rs >= 0:
            R, p = slots[rs][0].copy(), slots[rs][1].copy()
        X = comp["X"][k].astype(np.float64)
        p = p + R @ X[9:]
        R = R @ X[:9].reshape(3, 3)
        if int(comp["jtype"][k]) == 0:
            c, s = np.cos(q[:, k]), np.sin(q[:, k])
            c0, c1 = R[:, :, 0].copy(), R[:, :, 1].copy()
            R[:, :, 0] = c[:, None] * c0 + s[:, None] * c1
            R[:, :, 1] = -s[:, None] * c0 + c[:, None] * c1
        else:
            p = p + R[:, :, 2] * q[:, k:k + 1]
        axes.append(R[:, :, 2].copy())
        orgs.append(p.copy())
        sv = int(comp["save"][k])
        if sv >= 0:
            slots[sv] = (R.copy(), p.copy())
        for f in range(int(comp["fbeg"][k]), int(comp["fend"][k])):
            P[:, f] = p + R @ comp["frame_off"][f].astype(np.float64)
    return P, axes, orgs


def solve_vector(compiled, ref, last, iters=6, scaling=None, norm_delta=None):
    """Test-only CPU solve over the compiled tables of a VECTOR model: `iters` damped Gauss-Newton steps on
    sum_t |p_task - p_origin - s ref_t|^2 / V + norm_delta |x - last|^2 per component, projected onto the box.  Not the
    product's algorithm and not the oracle: deterministic per-item work that depends on every input of the item, for
    tests that need a real table-driven solve on a machine without a GPU (multi-process sharding tests)."""
    h = compiled.header
    s = float(h["scaling"]) if scaling is None else scaling
    nd = float(h["norm_delta"]) if norm_delta is None else norm_delta
    B = last.shape[0]
    x = np.array(last, dtype=np.float64)
    for comp in compiled.comps:
        nj, nt = int(comp["n_joint"]), int(comp["n_term"])
        api = comp["api"][:nj].astype(int)
        lo, hi = comp["lo"][:nj].astype(np.float64), comp["hi"][:nj].astype(np.float64)
        x[:, api] = np.clip(x[:, api], lo, hi)
        for _ in range(iters):
            q = joint_values(comp, x=x, fixed=np.zeros((B, 1)))
            P, axes, orgs = frame_positions(comp, q)
            g = np.zeros((B, nj))
            H = np.zeros((B, nj, nj))
            for t in range(nt):
                ft, fo = int(comp["term_task"][t]), int(comp["term_origin"][t])
                r = P[:, ft] - (P[:, fo] if fo >= 0 else 0.0) - s * ref[:, int(comp["term_ref"][t])].astype(np.float64)
                J = np.zeros((B, 3, nj))
                for k in range(nj):
                    for f, sg in ((ft, 1.0), (fo, -1.0)):
                        if f >= 0 and (int(comp["frame_anc"][f]) >> k) & 1:
                            J[:, :, k] += sg * np.cross(axes[k], P[:, f] - orgs[k])
                g += np.einsum("bck,bc->bk", J, r) / nt
                H += np.einsum("bck,bcl->bkl", J, J) / nt
            dx = x[:, api] - last[:, api].astype(np.float64)
            g = 2 * g + 2 * nd * dx
            H = 2 * H + (2 * nd + 1e-3) * np.eye(nj)[None]
            x[:, api] = np.clip(x[:, api] - np.linalg.solve(H, g[..., None])[..., 0], lo, hi)
    return x.astype(np.float32)


def reduced_model(comp, header, P, axes, orgs, targets, weights, newton=True):
    """Test-only numpy statement of the reduced-variable assembly of csrc/dexr_red.hpp for ONE frame (B = 1 arrays as
    returned by frame_positions): gradient and Hessian of the data term in the component's variables, with mimic
    joints folded while the Jacobian columns are formed and the second-order term accumulated from running per-variable
    axis sums.  targets (n_term, 3), weights (n_term,) incl. the 1/V (or 1/3P) factor."""
    nj, nt, nv = int(comp["n_joint"]), int(comp["n_term"]), int(comp["n_var"])
    kind = int(header["kind"])
    beta = float(header["huber_delta"])
    g = np.zeros(nv)
    H = np.zeros((nv, nv))
    F = 0.0
    for t in range(nt):
        ft, fo = int(comp["term_task"][t]), int(comp["term_origin"][t])
        pt = P[0, ft]
        po = P[0, fo] if fo >= 0 else np.zeros(3)
        r = pt - po - targets[t]
        w = weights[t]
        if kind == 1:  # position: SmoothL1 per coordinate
            quad = np.abs(r) < beta
            F += w * np.where(quad, 0.5 * r * r / beta, np.abs(r) - 0.5 * beta).sum()
            f = w * np.where(quad, r / beta, np.sign(r))
            hw = w * np.where(quad, 1.0 / beta, 0.0 if newton else 1.0 / np.maximum(np.abs(r), 1e-30))
            kap = 0.0
        else:
            d = np.linalg.norm(r)
            quad = d < beta
            F += w * (0.5 * d * d / beta if quad else d - 0.5 * beta)
            idd = 1.0 / beta if quad else 1.0 / d
            psi = w * idd
            f, hw = psi * r, np.full(3, psi)
            kap = 0.0 if quad else psi * idd * idd
        colv = np.zeros((nv, 3))
        for chain, (fr, sg, pf) in enumerate(((ft, 1.0, pt), (fo, -1.0, po))):
            if fr < 0:
                continue
            A = np.zeros((nv, 3))
            seen = set()
            for k in range(nj):
                if not (int(comp["frame_anc"][fr]) >> k) & 1 or int(comp["var"][k]) < 0:
                    continue
                a, o = axes[k][0], orgs[k][0]
                rev = int(comp["jtype"][k]) == 0
                col = sg * (np.cross(a, pf - o) if rev else a)
                v, m = int(comp["var"][k]), float(comp["vmul"][k])
                e = np.cross(col, f)
                if newton:
                    for vv in seen:
                        h = m * (A[vv] @ e)
                        if vv == v:
                            H[v, v] += 2 * h
                        else:
                            H[vv, v] += h
                            H[v, vv] += h
                    if rev:
                        H[v, v] += m * m * (a @ e)
                        A[v] += m * a
                colv[v] += m * col
                seen.add(v)
        g += colv @ f
        u = colv @ r
        H += (colv * hw) @ colv.T - kap * np.outer(u, u)
    return F, g, H
