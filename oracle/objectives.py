"""ORACLE (test infrastructure only -- never imported by the product path).

Float64, batched numpy restatement of the three objective closures of the reference
(/root/reference/src/dex_retargeting/optimizer.py):

* ``PositionOptimizer.get_objective_function``   optimizer.py:138-200
* ``VectorOptimizer.get_objective_function``     optimizer.py:241-306
* ``DexPilotOptimizer.get_objective_function``   optimizer.py:456-577 (pre-amble :462-508, closure :510-577)
* mimic adaptor forward/backward               kinematics_adaptor.py:102-113
* joint index maps                              optimizer.py:27-40, 65-75

Quirks carried exactly: the returned VALUE omits the ``norm_delta`` term that the
GRADIENT includes (optimizer.py:194-198, 300-304, 571-575); SmoothL1 of a norm (vector/dexpilot) vs SmoothL1
per coordinate (position); ``torch.norm`` has zero sub-gradient at 0; the vector target is scaled in
float32 when it arrives as float32 (optimizer.py:246 with seq_retarget.py:116); DexPilot's reference vectors
are rounded to float32 (optimizer.py:507).

Pinned against the reference's OWN code by tests/golden/gen_golden.py, which imports
/root/reference/src/dex_retargeting/optimizer.py with stand-ins for pinocchio (oracle.kin) and nlopt and
stores (x, last, ref) -> (value, grad) vectors under tests/golden/.
"""
from __future__ import annotations

from typing import List, Optional, Sequence, Tuple

import numpy as np

from .kin import OracleRobot


def smooth_l1(d: np.ndarray, beta: float) -> Tuple[np.ndarray, np.ndarray]:
    """torch.nn.SmoothL1Loss(beta) element value and derivative for input d vs target 0."""
    ad = np.abs(d)
    val = np.where(ad < beta, 0.5 * d * d / beta, ad - 0.5 * beta)
    der = np.where(ad < beta, d / beta, np.sign(d))
    return val, der


def generate_link_indices(num_fingers: int):
    """optimizer.py:407-428."""
    origin, task = [], []
    for i in range(1, num_fingers):
        for j in range(i + 1, num_fingers + 1):
            origin.append(j)
            task.append(i)
    for i in range(1, num_fingers + 1):
        origin.append(0)
        task.append(i)
    return origin, task


def dexpilot_cache(num_fingers: int, eta1: float, eta2: float):
    """optimizer.py:430-454."""
    n_pair = num_fingers * (num_fingers - 1) // 2
    s2_o, s2_t = [], []
    for i in range(0, num_fingers - 2):
        for j in range(i + 1, num_fingers - 1):
            s2_o.append(j)
            s2_t.append(i)
    dist = np.array([eta1] * (num_fingers - 1) + [eta2] * ((num_fingers - 1) * (num_fingers - 2) // 2))
    return n_pair, s2_o, s2_t, dist


class OracleProblem:
    """One retargeting problem = robot + optimizer type + link/joint selection + constants."""

    def __init__(self, robot: OracleRobot, kind: str, target_joint_names: Optional[Sequence[str]] = None, *,
                 target_origin_link_names: Optional[Sequence[str]] = None,
                 target_task_link_names: Optional[Sequence[str]] = None,
                 target_link_names: Optional[Sequence[str]] = None,
                 wrist_link_name: Optional[str] = None,
                 finger_tip_link_names: Optional[Sequence[str]] = None,
                 huber_delta: Optional[float] = None, norm_delta: float = 4e-3, scaling: float = 1.0,
                 use_mimic: bool = True, project_dist: float = 0.03, escape_dist: float = 0.05,
                 eta1: float = 1e-4, eta2: float = 3e-2, has_joint_limits: bool = True):
        self.robot = robot
        self.kind = kind.lower()
        names = list(target_joint_names) if target_joint_names is not None else list(robot.dof_joint_names)
        self.target_joint_names = names
        self.idx_pin2target = np.array([robot.qidx[n] for n in names], dtype=int)  # optimizer.py:28-36
        fixed = [i for i in range(robot.dof) if i not in set(self.idx_pin2target.tolist())]
        self.mimic = list(robot.mimic) if use_mimic else []
        mimic_pin = [robot.qidx[m[0]] for m in self.mimic]
        self.idx_pin2fixed = np.array([i for i in fixed if i not in mimic_pin], dtype=int)  # optimizer.py:65-75
        self.n_opt = len(names)
        self.norm_delta = float(norm_delta)
        self.scaling = float(scaling)
        lim = robot.joint_limits[self.idx_pin2target]
        if not has_joint_limits:  # seq_retarget.py:24-30
            lim = np.stack([np.full(self.n_opt, -1e4), np.full(self.n_opt, 1e4)], 1)
        self.joint_limits = lim
        self.has_joint_limits = has_joint_limits

        if self.kind == "position":
            self.huber_delta = 0.02 if huber_delta is None else float(huber_delta)
            self.computed_links = list(target_link_names)
            self.n_ref = len(self.computed_links)
            self.ftol = 1e-5
        else:
            if self.kind == "dexpilot":
                self.huber_delta = 0.03 if huber_delta is None else float(huber_delta)
                self.num_fingers = len(finger_tip_link_names)
                o_idx, t_idx = generate_link_indices(self.num_fingers)
                link_names = [wrist_link_name] + list(finger_tip_link_names)
                origin_names = [link_names[i] for i in o_idx]
                task_names = [link_names[i] for i in t_idx]
                self.project_dist, self.escape_dist = project_dist, escape_dist
                self.n_pair, self.s2_origin, self.s2_task, self.projected_dist = dexpilot_cache(
                    self.num_fingers, eta1, eta2)
                self.target_link_human_indices = (np.stack([o_idx, t_idx], 0) * 4).astype(int)
            elif self.kind == "vector":
                self.huber_delta = 0.02 if huber_delta is None else float(huber_delta)
                origin_names, task_names = list(target_origin_link_names), list(target_task_link_names)
            else:
                raise ValueError(kind)
            comp = []
            for n in origin_names + task_names:  # deterministic stand-in for the reference's set() order (Q7)
                if n not in comp:
                    comp.append(n)
            self.computed_links = comp
            self.origin_idx = np.array([comp.index(n) for n in origin_names])
            self.task_idx = np.array([comp.index(n) for n in task_names])
            self.n_ref = len(origin_names)
            self.ftol = 1e-6

        # mimic fold tables (kinematics_adaptor.py:73-84)
        self.idx_pin2mimic = np.array([robot.qidx[m[0]] for m in self.mimic], dtype=int)
        self.idx_pin2source = np.array([robot.qidx[m[1]] for m in self.mimic], dtype=int)
        self.idx_target2source = np.array([names.index(m[1]) for m in self.mimic], dtype=int)
        self.multipliers = np.array([m[2] for m in self.mimic])
        self.offsets = np.array([m[3] for m in self.mimic])

    # ------------------------------------------------------------------------------------------
    @property
    def bounds(self) -> Tuple[np.ndarray, np.ndarray]:
        """nlopt box: optimizer.py:54-60 (epsilon 1e-3); unbounded when has_joint_limits is False."""
        if not self.has_joint_limits:
            return np.full(self.n_opt, -np.inf), np.full(self.n_opt, np.inf)
        return self.joint_limits[:, 0] - 1e-3, self.joint_limits[:, 1] + 1e-3

    def full_qpos(self, x: np.ndarray, fixed: Optional[np.ndarray] = None) -> np.ndarray:
        x = np.atleast_2d(np.asarray(x, dtype=np.float64))
        q = np.zeros((x.shape[0], self.robot.dof))
        if len(self.idx_pin2fixed):
            q[:, self.idx_pin2fixed] = np.asarray(fixed, dtype=np.float64).reshape(x.shape[0], -1)
        q[:, self.idx_pin2target] = x
        if len(self.mimic):  # kinematics_adaptor.py:102-105
            q[:, self.idx_pin2mimic] = q[:, self.idx_pin2source] * self.multipliers + self.offsets
        return q

    def _fold_jacobian(self, J: np.ndarray) -> np.ndarray:
        """(..., nq) -> (..., n_opt): optimizer.py:293-296 + kinematics_adaptor.py:107-113."""
        Jt = J[..., self.idx_pin2target].copy()
        if len(self.mimic):
            Jm = J[..., self.idx_pin2mimic] * self.multipliers
            for i, idx in enumerate(self.idx_target2source):
                Jt[..., idx] += Jm[..., i]
        return Jt

    # ------------------------------------------------------------------------------------------
    def dexpilot_preamble(self, target_vector: np.ndarray, projected: np.ndarray):
        """optimizer.py:462-508, batched.  target_vector (B,V,3) as the caller passed it (float32 when it
        comes through SeqRetargeting); projected (B,n_pair) bool state, updated copy returned.
        Returns weights (B,V) float64 (exact small integers), reference vectors (B,V,3) float32->float64."""
        tv = np.asarray(target_vector)
        B = tv.shape[0]
        n_pair, F = self.n_pair, self.num_fingers
        len_s2 = len(self.s2_task)
        len_s1 = n_pair - len_s2
        proj = np.array(projected, dtype=bool, copy=True).reshape(B, n_pair)
        dist = np.linalg.norm(tv[:, :n_pair], axis=2)  # dtype follows tv (float32 stays float32)
        s1 = proj[:, :len_s1]
        s1[dist[:, :len_s1] < self.project_dist] = True
        s1[dist[:, :len_s1] > self.escape_dist] = False
        proj[:, :len_s1] = s1
        s2 = np.logical_and(s1[:, self.s2_origin], s1[:, self.s2_task]) if len_s2 else np.zeros((B, 0), bool)
        s2 = np.logical_and(s2, dist[:, len_s1:n_pair] <= 0.03)
        proj[:, len_s1:] = s2
        high = np.array([200.0] * len_s1 + [400.0] * len_s2)
        weight = np.where(proj, high[None], 1.0)
        weight = np.concatenate([weight, np.full((B, F), float(n_pair + F))], axis=1)
        normal_vec = tv * self.scaling
        dir_vec = tv[:, :n_pair] / (dist[:, :, None] + 1e-6)
        projected_vec = dir_vec * self.projected_dist[None, :, None]
        ref = np.where(proj[:, :, None], projected_vec, normal_vec[:, :n_pair])
        ref = np.concatenate([ref, normal_vec[:, n_pair:]], axis=1)
        ref = ref.astype(np.float32).astype(np.float64)  # optimizer.py:507
        return weight, ref, proj

    # ------------------------------------------------------------------------------------------
    def evaluate(self, x, ref_value, fixed=None, last=None, *, weights=None, dexpilot_ref=None,
                 need_grad: bool = True):
        """Batched objective exactly as the reference's closures compute it.

        x (B,n_opt) f64; ref_value (B,n_ref,3) as passed by the caller (its dtype matters for the vector scaling);
        last (B,n_opt) = the float32-rounded last_qpos of optimizer.py:93.
        For dexpilot pass `weights`, `dexpilot_ref` from :meth:`dexpilot_preamble`.
        Returns (value (B,), grad (B,n_opt) or None, link positions (B,L,3)).
        value has NO norm_delta term; grad HAS it (quirk Q1)."""
        x = np.atleast_2d(np.asarray(x, dtype=np.float64))
        B = x.shape[0]
        q = self.full_qpos(x, fixed)
        pos = self.robot.link_positions(q, self.computed_links)
        beta = self.huber_delta
        if self.kind == "position":
            tgt = np.asarray(ref_value).astype(np.float64).reshape(B, -1, 3)
            val, der = smooth_l1(pos - tgt, beta)
            f = val.reshape(B, -1).mean(1)
            grad_pos = der / val.reshape(B, -1).shape[1]
        else:
            if self.kind == "vector":
                rv = np.asarray(ref_value)
                tgt = (rv * rv.dtype.type(self.scaling)).astype(np.float64).reshape(B, -1, 3)  # optimizer.py:246
                w = np.ones((B, self.n_ref))
            else:
                tgt = np.asarray(dexpilot_ref, dtype=np.float64).reshape(B, -1, 3)
                w = np.asarray(weights, dtype=np.float64).reshape(B, -1)
            vec = pos[:, self.task_idx] - pos[:, self.origin_idx]
            diff = vec - tgt
            d = np.linalg.norm(diff, axis=2)
            val, der = smooth_l1(d, beta)
            V = self.n_ref
            f = (val * w).sum(1) / V
            with np.errstate(invalid="ignore", divide="ignore"):
                unit = np.where(d[..., None] > 0, diff / d[..., None], 0.0)  # torch.norm backward: 0 at 0
            gvec = unit * (der * w / V)[..., None]
            grad_pos = np.zeros_like(pos)
            np.add.at(grad_pos, (slice(None), self.task_idx), gvec)
            np.add.at(grad_pos, (slice(None), self.origin_idx), -gvec)
        if not need_grad:
            return f, None, pos
        J = self._fold_jacobian(self.robot.point_jacobians(q, self.computed_links))  # (B,L,3,n_opt)
        grad = np.einsum("blc,blcn->bn", grad_pos, J)
        if last is not None:
            grad = grad + 2 * self.norm_delta * (x - np.asarray(last, dtype=np.float64).reshape(B, -1))
        return f, grad, pos

    def total(self, x, ref_value, fixed=None, last=None, **kw) -> np.ndarray:
        """F(x) = f(x) + norm_delta * ||x - last||^2 : the function whose gradient the reference hands to SLSQP."""
        f, _, _ = self.evaluate(x, ref_value, fixed, last, need_grad=False, **kw)
        x = np.atleast_2d(np.asarray(x, dtype=np.float64))
        return f + self.norm_delta * ((x - np.asarray(last, dtype=np.float64).reshape(x.shape)) ** 2).sum(1)
