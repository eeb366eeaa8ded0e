"""ORACLE (test infrastructure only -- never imported by the product path).

CPU float64 restatement of the kinematics the reference obtains from pinocchio
through ``RobotWrapper`` (/root/reference/src/dex_retargeting/robot_wrapper.py):

* ``compute_forward_kinematics`` + ``get_link_pose``  (robot_wrapper.py:82-87)
  -> :meth:`OracleRobot.link_poses`
* ``compute_single_link_local_jacobian``              (robot_wrapper.py:93-95)
  -> :meth:`OracleRobot.frame_jacobian_local` (6 x nq, LOCAL frame, linear rows first)
* the world-aligned point Jacobian ``R_link @ J_local[:3]`` the optimizers build
  (/root/reference/src/dex_retargeting/optimizer.py:279-284)
  -> :meth:`OracleRobot.point_jacobians`

pinocchio itself is a third-party dependency absent from /root/reference
(pyproject.toml:28-36, ``pin>=3.3.1``) and not installed here, so this file
restates its *published* conventions: one 1-DoF joint per revolute/prismatic
URDF joint; world pose of a link = product over the chain of
``origin(xyz,rpy) * motion(axis, q)``; dof order = depth-first from the root,
siblings in lexicographic joint-name order.  PARITY UNPINNED at the pinocchio
boundary (no golden FK vectors exist in the reference); correctness here is
pinned by finite differences and hand-derived poses in tests/test_oracle.py.

This module deliberately shares NO code with dex_retargeting_amd/urdf.py or the
table compiler: it walks the raw URDF tree (no fixed-joint folding, no axis
re-alignment), so that it can catch mistakes in those transformations.
All functions are batched over a leading dimension B.
"""
from __future__ import annotations

import xml.etree.ElementTree as ET
from typing import Dict, List, Optional, Sequence, Tuple

import numpy as np

_DUMMY = [("dummy_x_translation_joint", "prismatic", 0), ("dummy_y_translation_joint", "prismatic", 1),
          ("dummy_z_translation_joint", "prismatic", 2), ("dummy_x_rotation_joint", "revolute", 0),
          ("dummy_y_rotation_joint", "revolute", 1), ("dummy_z_rotation_joint", "revolute", 2)]


def _rot_from_rpy(r: float, p: float, y: float) -> np.ndarray:
    # URDF fixed-axis roll/pitch/yaw (yourdfpy.py:1382-1387: matrix_from_euler(.., 0, 1, 2, extrinsic=True))
    ca, sa, cb, sb, cg, sg = np.cos(y), np.sin(y), np.cos(p), np.sin(p), np.cos(r), np.sin(r)
    return np.array([
        [ca * cb, ca * sb * sg - sa * cg, ca * sb * cg + sa * sg],
        [sa * cb, sa * sb * sg + ca * cg, sa * sb * cg - ca * sg],
        [-sb, cb * sg, cb * cg],
    ])


def _skew(a: np.ndarray) -> np.ndarray:
    return np.array([[0, -a[2], a[1]], [a[2], 0, -a[0]], [-a[1], a[0], 0]])


class _J:
    __slots__ = ("name", "type", "parent", "child", "R0", "p0", "axis", "lo", "hi", "mimic")


class OracleRobot:
    def __init__(self, urdf_path: str, add_dummy_free_joints: bool = False):
        root = ET.parse(urdf_path).getroot()
        links = [l.attrib["name"] for l in root.findall("link")]
        joints: List[_J] = []
        for e in root.findall("joint"):
            j = _J()
            j.name, j.type = e.attrib["name"], e.attrib["type"]
            j.parent, j.child = e.find("parent").attrib["link"], e.find("child").attrib["link"]
            o = e.find("origin")
            xyz = [float(v) for v in (o.attrib.get("xyz", "0 0 0") if o is not None else "0 0 0").split()]
            rpy = [float(v) for v in (o.attrib.get("rpy", "0 0 0") if o is not None else "0 0 0").split()]
            j.R0, j.p0 = _rot_from_rpy(*rpy), np.array(xyz)
            a = e.find("axis")
            ax = np.array([float(v) for v in (a.attrib.get("xyz", "1 0 0") if a is not None else "1 0 0").split()])
            j.axis = ax / (np.linalg.norm(ax) if np.linalg.norm(ax) > 0 else 1.0)
            l = e.find("limit")
            j.lo = float(l.attrib.get("lower", "nan")) if l is not None else float("nan")
            j.hi = float(l.attrib.get("upper", "nan")) if l is not None else float("nan")
            m = e.find("mimic")
            j.mimic = None if m is None else (m.attrib["joint"], float(m.attrib.get("multiplier", 1.0)),
                                             float(m.attrib.get("offset", 0.0)))
            joints.append(j)

        child_links = {j.child for j in joints}
        root_link = [l for l in links if l not in child_links][0]
        if add_dummy_free_joints:  # yourdfpy.py:1942-1984
            names = [f"dummy_{n}_translation_link" for n in "xyz"] + [f"dummy_{n}_rotation_link" for n in "xyz"]
            dj = []
            for i, (jn, jt, k) in enumerate(_DUMMY):
                j = _J()
                j.name, j.type, j.parent = jn, jt, names[i]
                j.child = names[i + 1] if i < 5 else root_link
                j.R0, j.p0, j.axis = np.eye(3), np.zeros(3), np.eye(3)[k]
                j.lo, j.hi = (-5.0, 5.0) if jt == "prismatic" else (-2 * np.pi, 2 * np.pi)
                j.mimic = None
                dj.append(j)
            joints = dj + joints
            links = names + links
            root_link = names[0]

        self.root_link = root_link
        self.links = links
        self.joints = joints
        self.by_parent: Dict[str, List[_J]] = {}
        for j in joints:
            self.by_parent.setdefault(j.parent, []).append(j)
        for v in self.by_parent.values():
            v.sort(key=lambda jj: jj.name)
        self.parent_joint_of_link: Dict[str, _J] = {j.child: j for j in joints}

        # pinocchio dof order
        self.dof_joint_names: List[str] = []

        def visit(link):
            for j in self.by_parent.get(link, []):
                if j.type in ("revolute", "prismatic"):
                    self.dof_joint_names.append(j.name)
                elif j.type != "fixed":
                    raise NotImplementedError("Can not handle robot with special joint.")
                visit(j.child)

        visit(root_link)
        self.dof = len(self.dof_joint_names)
        self.qidx = {n: i for i, n in enumerate(self.dof_joint_names)}
        jm = {j.name: j for j in joints}
        self.joint_limits = np.array([[jm[n].lo, jm[n].hi] for n in self.dof_joint_names]).reshape(-1, 2)
        self.mimic = [(j.name, *j.mimic) for j in joints if j.mimic is not None]  # (mimic, source, mult, off)

    # ------------------------------------------------------------------------------------------
    def _chain(self, link: str) -> List[_J]:
        chain = []
        while link in self.parent_joint_of_link:
            j = self.parent_joint_of_link[link]
            chain.append(j)
            link = j.parent
        return chain[::-1]

    def _walk(self, q: np.ndarray, link: str):
        """World pose of `link` plus, per movable ancestor joint, (dof index, type, world axis, world origin)."""
        B = q.shape[0]
        R = np.broadcast_to(np.eye(3), (B, 3, 3)).copy()
        p = np.zeros((B, 3))
        info = []
        for j in self._chain(link):
            p = p + np.einsum("bij,j->bi", R, j.p0)
            R = R @ j.R0
            if j.type == "fixed":
                continue
            qi = self.qidx[j.name]
            a_w = np.einsum("bij,j->bi", R, j.axis)
            info.append((qi, j.type, a_w, p.copy()))
            if j.type == "revolute":
                th = q[:, qi]
                K = _skew(j.axis)
                Rq = (np.eye(3)[None] + np.sin(th)[:, None, None] * K[None]
                      + (1 - np.cos(th))[:, None, None] * (K @ K)[None])  # Rodrigues
                R = R @ Rq
            else:
                p = p + a_w * q[:, qi:qi + 1]
        return R, p, info

    def link_poses(self, q: np.ndarray, link_names: Sequence[str]) -> Tuple[np.ndarray, np.ndarray]:
        """q (B,nq) -> rotations (B,L,3,3), positions (B,L,3) of the named links in the world frame."""
        q = np.atleast_2d(np.asarray(q, dtype=np.float64))
        Rs, ps = [], []
        for name in link_names:
            R, p, _ = self._walk(q, name)
            Rs.append(R)
            ps.append(p)
        return np.stack(Rs, 1), np.stack(ps, 1)

    def link_positions(self, q: np.ndarray, link_names: Sequence[str]) -> np.ndarray:
        return self.link_poses(q, link_names)[1]

    def point_jacobians(self, q: np.ndarray, link_names: Sequence[str]) -> np.ndarray:
        """(B, L, 3, nq) world-aligned Jacobian of each link origin: d p_l / d q  (== R_l @ J_local[:3])."""
        q = np.atleast_2d(np.asarray(q, dtype=np.float64))
        B = q.shape[0]
        out = np.zeros((B, len(link_names), 3, self.dof))
        for li, name in enumerate(link_names):
            _, p, info = self._walk(q, name)
            for qi, typ, a_w, o_w in info:
                out[:, li, :, qi] = np.cross(a_w, p - o_w) if typ == "revolute" else a_w
        return out

    def point_hessian_contraction(self, q: np.ndarray, link_names: Sequence[str], force: np.ndarray) -> np.ndarray:
        """(B, nq, nq): sum_l force[b,l,:] . d^2 p_l / dq_j dq_k  (second-order kinematic term of a Newton
        Hessian).  For j an ancestor-or-self of k on the chain of l: revolute j -> a_j x (dp_l/dq_k); prismatic j -> 0."""
        q = np.atleast_2d(np.asarray(q, dtype=np.float64))
        B = q.shape[0]
        out = np.zeros((B, self.dof, self.dof))
        for li, name in enumerate(link_names):
            _, p, info = self._walk(q, name)
            f = force[:, li]
            cols = [np.cross(a_w, p - o_w) if typ == "revolute" else a_w for (_, typ, a_w, o_w) in info]
            for jj, (qj, tj, aj, oj) in enumerate(info):
                if tj != "revolute":
                    continue
                for kk in range(jj, len(info)):
                    qk = info[kk][0]
                    v = np.einsum("bi,bi->b", f, np.cross(aj, cols[kk]))
                    out[:, qj, qk] += v
                    if qk != qj:
                        out[:, qk, qj] += v
        return out

    def frame_jacobian_local(self, q: np.ndarray, link_name: str) -> np.ndarray:
        """(6, nq) LOCAL-frame Jacobian of one link for a single configuration (pinocchio computeFrameJacobian
        default reference frame; rows 0-2 linear, 3-5 angular), as consumed by optimizer.py:279-284."""
        q = np.asarray(q, dtype=np.float64).reshape(1, -1)
        R, p, info = self._walk(q, link_name)
        J = np.zeros((6, self.dof))
        for qi, typ, a_w, o_w in info:
            if typ == "revolute":
                J[:3, qi] = R[0].T @ np.cross(a_w[0], p[0] - o_w[0])
                J[3:, qi] = R[0].T @ a_w[0]
            else:
                J[:3, qi] = R[0].T @ a_w[0]
        return J

    def mimic_forward(self, q: np.ndarray) -> np.ndarray:
        """kinematics_adaptor.py:102-105, batched."""
        q = np.array(q, dtype=np.float64, copy=True)
        for mim, src, mul, off in self.mimic:
            q[..., self.qidx[mim]] = q[..., self.qidx[src]] * mul + off
        return q
