"""Minimal URDF reader + pinocchio-compatible kinematic model (host plumbing, cold path).

Replaces, for the retargeting hot path only, what the reference gets from
``yourdfpy.URDF.load`` + ``pin.buildModelFromUrdf``:

* URDF conventions follow the reference's vendored parser:
  origin rpy is fixed-axis XYZ, ``R = Rz(yaw) Ry(pitch) Rx(roll)``
  (/root/reference/src/dex_retargeting/yourdfpy.py:1375-1387); missing axis
  defaults to ``1 0 0`` (:1631-1643); mimic multiplier/offset default 1/0
  (:1107-1115); ``add_dummy_free_joints`` prepends 3 prismatic (+-5 m) and
  3 revolute (+-2 pi) joints about x,y,z (:1942-1989).
* Model conventions follow what ``RobotWrapper`` observes through pinocchio
  (/root/reference/src/dex_retargeting/robot_wrapper.py:13-95): one 1-DoF joint
  per revolute/prismatic URDF joint, fixed joints folded into their parent
  joint, every link exposed as a BODY frame with a constant placement, dof
  order = depth-first walk from the root with siblings visited in
  lexicographic joint-name order (urdfdom keeps joints in a std::map).
  ``continuous``/``floating``/``planar`` joints are rejected like the
  reference rejects ``nq != nv`` (robot_wrapper.py:22-23).

Nothing here runs per frame; the per-frame arithmetic lives in csrc/ (HIP).
"""
from __future__ import annotations

import xml.etree.ElementTree as ET
from dataclasses import dataclass, field
from typing import Dict, List, Optional, Tuple

import numpy as np

DUMMY_JOINT_NAMES = [f"dummy_{n}_translation_joint" for n in "xyz"] + [
    f"dummy_{n}_rotation_joint" for n in "xyz"
]


def rpy_to_matrix(rpy) -> np.ndarray:
    """Fixed-axis (extrinsic) XYZ euler -> rotation matrix: R = Rz(y) Ry(p) Rx(r)."""
    r, p, y = (float(v) for v in rpy)
    cr, sr, cp, sp, cy, sy = np.cos(r), np.sin(r), np.cos(p), np.sin(p), np.cos(y), np.sin(y)
    rx = np.array([[1, 0, 0], [0, cr, -sr], [0, sr, cr]])
    ry = np.array([[cp, 0, sp], [0, 1, 0], [-sp, 0, cp]])
    rz = np.array([[cy, -sy, 0], [sy, cy, 0], [0, 0, 1]])
    return rz @ ry @ rx


def make_transform(xyz, rpy) -> np.ndarray:
    t = np.eye(4)
    t[:3, :3] = rpy_to_matrix(rpy)
    t[:3, 3] = np.asarray(xyz, dtype=np.float64)
    return t


@dataclass
class UrdfMimic:
    joint: str
    multiplier: float = 1.0
    offset: float = 0.0


@dataclass
class UrdfJoint:
    name: str
    type: str
    parent: str
    child: str
    origin: np.ndarray  # 4x4
    axis: np.ndarray  # (3,)
    lower: Optional[float] = None
    upper: Optional[float] = None
    mimic: Optional[UrdfMimic] = None


@dataclass
class UrdfRobot:
    name: str
    links: List[str] = field(default_factory=list)
    joints: List[UrdfJoint] = field(default_factory=list)

    @property
    def joint_map(self) -> Dict[str, UrdfJoint]:
        return {j.name: j for j in self.joints}

    @property
    def root_link(self) -> str:
        names = list(self.links)
        for j in self.joints:
            if j.child in names:
                names.remove(j.child)
        if not names:
            raise RuntimeError("No root link found for robot.")
        return names[0]


def _floats(s: str, n: int) -> List[float]:
    out = []
    for tok in s.split():
        try:
            out.append(float(tok))
        except ValueError:
            out.append(0.0)
    if len(out) != n:
        raise ValueError(f"expected {n} numbers, got {s!r}")
    return out


def parse_urdf(path: str, add_dummy_free_joints: bool = False) -> UrdfRobot:
    root = ET.parse(path).getroot()
    if root.tag != "robot":
        raise ValueError(f"{path}: root element is <{root.tag}>, expected <robot>")
    robot = UrdfRobot(name=root.attrib.get("name", "robot"))
    for link in root.findall("link"):
        robot.links.append(link.attrib["name"])
    allowed = ["revolute", "continuous", "prismatic", "fixed", "floating", "planar"]
    for je in root.findall("joint"):
        jtype = je.attrib.get("type")
        if jtype not in allowed:
            raise ValueError(f"joint {je.attrib.get('name')}: type {jtype!r} not in {allowed}")
        pe, ce = je.find("parent"), je.find("child")
        if pe is None or ce is None:
            raise ValueError(f"joint {je.attrib.get('name')}: missing <parent>/<child>")
        oe = je.find("origin")
        xyz = _floats(oe.attrib.get("xyz", "0 0 0"), 3) if oe is not None else [0, 0, 0]
        rpy = _floats(oe.attrib.get("rpy", "0 0 0"), 3) if oe is not None else [0, 0, 0]
        ae = je.find("axis")
        axis = np.array(_floats(ae.attrib.get("xyz", "1 0 0"), 3)) if ae is not None else np.array([1.0, 0, 0])
        le = je.find("limit")
        lower = upper = None
        if le is not None:
            lower = float(le.attrib["lower"]) if "lower" in le.attrib else None
            upper = float(le.attrib["upper"]) if "upper" in le.attrib else None
        me = je.find("mimic")
        mimic = None
        if me is not None:
            mimic = UrdfMimic(
                joint=me.attrib["joint"],
                multiplier=float(me.attrib.get("multiplier", 1.0)),
                offset=float(me.attrib.get("offset", 0.0)),
            )
        robot.joints.append(
            UrdfJoint(je.attrib["name"], jtype, pe.attrib["link"], ce.attrib["link"],
                      make_transform(xyz, rpy), axis, lower, upper, mimic)
        )
    if add_dummy_free_joints:
        _add_dummy_joints(robot, robot.root_link)
    return robot


def _add_dummy_joints(robot: UrdfRobot, root_link_name: str) -> None:
    link_names = [f"dummy_{n}_translation_link" for n in "xyz"] + [f"dummy_{n}_rotation_link" for n in "xyz"]
    limits = [(-5.0, 5.0)] * 3 + [(-2 * np.pi, 2 * np.pi)] * 3
    types = ["prismatic"] * 3 + ["revolute"] * 3
    joints = []
    for i in range(6):
        axis = np.zeros(3)
        axis[i % 3] = 1.0
        child = link_names[i + 1] if i < 5 else root_link_name
        joints.append(UrdfJoint(DUMMY_JOINT_NAMES[i], types[i], link_names[i], child, np.eye(4), axis,
                                limits[i][0], limits[i][1], None))
    robot.joints = joints + robot.joints
    robot.links = link_names + robot.links


@dataclass
class DofJoint:
    name: str
    type: str  # "revolute" | "prismatic"
    parent: int  # index of parent dof joint, -1 = universe
    placement: np.ndarray  # 4x4, pose of this joint frame in the parent dof-joint frame (fixed joints folded)
    axis: np.ndarray  # unit axis in the joint frame
    lower: float
    upper: float


@dataclass
class BodyFrame:
    name: str
    parent: int  # dof joint index, -1 = universe
    placement: np.ndarray  # 4x4 in the parent joint frame


class KinematicModel:
    """Flattened kinematic tree with pinocchio's joint/frame conventions (see module docstring)."""

    def __init__(self, robot: UrdfRobot):
        self.urdf = robot
        children: Dict[str, List[UrdfJoint]] = {}
        for j in robot.joints:
            children.setdefault(j.parent, []).append(j)
        for lst in children.values():
            lst.sort(key=lambda jj: jj.name)  # urdfdom std::map order

        self.joints: List[DofJoint] = []
        self.frames: List[BodyFrame] = []
        self.frame_names: List[str] = ["universe"]  # every frame name pinocchio would expose (joints + links)
        # pinocchio-style frame table, same order as frame_names: (kind, parent dof joint or -1, previous frame id,
        # index into self.frames for BODY entries or -1).  Frame ids handed out by RobotWrapper index THIS list,
        # like pin.Model.getFrameId (robot_wrapper.py:57-77).
        self.frame_table: List[Tuple[str, int, int, int]] = [("FIXED_JOINT", -1, 0, -1)]

        root = robot.root_link
        self.frames.append(BodyFrame(root, -1, np.eye(4)))
        self.frame_names.append(root)
        self.frame_table.append(("BODY", -1, 0, 0))

        def walk(link: str, parent_joint: int, link_in_joint: np.ndarray):
            link_fid = max(i for i, n in enumerate(self.frame_names) if n == link and self.frame_table[i][0] == "BODY")
            for uj in children.get(link, []):
                self.frame_names.append(uj.name)
                joint_fid = len(self.frame_names) - 1
                if uj.type in ("revolute", "prismatic"):
                    if uj.lower is None or uj.upper is None:
                        raise ValueError(f"joint {uj.name}: revolute/prismatic joints need lower/upper limits")
                    axis = np.asarray(uj.axis, dtype=np.float64)
                    nrm = np.linalg.norm(axis)
                    if nrm < 1e-12:
                        raise ValueError(f"joint {uj.name}: zero axis")
                    idx = len(self.joints)
                    self.joints.append(DofJoint(uj.name, uj.type, parent_joint, link_in_joint @ uj.origin,
                                                axis / nrm, float(uj.lower), float(uj.upper)))
                    self.frame_table.append(("JOINT", idx, link_fid, -1))
                    self.frames.append(BodyFrame(uj.child, idx, np.eye(4)))
                    self.frame_names.append(uj.child)
                    self.frame_table.append(("BODY", idx, joint_fid, len(self.frames) - 1))
                    walk(uj.child, idx, np.eye(4))
                elif uj.type == "fixed":
                    child_in_joint = link_in_joint @ uj.origin
                    self.frame_table.append(("FIXED_JOINT", parent_joint, link_fid, -1))
                    self.frames.append(BodyFrame(uj.child, parent_joint, child_in_joint))
                    self.frame_names.append(uj.child)
                    self.frame_table.append(("BODY", parent_joint, joint_fid, len(self.frames) - 1))
                    walk(uj.child, parent_joint, child_in_joint)
                else:
                    # continuous -> nq=2 != nv=1 in pinocchio, rejected by the reference (robot_wrapper.py:22-23)
                    raise NotImplementedError("Can not handle robot with special joint.")

        walk(root, -1, np.eye(4))
        self._frame_index = {f.name: i for i, f in enumerate(self.frames)}
        self._body_fid = {self.frame_names[i]: i for i, e in enumerate(self.frame_table) if e[0] == "BODY"}

    # ---- metadata mirroring RobotWrapper's properties -------------------------------------------
    @property
    def dof(self) -> int:
        return len(self.joints)

    @property
    def dof_joint_names(self) -> List[str]:
        return [j.name for j in self.joints]

    @property
    def link_names(self) -> List[str]:
        return list(self.frame_names)

    @property
    def joint_limits(self) -> np.ndarray:
        return np.array([[j.lower, j.upper] for j in self.joints], dtype=np.float64).reshape(-1, 2)

    def body_frame_index(self, name: str) -> int:
        if name not in self._frame_index:
            raise ValueError(f"{name} is not a link name. Valid link names: \n{self.link_names}")
        return self._frame_index[name]

    def body_frame_id(self, name: str) -> int:
        """pinocchio-style frame id of a link (== model.getFrameId(name, pin.BODY), robot_wrapper.py:57-65)."""
        if name not in self._body_fid:
            raise ValueError(f"{name} is not a link name. Valid link names: \n{self.link_names}")
        return self._body_fid[name]

    def body_of_frame_id(self, fid: int) -> int:
        """index into self.frames of the BODY frame with pinocchio-style id `fid`."""
        if not (0 <= fid < len(self.frame_table)) or self.frame_table[fid][0] != "BODY":
            raise ValueError(f"frame id {fid} is not a BODY frame")
        return self.frame_table[fid][3]

    def frame_pose_and_local_jacobian(self, q: np.ndarray, body: int) -> Tuple[np.ndarray, np.ndarray]:
        """Host float64 (cold path): 4x4 world pose of body frame `body` (index into self.frames) at configuration q
        and its 6 x dof Jacobian expressed in the frame's own (LOCAL) axes, linear rows first -- what
        ``pin.computeFrameJacobian`` returns by default (robot_wrapper.py:93-95)."""
        q = np.asarray(q, dtype=np.float64).reshape(-1)
        f = self.frames[body]
        T = np.eye(4)
        info = []  # (dof idx, type, world axis, world origin)
        for j in (self.ancestors(f.parent) if f.parent >= 0 else []):
            jt = self.joints[j]
            T = T @ jt.placement
            a_w = T[:3, :3] @ jt.axis
            info.append((j, jt.type, a_w, T[:3, 3].copy()))
            M = np.eye(4)
            if jt.type == "revolute":
                a, th = jt.axis, q[j]
                K = np.array([[0, -a[2], a[1]], [a[2], 0, -a[0]], [-a[1], a[0], 0]])
                M[:3, :3] = np.eye(3) + np.sin(th) * K + (1 - np.cos(th)) * (K @ K)
            else:
                M[:3, 3] = jt.axis * q[j]
            T = T @ M
        T = T @ f.placement
        R, p = T[:3, :3], T[:3, 3]
        J = np.zeros((6, self.dof))
        for j, typ, a_w, o_w in info:
            if typ == "revolute":
                J[:3, j] = R.T @ np.cross(a_w, p - o_w)
                J[3:, j] = R.T @ a_w
            else:
                J[:3, j] = R.T @ a_w
        return T, J

    def mimic_joints(self) -> Tuple[List[str], List[str], List[float], List[float]]:
        """(source names, mimic names, multipliers, offsets) in URDF joint order
        (mirrors parse_mimic_joint, /root/reference/src/dex_retargeting/retargeting_config.py:265-285)."""
        src, mim, mul, off = [], [], [], []
        for j in self.urdf.joints:
            if j.mimic is not None:
                mim.append(j.name)
                src.append(j.mimic.joint)
                mul.append(j.mimic.multiplier)
                off.append(j.mimic.offset)
        return src, mim, mul, off

    def ancestors(self, joint: int) -> List[int]:
        """dof-joint indices from the root down to (and including) `joint`."""
        chain = []
        while joint >= 0:
            chain.append(joint)
            joint = self.joints[joint].parent
        return chain[::-1]
