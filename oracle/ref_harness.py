"""ORACLE (test infrastructure only -- never imported by the product path).

Runs the REFERENCE'S OWN optimizer code (/root/reference/src/dex_retargeting/optimizer.py,
kinematics_adaptor.py, seq_retarget.py, optimizer_utils.py), imported from where it lies -- nothing is
copied -- with stand-ins for its two native dependencies that are not installed in this container:

* ``pinocchio``  -> :class:`FakeRobotWrapper`, a duck-typed ``RobotWrapper`` backed by ``oracle.kin``
  (same public surface as /root/reference/src/dex_retargeting/robot_wrapper.py:28-95);
* ``nlopt``      -> :class:`_NloptOptStandIn`, ``nlopt.opt(LD_SLSQP)`` emulated with scipy's SLSQP (the same
  Kraft routine nlopt wraps; the stop rule is scipy's ``ftol``, so solver outputs are "reference-as-configured
  stand-in", not bit-equal nlopt).
* ``pytransform3d.rotations`` -> two small functions used only by ``SeqRetargeting.warm_start``.

Only usable where /root/reference exists (this build container).  The GPU box never imports this file:
tests/golden/gen_golden.py uses it to write the committed fixtures.
"""
from __future__ import annotations

import os
import sys
import types
from typing import List

import numpy as np

from .kin import OracleRobot

REFERENCE_SRC = "/root/reference/src"


def reference_available() -> bool:
    return os.path.isdir(os.path.join(REFERENCE_SRC, "dex_retargeting"))


class _NloptOptStandIn:
    def __init__(self, algorithm, n):
        self.n = n
        self.lb = np.full(n, -np.inf)
        self.ub = np.full(n, np.inf)
        self.ftol_abs = 1e-6
        self.fn = None
        self._last = float("nan")
        self.n_evals = 0

    def set_lower_bounds(self, lb):
        self.lb = np.asarray(lb, dtype=np.float64)

    def set_upper_bounds(self, ub):
        self.ub = np.asarray(ub, dtype=np.float64)

    def set_ftol_abs(self, v):
        self.ftol_abs = float(v)

    def set_min_objective(self, fn):
        self.fn = fn

    def last_optimum_value(self):
        return self._last

    def optimize(self, x0):
        from scipy.optimize import minimize

        def fun(x):
            g = np.zeros(self.n)
            f = self.fn(np.asarray(x, dtype=np.float64), g)
            self.n_evals += 1
            return float(f), g

        x0 = np.clip(np.asarray(x0, dtype=np.float64), self.lb, self.ub)
        res = minimize(fun, x0, jac=True, method="SLSQP", bounds=list(zip(self.lb, self.ub)),
                       options=dict(ftol=self.ftol_abs, maxiter=200))
        self._last = float(res.fun)
        return res.x


def _install_stubs():
    if "nlopt" not in sys.modules:
        m = types.ModuleType("nlopt")
        m.LD_SLSQP = 40
        m.opt = _NloptOptStandIn
        sys.modules["nlopt"] = m
    if "pinocchio" not in sys.modules:
        sys.modules["pinocchio"] = types.ModuleType("pinocchio")
    if "pytransform3d" not in sys.modules:
        pt = types.ModuleType("pytransform3d")
        rot = types.ModuleType("pytransform3d.rotations")

        def matrix_from_quaternion(q):  # (w, x, y, z)
            w, x, y, z = np.asarray(q, dtype=np.float64) / np.linalg.norm(q)
            return np.array([[1 - 2 * (y * y + z * z), 2 * (x * y - z * w), 2 * (x * z + y * w)],
                             [2 * (x * y + z * w), 1 - 2 * (x * x + z * z), 2 * (y * z - x * w)],
                             [2 * (x * z - y * w), 2 * (y * z + x * w), 1 - 2 * (x * x + y * y)]])

        def euler_from_matrix(R, i, j, k, extrinsic):
            # only the (0,1,2, extrinsic=False) case used by seq_retarget.py:96-98: R = Rx(a) Ry(b) Rz(c)
            assert (i, j, k, extrinsic) == (0, 1, 2, False)
            b = np.arcsin(np.clip(R[0, 2], -1, 1))
            a = np.arctan2(-R[1, 2], R[2, 2])
            c = np.arctan2(-R[0, 1], R[0, 0])
            return np.array([a, b, c])

        rot.matrix_from_quaternion = matrix_from_quaternion
        rot.euler_from_matrix = euler_from_matrix
        pt.rotations = rot
        sys.modules["pytransform3d"] = pt
        sys.modules["pytransform3d.rotations"] = rot
    if REFERENCE_SRC not in sys.path:
        sys.path.insert(0, REFERENCE_SRC)


def import_reference():
    """Returns the reference's (optimizer, kinematics_adaptor, seq_retarget, optimizer_utils) modules."""
    if not reference_available():
        raise RuntimeError("/root/reference is not present on this machine")
    _install_stubs()
    import dex_retargeting.kinematics_adaptor as ka  # noqa: E402
    import dex_retargeting.optimizer as opt  # noqa: E402
    import dex_retargeting.optimizer_utils as ou  # noqa: E402
    import dex_retargeting.seq_retarget as sr  # noqa: E402

    # (tests/reference_suite aliases `dex_retargeting` to the drop-in inside its own pytest process; a checker that
    # then imported "the reference" would silently compare the drop-in with itself -- refuse)
    for mod in (opt, ka, sr, ou):
        where = os.path.realpath(getattr(mod, "__file__", "") or "")
        if not where.startswith("/root/reference/"):
            raise RuntimeError(f"`{mod.__name__}` resolves to {where}, not to /root/reference: `dex_retargeting` is aliased in "
                               f"this process (tests/reference_suite); import the reference in a process of its own")
    return opt, ka, sr, ou


def import_reference_detector():
    """The reference's MediaPipe wrapper module (example/vector_retargeting/single_hand_detector.py), imported
    from where it lies with empty stand-ins for the ``mediapipe`` package (not installed here); only its static
    ``SingleHandDetector.estimate_frame_from_hand_points`` and the OPERATOR2MANO constants are usable."""
    import importlib.util

    path = "/root/reference/example/vector_retargeting/single_hand_detector.py"
    if not os.path.isfile(path):
        raise RuntimeError("/root/reference is not present on this machine")
    names = ["mediapipe", "mediapipe.framework", "mediapipe.framework.formats",
             "mediapipe.framework.formats.landmark_pb2", "mediapipe.python", "mediapipe.python.solutions",
             "mediapipe.python.solutions.hands_connections", "mediapipe.python.solutions.drawing_utils",
             "mediapipe.python.solutions.hands"]
    for n in names:
        if n not in sys.modules:
            sys.modules[n] = types.ModuleType(n)
    for n in names[1:]:
        parent, _, leaf = n.rpartition(".")
        setattr(sys.modules[parent], leaf, sys.modules[n])
    pb = sys.modules["mediapipe.framework.formats.landmark_pb2"]
    pb.LandmarkList = pb.NormalizedLandmarkList = object
    sys.modules["mediapipe.python.solutions.drawing_utils"].DrawingSpec = object
    sys.modules["mediapipe.python.solutions.hands"].HandLandmark = object
    spec = importlib.util.spec_from_file_location("_ref_single_hand_detector", path)
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


class _FakeModel:
    def __init__(self, nq):
        self.nq = nq
        self.nv = nq


class FakeRobotWrapper:
    """Duck-typed stand-in for the reference's RobotWrapper (robot_wrapper.py:8-95) on top of oracle.kin."""

    def __init__(self, urdf_path: str, add_dummy_free_joints: bool = False):
        self.kin = OracleRobot(urdf_path, add_dummy_free_joints)
        self.model = _FakeModel(self.kin.dof)
        self._q = np.zeros(self.kin.dof)
        self.q0 = np.zeros(self.kin.dof)
        self._links = list(self.kin.links)

    @property
    def joint_names(self) -> List[str]:
        return ["universe"] + self.kin.dof_joint_names

    @property
    def dof_joint_names(self) -> List[str]:
        return list(self.kin.dof_joint_names)

    @property
    def dof(self) -> int:
        return self.kin.dof

    @property
    def link_names(self) -> List[str]:
        return ["universe"] + self._links + [j.name for j in self.kin.joints]

    @property
    def joint_limits(self):
        return self.kin.joint_limits.copy()

    def get_joint_index(self, name: str):
        return self.dof_joint_names.index(name)

    def get_link_index(self, name: str):
        if name not in self.link_names:
            raise ValueError(f"{name} is not a link name. Valid link names: \n{self.link_names}")
        return self._links.index(name)

    def get_joint_parent_child_frames(self, joint_name: str):
        """robot_wrapper.py:67-77 in this stand-in's own link numbering (ids index ``self._links``); only the child id is
        used by the reference (seq_retarget.py:83)."""
        j = [jj for jj in self.kin.joints if jj.name == joint_name][0]
        return self._links.index(j.parent), self._links.index(j.child)

    def compute_forward_kinematics(self, qpos):
        self._q = np.array(qpos, dtype=np.float64)

    def get_link_pose(self, link_id: int):
        R, p = self.kin.link_poses(self._q[None], [self._links[link_id]])
        T = np.eye(4)
        T[:3, :3], T[:3, 3] = R[0, 0], p[0, 0]
        return T

    def get_link_pose_inv(self, link_id: int):
        return np.linalg.inv(self.get_link_pose(link_id))

    def compute_single_link_local_jacobian(self, qpos, link_id: int):
        return self.kin.frame_jacobian_local(np.asarray(qpos, dtype=np.float64), self._links[link_id])
