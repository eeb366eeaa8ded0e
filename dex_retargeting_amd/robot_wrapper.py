"""RobotWrapper: the reference's pinocchio façade (/root/reference/src/dex_retargeting/robot_wrapper.py:8-95)
re-hosted on the compiled kinematic tables.  Metadata queries are answered from the host-side
KinematicModel; forward kinematics runs on the GPU through libdexr's ``dexr_fk``."""
from __future__ import annotations

from typing import List, Sequence

import numpy as np
import numpy.typing as npt

from . import _lib
from .model_compiler import compile_fk
from .urdf import KinematicModel, parse_urdf


class _ModelView:
    """Just enough of ``pin.Model`` for code that pokes at ``robot.model`` (tests/test_optimizer.py:50)."""

    def __init__(self, km: KinematicModel):
        self.nq = km.dof
        self.nv = km.dof
        self.names = ["universe"] + km.dof_joint_names
        self.lowerPositionLimit = km.joint_limits[:, 0].copy()
        self.upperPositionLimit = km.joint_limits[:, 1].copy()


class RobotWrapper:
    """This class does not take mimic joint into consideration (same as the reference)."""

    def __init__(self, urdf_path: str, use_collision=False, use_visual=False, add_dummy_free_joints: bool = False):
        if use_visual or use_collision:
            raise NotImplementedError
        self.kin = KinematicModel(parse_urdf(urdf_path, add_dummy_free_joints=add_dummy_free_joints))
        self.model = _ModelView(self.kin)
        self.q0 = np.zeros(self.kin.dof)  # pin.neutral for 1-DoF joints
        self._qpos = np.zeros((1, self.kin.dof))
        self._fk_models = {}

    # ---- properties (robot_wrapper.py:28-52) ------------------------------------------------------
    @property
    def joint_names(self) -> List[str]:
        return list(self.model.names)

    @property
    def dof_joint_names(self) -> List[str]:
        return self.kin.dof_joint_names

    @property
    def dof(self) -> int:
        return self.kin.dof

    @property
    def link_names(self) -> List[str]:
        return self.kin.link_names

    @property
    def joint_limits(self):
        return self.kin.joint_limits

    # ---- queries (robot_wrapper.py:57-77) ---------------------------------------------------------
    def get_joint_index(self, name: str):
        return self.dof_joint_names.index(name)

    def get_link_index(self, name: str):
        """Frame id of a link in pinocchio's frame numbering (universe, root link, then joint frame / child link in
        depth-first order) == ``model.getFrameId(name, pin.BODY)`` (robot_wrapper.py:61-65): it indexes
        ``link_names``."""
        if name not in self.link_names:
            raise ValueError(f"{name} is not a link name. Valid link names: \n{self.link_names}")
        return self.kin.body_frame_id(name)

    def get_link_name(self, index: int) -> str:
        return self.kin.frame_names[index]

    def get_joint_parent_child_frames(self, joint_name: str):
        """robot_wrapper.py:67-77: (``frames[joint_frame].parent``, id of the frame whose previousFrame is the joint's
        frame).  pinocchio's ``Frame.parent`` is the index of the supporting joint in ``model.joints`` (universe = 0)."""
        kin = self.kin
        ids = [i for i, n in enumerate(kin.frame_names) if n == joint_name and kin.frame_table[i][0] != "BODY"]
        if not ids:
            raise ValueError(f"{joint_name} is not a joint name")
        joint_id = ids[0]
        parent_id = kin.frame_table[joint_id][1] + 1
        child_id = -1
        for idx, (_, _, prev, _) in enumerate(kin.frame_table):
            if prev == joint_id and idx != joint_id:
                child_id = idx
        if child_id == -1:
            raise ValueError(f"Can not find child link of {joint_name}")
        return parent_id, child_id

    # ---- kinematics (robot_wrapper.py:82-95), batched ----------------------------------------------
    def compute_forward_kinematics(self, qpos: npt.NDArray):
        self._qpos = np.atleast_2d(np.asarray(qpos, dtype=np.float64))

    def link_positions(self, qpos: npt.NDArray, link_indices: Sequence[int]) -> np.ndarray:
        """(B, nq) -> (B, L, 3) world positions of the given links (frame ids from get_link_index), on the GPU."""
        key = tuple(int(i) for i in link_indices)
        if len(key) > 48:  # a generic FK table (robots of more than 32 joints) holds up to 64 links: ask in chunks
            return np.concatenate([self.link_positions(qpos, key[c:c + 48]) for c in range(0, len(key), 48)], axis=1)
        if key not in self._fk_models:
            names = [self.kin.frames[self.kin.body_of_frame_id(i)].name for i in key]
            self._fk_models[key] = _lib.Model(compile_fk(self.kin, names).to_blob())
        q = np.atleast_2d(np.asarray(qpos, dtype=np.float64))
        return self._fk_models[key].fk(q, len(key))

    def get_link_pose(self, link_id: int) -> npt.NDArray:
        """4x4 pose of one link at the configuration last given to compute_forward_kinematics.  The translation comes
        from the device path (that is all the retargeting objectives read, optimizer.py:157-159,260,521); the rotation
        block is filled on the host from the same kinematic model."""
        pos = self.link_positions(self._qpos[:1], [link_id])[0, 0]
        T, _ = self.kin.frame_pose_and_local_jacobian(self._qpos[0], self.kin.body_of_frame_id(link_id))
        T[:3, 3] = pos
        return T

    def get_link_pose_inv(self, link_id: int) -> npt.NDArray:
        return np.linalg.inv(self.get_link_pose(link_id))

    def compute_single_link_local_jacobian(self, qpos, link_id: int) -> npt.NDArray:
        """6 x dof frame Jacobian in the LOCAL frame (robot_wrapper.py:93-95), host float64: the solver kernels never
        build this matrix (they form the world-aligned columns a x (p - o) directly), it exists for callers of the
        reference API."""
        return self.kin.frame_pose_and_local_jacobian(np.asarray(qpos, dtype=np.float64),
                                                      self.kin.body_of_frame_id(link_id))[1]
