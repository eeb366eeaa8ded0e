"""Mimic-joint adaptor metadata.  Same constructor checks and index tables as the reference
(/root/reference/src/dex_retargeting/kinematics_adaptor.py:46-113).  The per-evaluation work of the
reference's adaptor (``forward_qpos`` inside the objective and ``backward_jacobian`` on the Jacobian stack)
is compiled into the kinematic tables and executed inside the HIP kernel; ``forward_qpos`` remains here only for
composing the returned robot qpos (seq_retarget.py:129-130), which is host-side bookkeeping."""
from __future__ import annotations

from abc import abstractmethod
from typing import List

import numpy as np

from .robot_wrapper import RobotWrapper


class KinematicAdaptor:
    def __init__(self, robot: RobotWrapper, target_joint_names: List[str]):
        self.robot = robot
        self.target_joint_names = target_joint_names
        self.idx_pin2target = np.array([robot.get_joint_index(n) for n in target_joint_names])

    @abstractmethod
    def forward_qpos(self, qpos: np.ndarray) -> np.ndarray:
        pass


class MimicJointKinematicAdaptor(KinematicAdaptor):
    def __init__(self, robot: RobotWrapper, target_joint_names: List[str], source_joint_names: List[str],
                 mimic_joint_names: List[str], multipliers: List[float], offsets: List[float]):
        super().__init__(robot, target_joint_names)
        self.multipliers = np.array(multipliers)
        self.offsets = np.array(offsets)

        union_set = set(mimic_joint_names).intersection(set(target_joint_names))
        if len(union_set) > 0:
            raise ValueError(
                f"Mimic joint should not be one of the target joints.\n"
                f"Mimic joints: {mimic_joint_names}.\n"
                f"Target joints: {target_joint_names}\n"
                f"You need to specify the target joint names explicitly in your retargeting config"
                f" for robot with mimic joint constraints: {target_joint_names}")

        self.idx_pin2source = np.array([robot.get_joint_index(name) for name in source_joint_names])
        self.idx_pin2mimic = np.array([robot.get_joint_index(name) for name in mimic_joint_names])
        self.idx_target2source = np.array([self.target_joint_names.index(n) for n in source_joint_names])

        len_source, len_mimic = self.idx_target2source.shape[0], self.idx_pin2mimic.shape[0]
        len_mul, len_offset = self.multipliers.shape[0], self.offsets.shape[0]
        if not (len_mimic == len_source == len_mul == len_offset):
            raise ValueError(
                f"Mimic joints setting dimension mismatch.\n"
                f"Source joints: {len_source}, mimic joints: {len_mimic}, multiplier: {len_mul}, offset: {len_offset}")
        self.num_active_joints = len(robot.dof_joint_names) - len_mimic

        if len(mimic_joint_names) != len(np.unique(mimic_joint_names)):
            raise ValueError(f"Redundant mimic joint names: {mimic_joint_names}")

    def forward_qpos(self, pin_qpos: np.ndarray) -> np.ndarray:
        """In place on the last axis; accepts (nq,) or (B, nq)."""
        mimic_qpos = pin_qpos[..., self.idx_pin2source] * self.multipliers + self.offsets
        pin_qpos[..., self.idx_pin2mimic] = mimic_qpos
        return pin_qpos
