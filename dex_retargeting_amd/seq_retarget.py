"""SeqRetargeting: the reference's per-frame wrapper (/root/reference/src/dex_retargeting/seq_retarget.py:12-161)
with the same public surface (retarget, warm_start, set_qpos, get_qpos, reset, verbose, joint_names), plus
``BatchedSeqRetargeting`` for B sequences advanced in lock-step (one kernel launch per time step)."""
from __future__ import annotations

import time
from typing import Optional

import numpy as np

from .constants import OPERATOR2MANO, HandType
from .optimizer import Optimizer
from .optimizer_utils import LPFilter


def _matrix_from_quaternion(q) -> np.ndarray:
    """(..., 4) quaternions (w, x, y, z) -> (..., 3, 3) rotation matrices (pytransform3d's
    ``rotations.matrix_from_quaternion`` convention, which normalises its input)."""
    q = np.asarray(q, dtype=np.float64)
    q = q / np.linalg.norm(q, axis=-1, keepdims=True)
    w, x, y, z = q[..., 0], q[..., 1], q[..., 2], q[..., 3]
    return np.stack([np.stack([1 - 2 * (y * y + z * z), 2 * (x * y - z * w), 2 * (x * z + y * w)], -1),
                     np.stack([2 * (x * y + z * w), 1 - 2 * (x * x + z * z), 2 * (y * z - x * w)], -1),
                     np.stack([2 * (x * z - y * w), 2 * (y * z + x * w), 1 - 2 * (x * x + y * y)], -1)], -2)


def _intrinsic_xyz_euler(R: np.ndarray) -> np.ndarray:
    """(..., 3, 3) -> angles (a, b, c) with R = Rx(a) Ry(b) Rz(c): euler_from_matrix(R, 0, 1, 2, extrinsic=False)."""
    b = np.arcsin(np.clip(R[..., 0, 2], -1.0, 1.0))
    a = np.arctan2(-R[..., 1, 2], R[..., 2, 2])
    c = np.arctan2(-R[..., 0, 1], R[..., 0, 0])
    return np.stack([a, b, c], -1)


_DUMMY_NAMES = ["dummy_x_translation_joint", "dummy_y_translation_joint", "dummy_z_translation_joint",
                "dummy_x_rotation_joint", "dummy_y_rotation_joint", "dummy_z_rotation_joint"]


def warm_start_pose_vec(optimizer: Optimizer, wrist_pos: np.ndarray, wrist_quat: np.ndarray,
                        hand_type: HandType = HandType.right, is_mano_convention: bool = False) -> np.ndarray:
    """Batched analytic 6-DoF initialisation of the dummy free joints (seq_retarget.py:45-110):
    wrist_pos (B,3), wrist_quat (B,4) (w,x,y,z) -> (B,6) values for the x/y/z translation and the intrinsic-xyz
    rotation dummy joints that put the hand's root link at the given wrist pose."""
    wrist_pos = np.atleast_2d(np.asarray(wrist_pos, dtype=np.float64))
    wrist_quat = np.atleast_2d(np.asarray(wrist_quat, dtype=np.float64))
    if wrist_pos.shape[1] != 3:
        raise ValueError(f"Wrist pos: {wrist_pos} is not a 3-dim vector.")
    if wrist_quat.shape[1] != 4:
        raise ValueError(f"Wrist quat: {wrist_quat} is not a 4-dim vector.")
    operator2mano = OPERATOR2MANO[hand_type] if is_mano_convention else np.eye(3)
    robot = optimizer.robot
    urdf_joint = robot.kin.urdf.joint_map[_DUMMY_NAMES[5]]  # its child link is the hand's original root link
    wrist_link_id = robot.get_link_index(urdf_joint.child)
    robot.compute_forward_kinematics(robot.q0.copy())  # dummy joints at zero (seq_retarget.py:84-93)
    root2wrist = robot.get_link_pose_inv(wrist_link_id)
    # target_root = [R_q op^T | pos] @ root2wrist for all B poses at once (seq_retarget.py:73-95)
    Rw = _matrix_from_quaternion(wrist_quat) @ operator2mano.T
    Rroot = Rw @ root2wrist[:3, :3]
    proot = Rw @ root2wrist[:3, 3] + wrist_pos
    return np.concatenate([proot, _intrinsic_xyz_euler(Rroot)], axis=1)


class SeqRetargeting:
    def __init__(self, optimizer: Optimizer, has_joint_limits=True, lp_filter: Optional[LPFilter] = None):
        self.optimizer = optimizer
        robot = self.optimizer.robot

        # Joint limit (seq_retarget.py:23-31)
        self.has_joint_limits = has_joint_limits
        joint_limits = np.ones_like(robot.joint_limits)
        joint_limits[:, 0] = -1e4
        joint_limits[:, 1] = 1e4
        if has_joint_limits:
            joint_limits[:] = robot.joint_limits[:]
            self.optimizer.set_joint_limit(joint_limits[self.optimizer.idx_pin2target])
        self.joint_limits = joint_limits[self.optimizer.idx_pin2target]

        # Temporal information
        self.last_qpos = joint_limits.mean(1)[self.optimizer.idx_pin2target].astype(np.float32)
        self.accumulated_time = 0
        self.num_retargeting = 0
        self.filter = lp_filter
        self.is_warm_started = False

    def warm_start(self, wrist_pos: np.ndarray, wrist_quat: np.ndarray, hand_type: HandType = HandType.right,
                   is_mano_convention: bool = False):
        """Analytic initialisation of the 6 dummy free joints (seq_retarget.py:45-110)."""
        if len(wrist_pos) != 3:
            raise ValueError(f"Wrist pos: {wrist_pos} is not a 3-dim vector.")
        if len(wrist_quat) != 4:
            raise ValueError(f"Wrist quat: {wrist_quat} is not a 4-dim vector.")
        pose_vec = warm_start_pose_vec(self.optimizer, wrist_pos, wrist_quat, hand_type, is_mano_convention)[0]
        for num, joint_name in enumerate(self.optimizer.target_joint_names):
            if joint_name in _DUMMY_NAMES:
                self.last_qpos[num] = pose_vec[_DUMMY_NAMES.index(joint_name)]

        self.is_warm_started = True

    def retarget(self, ref_value, fixed_qpos=np.array([])):
        tic = time.perf_counter()
        qpos = self.optimizer.retarget(
            ref_value=ref_value.astype(np.float32),
            fixed_qpos=fixed_qpos.astype(np.float32),
            last_qpos=np.clip(self.last_qpos, self.joint_limits[:, 0], self.joint_limits[:, 1]),
        )
        self.accumulated_time += time.perf_counter() - tic
        self.num_retargeting += 1
        self.last_qpos = qpos
        robot_qpos = np.zeros(self.optimizer.robot.dof)
        robot_qpos[self.optimizer.idx_pin2fixed] = fixed_qpos
        robot_qpos[self.optimizer.idx_pin2target] = qpos

        if self.optimizer.adaptor is not None:
            robot_qpos = self.optimizer.adaptor.forward_qpos(robot_qpos)

        if self.filter is not None:
            robot_qpos = self.filter.next(robot_qpos)
        return robot_qpos

    def set_qpos(self, robot_qpos: np.ndarray):
        target_qpos = robot_qpos[self.optimizer.idx_pin2target]
        self.last_qpos = target_qpos

    def get_qpos(self, fixed_qpos: Optional[np.ndarray] = None):
        robot_qpos = np.zeros(self.optimizer.robot.dof)
        robot_qpos[self.optimizer.idx_pin2target] = self.last_qpos
        if fixed_qpos is not None:
            robot_qpos[self.optimizer.idx_pin2fixed] = fixed_qpos
        return robot_qpos

    def verbose(self):
        min_value = self.optimizer.opt.last_optimum_value()
        print(f"Retargeting {self.num_retargeting} times takes: {self.accumulated_time}s")
        print(f"Last distance: {min_value}")

    def reset(self):
        self.last_qpos = self.joint_limits.mean(1).astype(np.float32)
        self.num_retargeting = 0
        self.accumulated_time = 0

    @property
    def joint_names(self):
        return self.optimizer.robot.dof_joint_names


class BatchedSeqRetargeting:
    """B independent sequences advanced in lock-step: every call to ``retarget`` solves frame t of all B
    sequences in one kernel launch and carries, per sequence, exactly the state the reference carries per object:
    the UNFILTERED last_qpos (seq_retarget.py:124), the low-pass filter output (optimizer_utils.py:7-13) and the
    DexPilot projection bits (optimizer.py:466-476)."""

    def __init__(self, optimizer: Optimizer, batch: int, has_joint_limits=True, low_pass_alpha: Optional[float] = None):
        self.optimizer = optimizer
        self.batch = int(batch)
        robot = optimizer.robot
        joint_limits = np.ones_like(robot.joint_limits)
        joint_limits[:, 0], joint_limits[:, 1] = -1e4, 1e4
        if has_joint_limits:
            joint_limits[:] = robot.joint_limits[:]
            optimizer.set_joint_limit(joint_limits[optimizer.idx_pin2target])
        self.joint_limits = joint_limits[optimizer.idx_pin2target]
        self.alpha = low_pass_alpha
        self.filtered: Optional[np.ndarray] = None
        st = self.optimizer._state_in(self.batch)
        self.state = None if st is None else np.zeros(self.batch, dtype=np.uint32)
        self.reset()

    def reset(self):
        """seq_retarget.py:155-158 of the reference, per sequence: last_qpos back to the limit midpoint, counters to zero.  The
        low-pass filter and the DexPilot projection bits keep their state -- the reference's reset() touches neither
        `self.filter` nor `optimizer.projected`; reset_filter() / reset_state() clear them (the three batch wrappers -- this one,
        DeviceSeqRetargeting, MultiRobotSeqRetargeting -- mean the same thing by reset())."""
        mid = self.joint_limits.mean(1).astype(np.float32)
        self.last_qpos = np.repeat(mid[None], self.batch, 0)
        self.num_retargeting = 0

    def reset_filter(self):
        """[not-in-ref] forget the low-pass filter state: the next frame initialises it (LPFilter.reset)."""
        self.filtered = None

    def reset_state(self):
        """[not-in-ref] clear the DexPilot projection bits of every sequence."""
        if self.state is not None:
            self.state[:] = 0

    def warm_start(self, wrist_pos: np.ndarray, wrist_quat: np.ndarray, hand_type: HandType = HandType.right,
                   is_mano_convention: bool = False):
        """Per-sequence analytic wrist initialisation: wrist_pos (B,3), wrist_quat (B,4)."""
        pose = warm_start_pose_vec(self.optimizer, wrist_pos, wrist_quat, hand_type, is_mano_convention)
        if pose.shape[0] != self.batch:
            raise ValueError(f"expected {self.batch} wrist poses, got {pose.shape[0]}")
        for num, joint_name in enumerate(self.optimizer.target_joint_names):
            if joint_name in _DUMMY_NAMES:
                self.last_qpos[:, num] = pose[:, _DUMMY_NAMES.index(joint_name)]

    def retarget_keypoints(self, keypoints: np.ndarray, fixed_qpos: Optional[np.ndarray] = None) -> np.ndarray:
        """Same as retarget() but fed with (B, 21, 3) MANO-frame hand keypoints; ref_value is formed in the kernel."""
        return self.retarget(keypoints, fixed_qpos, _keypoints=True)

    def retarget_raw_keypoints(self, keypoints: np.ndarray, hand_type="Right",
                               fixed_qpos: Optional[np.ndarray] = None) -> np.ndarray:
        """(B, 21, 3) keypoints in the detector's frame: wrist-frame estimate + MANO re-expression
        (single_hand_detector.py:102-104,129-158) on the GPU, then retarget_keypoints()."""
        from .keypoints import mano_keypoints

        return self.retarget(mano_keypoints(keypoints, hand_type)[0], fixed_qpos, _keypoints=True)

    def retarget(self, ref_value: np.ndarray, fixed_qpos: Optional[np.ndarray] = None, _keypoints=False) -> np.ndarray:
        """ref_value (B,n_ref,3); returns float64 (B, robot.dof) in pinocchio dof order."""
        opt = self.optimizer
        B = self.batch
        last = np.clip(self.last_qpos, self.joint_limits[:, 0], self.joint_limits[:, 1]).astype(np.float32)
        fixed = np.zeros((B, len(opt.idx_pin2fixed)), dtype=np.float32) if fixed_qpos is None else fixed_qpos
        solve = opt.retarget_keypoints_batch if _keypoints else opt.retarget_batch
        q = solve(ref_value.astype(np.float32), fixed, last, state=self.state)
        bad = opt.last_info["status"] == 2
        if bad.any():
            q[bad] = last[bad]
        self.last_qpos = q
        self.num_retargeting += 1
        robot_qpos = np.zeros((B, opt.robot.dof))
        robot_qpos[:, opt.idx_pin2fixed] = fixed
        robot_qpos[:, opt.idx_pin2target] = q
        if opt.adaptor is not None:
            robot_qpos = opt.adaptor.forward_qpos(robot_qpos)
        if self.alpha is not None and 0 <= self.alpha <= 1:
            if self.filtered is None:
                self.filtered = robot_qpos
            else:
                self.filtered = self.filtered + self.alpha * (robot_qpos - self.filtered)
            return self.filtered.copy()
        return robot_qpos
