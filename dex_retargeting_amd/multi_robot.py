"""K robots retargeted from the SAME human hand sequences, device-resident (second half).

The reference's offline viewer drives one ``SeqRetargeting`` per robot from one DexYCB hand track
(/root/reference/example/position_retargeting/hand_robot_viewer.py:134-181): ``warm_start(wrist_pos, wrist_quat,
hand_type, is_mano_convention=True)`` once per robot (``:150-160``), then for every frame and every robot
``ref_value = joint[optimizer.target_link_human_indices]; qpos = retargeting.retarget(ref_value)`` (``:170-176``) with the
position configs, whose URDFs carry six dummy free joints (``add_dummy_free_joint``).

Here B independent hand tracks advance in lock-step and a frame of all K robots is ONE fleet batch of K x B rows
(``dexr_retarget_multi_dev``): row k * B + b is robot k following track b; every row reads the same raw (21, 3) keypoints
of its track and forms its own ``ref_value`` from its model's ``target_link_human_indices`` inside the kernel.  Per row
the state the reference keeps per ``SeqRetargeting`` object lives in HBM: the unfiltered ``last_qpos`` (clipped to the
joint limits before each solve, seq_retarget.py:118-124) and the low-pass filter output (optimizer_utils.py:7-13).
Nothing synchronises with the host between frames.
"""
from __future__ import annotations

from typing import List, Optional, Sequence

import numpy as np

from .constants import HandType
from .fleet import MixedFleet
from .kinematics_adaptor import MimicJointKinematicAdaptor
from .seq_retarget import _DUMMY_NAMES, SeqRetargeting, warm_start_pose_vec


class MultiRobotSeqRetargeting:
    """``retargetings``: K ``SeqRetargeting`` objects as ``RetargetingConfig.build()`` returns them (their optimizers, joint
    limits and filters' alpha are used; the objects themselves stay untouched and usable).  ``batch``: B hand tracks."""

    def __init__(self, retargetings: Sequence[SeqRetargeting], batch: int, device: str = "cuda:0"):
        import torch

        self.torch = torch
        self.device = torch.device(device)
        self.retargetings: List[SeqRetargeting] = list(retargetings)
        self.optimizers = [r.optimizer for r in self.retargetings]
        for o in self.optimizers:
            if len(o.idx_pin2fixed):
                raise ValueError("MultiRobotSeqRetargeting: every non-mimic joint must be a target joint (the offline "
                                 "position configs optimise all of them)")
            if o.retargeting_type == "DEXPILOT":
                raise ValueError("MultiRobotSeqRetargeting serves vector / position models")
        self.K, self.B = len(self.retargetings), int(batch)
        self.fleet = MixedFleet(self.optimizers, device=device)
        K, B, n_max = self.K, self.B, self.fleet.n_max
        dev = self.device
        lo = np.full((K, n_max), -np.inf, np.float32)
        hi = np.full((K, n_max), np.inf, np.float32)
        mid = np.zeros((K, n_max), np.float32)
        for k, r in enumerate(self.retargetings):
            n = r.optimizer.opt_dof
            lo[k, :n], hi[k, :n] = r.joint_limits[:, 0], r.joint_limits[:, 1]
            mid[k, :n] = r.joint_limits.mean(1)
        # row k * B + b: per-row limits / start points (robot-major, so a robot's rows are one contiguous block)
        self._lo = torch.from_numpy(np.repeat(lo, B, 0)).to(dev)
        self._hi = torch.from_numpy(np.repeat(hi, B, 0)).to(dev)
        self._mid = torch.from_numpy(np.repeat(mid, B, 0)).to(dev)
        self.model_id = torch.arange(K, dtype=torch.int32, device=dev).repeat_interleave(B).contiguous()
        self.last_qpos = self._mid.clone()
        self._last_clipped = torch.empty_like(self.last_qpos)
        self._q = torch.zeros_like(self.last_qpos)
        self._status = torch.zeros(K * B, dtype=torch.int32, device=dev)
        self._kp = torch.empty((K * B, 21, 3), dtype=torch.float32, device=dev)
        self._compose = []
        for r in self.retargetings:
            o = r.optimizer
            ad = o.adaptor
            mim = None
            if isinstance(ad, MimicJointKinematicAdaptor):
                mim = (torch.tensor(ad.idx_pin2mimic, dtype=torch.long, device=dev),
                       torch.tensor(ad.idx_pin2source, dtype=torch.long, device=dev),
                       torch.tensor(ad.multipliers, dtype=torch.float64, device=dev),
                       torch.tensor(ad.offsets, dtype=torch.float64, device=dev))
            alpha = None if r.filter is None else float(r.filter.alpha)
            self._compose.append(dict(idx_t=torch.tensor(o.idx_pin2target, dtype=torch.long, device=dev), mimic=mim, alpha=alpha,
                                      rq=torch.zeros((B, o.robot.dof), dtype=torch.float64, device=dev), y=None))
        self.num_retargeting = 0

    def reset(self):
        """seq_retarget.py:150-153: last_qpos back to the limit midpoint, counters to zero -- the low-pass filter keeps its
        state, exactly as the reference's SeqRetargeting.reset() leaves `self.filter` alone (use reset_filter() for a clean
        filter as well)."""
        self.last_qpos.copy_(self._mid)
        self.num_retargeting = 0

    def reset_filter(self):
        """[not-in-ref] forget the low-pass filter state of every robot: the next frame initialises it (LPFilter.reset)."""
        for c in self._compose:
            c["y"] = None

    def warm_start(self, wrist_pos, wrist_quat, hand_type: HandType = HandType.right, is_mano_convention: bool = False):
        """hand_robot_viewer.py:150-160 for every robot and track: wrist_pos (B, 3), wrist_quat (B, 4) (w, x, y, z) -> the
        six dummy free joints of every robot's rows (seq_retarget.py:45-110), scattered on the device."""
        torch = self.torch
        wrist_pos, wrist_quat = np.atleast_2d(np.asarray(wrist_pos)), np.atleast_2d(np.asarray(wrist_quat))
        if wrist_pos.shape[0] != self.B or wrist_quat.shape[0] != self.B:
            raise ValueError(f"expected {self.B} wrist poses, got {wrist_pos.shape[0]} / {wrist_quat.shape[0]}")
        for k, o in enumerate(self.optimizers):
            pose = warm_start_pose_vec(o, wrist_pos, wrist_quat, hand_type, is_mano_convention)  # (B, 6)
            pairs = [(num, _DUMMY_NAMES.index(n)) for num, n in enumerate(o.target_joint_names) if n in _DUMMY_NAMES]
            if not pairs:
                continue
            dst = torch.tensor([p[0] for p in pairs], dtype=torch.long, device=self.device)
            src = torch.from_numpy(np.ascontiguousarray(pose[:, [p[1] for p in pairs]], dtype=np.float32)).to(self.device)
            self.last_qpos[k * self.B:(k + 1) * self.B].index_copy_(1, dst, src)

    def retarget(self, joint_pos) -> list:
        """joint_pos: (B, 21, 3) float32 tensor on the device -- one frame of every hand track (world frame, as
        hand_robot_viewer.py:170-174 feeds it).  Returns K float64 tensors (B, robot_k.dof) in each robot's pinocchio dof
        order (views of internal buffers, overwritten by the next call)."""
        torch = self.torch
        K, B = self.K, self.B
        if joint_pos.dtype != torch.float32 or joint_pos.device != self.device:
            joint_pos = joint_pos.to(device=self.device, dtype=torch.float32)
        if tuple(joint_pos.shape) != (B, 21, 3):
            raise ValueError(f"joint_pos must have shape ({B}, 21, 3), got {tuple(joint_pos.shape)}")
        self._kp.view(K, B, 21, 3).copy_(joint_pos.unsqueeze(0).expand(K, -1, -1, -1))  # every robot reads its track's frame
        torch.maximum(self.last_qpos, self._lo, out=self._last_clipped)
        torch.minimum(self._last_clipped, self._hi, out=self._last_clipped)             # seq_retarget.py:118-120
        self._status.zero_()
        self.fleet.retarget(self.model_id, self._kp, self._last_clipped, None, out=self._q, status=self._status)
        bad = (self._status == 2).unsqueeze(1)  # non-finite solve: keep last_qpos (optimizer.py:100-102)
        self.last_qpos.copy_(torch.where(bad, self._last_clipped, self._q))
        outs = []
        for k, c in enumerate(self._compose):
            rq = c["rq"]
            rq.zero_()
            n = self.optimizers[k].opt_dof
            rq.index_copy_(1, c["idx_t"], self.last_qpos[k * B:(k + 1) * B, :n].to(torch.float64))
            if c["mimic"] is not None:  # kinematics_adaptor.py:102-105
                im, isrc, mul, off = c["mimic"]
                rq.index_copy_(1, im, rq.index_select(1, isrc) * mul + off)
            if c["alpha"] is None:
                outs.append(rq)
                continue
            if c["y"] is None:  # LPFilter.next: the first frame passes through (optimizer_utils.py:7-13)
                c["y"] = rq.clone()
            else:
                c["y"].add_(rq - c["y"], alpha=c["alpha"])
            outs.append(c["y"])
        self.num_retargeting += 1
        return outs

    def raw_qpos(self, k: int):
        """(B, n_opt_k) float32 view of robot k's unfiltered last answers, in its target_joint_names order."""
        return self.last_qpos[k * self.B:(k + 1) * self.B, : self.optimizers[k].opt_dof]
