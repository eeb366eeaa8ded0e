"""Device-resident batched SeqRetargeting.

B independent sequences advance in lock-step; every piece of per-sequence state the reference keeps on its
``SeqRetargeting`` / ``LPFilter`` / ``DexPilotOptimizer`` objects lives in HBM as a torch tensor and never visits the
host between frames:

* ``last_qpos``   (B, n_opt) f32 -- the UNFILTERED previous solution (seq_retarget.py:124), clipped to the joint limits
  before it is used as start point and regularisation target (seq_retarget.py:118-120);
* ``filtered``    (B, dof)  f64 -- LPFilter.y (optimizer_utils.py:7-13): first frame passes through, then
  ``y += alpha (x - y)``;
* ``state``       (B,) int32  -- DexPilot projection bits (optimizer.py:466-476).

One frame = one ``dexr_retarget_dev`` enqueue on the current torch stream (+ its float64 polish launch for position /
DexPilot models) and a handful of element-wise torch ops (compose robot qpos, mimic fill, EMA); nothing synchronises,
so T frames can be captured into one HIP graph (``torch.cuda.CUDAGraph``) and replayed.
"""
from __future__ import annotations

from typing import Optional

import numpy as np

from . import _lib
from .kinematics_adaptor import MimicJointKinematicAdaptor
from .optimizer import Optimizer


class DeviceSeqRetargeting:
    def __init__(self, optimizer: Optimizer, batch: int, has_joint_limits: bool = True,
                 low_pass_alpha: Optional[float] = None, device: str = "cuda:0"):
        import torch

        self.torch = torch
        self.optimizer = optimizer
        self.batch = int(batch)
        self.device = torch.device(device)
        robot = optimizer.robot
        joint_limits = np.ones_like(robot.joint_limits)
        joint_limits[:, 0], joint_limits[:, 1] = -1e4, 1e4
        if has_joint_limits:
            joint_limits[:] = robot.joint_limits[:]
            optimizer.set_joint_limit(joint_limits[optimizer.idx_pin2target])
        self.joint_limits = joint_limits[optimizer.idx_pin2target]
        self.alpha = low_pass_alpha if (low_pass_alpha is not None and 0 <= low_pass_alpha <= 1) else None
        self.model = optimizer.device_model()
        self.n_opt = optimizer.opt_dof
        self.n_fixed = len(optimizer.idx_pin2fixed)
        self.dexpilot = optimizer.retargeting_type == "DEXPILOT"
        dev = self.device
        self._lo = torch.tensor(self.joint_limits[:, 0], dtype=torch.float32, device=dev)
        self._hi = torch.tensor(self.joint_limits[:, 1], dtype=torch.float32, device=dev)
        self._idx_t = torch.tensor(optimizer.idx_pin2target, dtype=torch.long, device=dev)
        self._idx_f = torch.tensor(optimizer.idx_pin2fixed, dtype=torch.long, device=dev)
        ad = optimizer.adaptor
        self._mimic = None
        if isinstance(ad, MimicJointKinematicAdaptor):
            self._mimic = (torch.tensor(ad.idx_pin2mimic, dtype=torch.long, device=dev),
                           torch.tensor(ad.idx_pin2source, dtype=torch.long, device=dev),
                           torch.tensor(ad.multipliers, dtype=torch.float64, device=dev),
                           torch.tensor(ad.offsets, dtype=torch.float64, device=dev))
        self._opts = optimizer._options()
        # persistent buffers (fixed addresses -> graph-capturable)
        B = self.batch
        self.last_qpos = torch.empty((B, self.n_opt), dtype=torch.float32, device=dev)
        self._last_clipped = torch.empty_like(self.last_qpos)
        self._q = torch.empty_like(self.last_qpos)
        self._status = torch.zeros(B, dtype=torch.int32, device=dev)
        self.state = torch.zeros(B, dtype=torch.int32, device=dev)
        self.robot_qpos = torch.zeros((B, robot.dof), dtype=torch.float64, device=dev)
        self.filtered = torch.zeros_like(self.robot_qpos)
        self._kp_mano = None
        self._no_fixed = torch.zeros((B, max(self.n_fixed, 1)), dtype=torch.float32, device=dev)
        self._filter_init = False
        self.reset()

    def reset(self):
        """seq_retarget.py:155-158 of the reference, per sequence: last_qpos back to the limit midpoint, counters to zero; the
        low-pass filter and the DexPilot projection bits keep their state, as the reference's reset() leaves `self.filter` and
        `optimizer.projected` alone (reset_filter() / reset_state() clear them; same meaning in all three batch wrappers)."""
        mid = self.torch.tensor(self.joint_limits.mean(1).astype(np.float32), device=self.device)
        self.last_qpos.copy_(mid[None].expand(self.batch, -1))
        self.num_retargeting = 0

    def reset_filter(self):
        """[not-in-ref] forget the low-pass filter state: the next frame initialises it (LPFilter.reset)."""
        self._filter_init = False

    def reset_state(self):
        """[not-in-ref] clear the DexPilot projection bits of every sequence."""
        self.state.zero_()

    def set_qpos(self, target_qpos):
        """(B, n_opt) start point for the next frame (== SeqRetargeting.set_qpos per sequence)."""
        self.last_qpos.copy_(self.torch.as_tensor(target_qpos, dtype=self.torch.float32, device=self.device))

    def warm_start(self, wrist_pos, wrist_quat, hand_type=None, is_mano_convention: bool = False):
        """Per-sequence analytic wrist initialisation (seq_retarget.py:45-110): wrist_pos (B,3), wrist_quat (B,4)."""
        from .constants import HandType
        from .seq_retarget import _DUMMY_NAMES, warm_start_pose_vec

        pose = warm_start_pose_vec(self.optimizer, np.asarray(wrist_pos), np.asarray(wrist_quat),
                                   hand_type or HandType.right, is_mano_convention)
        if pose.shape[0] != self.batch:
            raise ValueError(f"expected {self.batch} wrist poses, got {pose.shape[0]}")
        # scatter the six pose columns into the dummy joints' slots of last_qpos on the device (no read-back)
        pairs = [(num, _DUMMY_NAMES.index(n)) for num, n in enumerate(self.optimizer.target_joint_names) if n in _DUMMY_NAMES]
        if pairs:
            dst = self.torch.tensor([p[0] for p in pairs], dtype=self.torch.long, device=self.device)
            src = np.ascontiguousarray(pose[:, [p[1] for p in pairs]], dtype=np.float32)
            self.last_qpos.index_copy_(1, dst, self.torch.from_numpy(src).to(self.device))

    def retarget_keypoints(self, keypoints, fixed_qpos=None):
        """Same as retarget() but fed with raw (B, 21, 3) hand keypoints; ref_value is formed inside the kernel."""
        return self.retarget(keypoints, fixed_qpos, _keypoints=True)

    def retarget_raw_keypoints(self, keypoints, hand_type="Right", fixed_qpos=None):
        """Raw detector output -> robot qpos without leaving the device: (B, 21, 3) keypoints in the detector's own
        frame are centred on the wrist and rotated into the MANO frame (``dexr_mano_keypoints_dev`` ==
        single_hand_detector.py:102-104,129-158), then retargeted as :meth:`retarget_keypoints` does."""
        from . import keypoints as kpmod

        torch = self.torch
        if keypoints.dtype != torch.float32 or not keypoints.is_contiguous() or keypoints.device != self.device:
            keypoints = keypoints.to(device=self.device, dtype=torch.float32).contiguous()
        if tuple(keypoints.shape) != (self.batch, kpmod.N_KEYPOINTS, 3):
            raise ValueError(f"keypoints must have shape ({self.batch}, {kpmod.N_KEYPOINTS}, 3), got {tuple(keypoints.shape)}")
        if self._kp_mano is None:
            self._kp_mano = torch.empty_like(keypoints)
        kpmod.mano_keypoints_dev(self.batch, keypoints.data_ptr(), self._kp_mano.data_ptr(), 0, hand_type,
                                 torch.cuda.current_stream(self.device).cuda_stream)
        return self.retarget(self._kp_mano, fixed_qpos, _keypoints=True)

    def _dof_map(self):
        """How every robot dof (pinocchio order) is composed from the optimiser's answer (seq_retarget.py:125-130)."""
        opt = self.optimizer
        n_q = opt.robot.dof
        kind, idx = np.ones(n_q, np.int32), np.zeros(n_q, np.int32)
        mult, off = np.ones(n_q), np.zeros(n_q)
        for i, p in enumerate(opt.idx_pin2target):
            kind[p], idx[p] = 0, i
        for i, p in enumerate(opt.idx_pin2fixed):
            kind[p], idx[p] = 1, i
        ad = opt.adaptor
        if isinstance(ad, MimicJointKinematicAdaptor):
            for m, s_, a, b in zip(ad.idx_pin2mimic, ad.idx_pin2source, ad.multipliers, ad.offsets):
                kind[m], idx[m], mult[m], off[m] = 2, s_, a, b
        return kind, idx, mult, off

    def retarget_sequence(self, inputs_seq, fixed_seq=None, out=None, keypoints: bool = True, raw_out=None,
                          status_out=None):
        """T lock-step frames of the B sequences in TWO kernel launches: ``dexr_retarget_seq_dev`` (every lane loops
        over its sequence's T frames inside the kernel, carrying the clipped last_qpos and the DexPilot bits) and
        ``dexr_seq_compose_dev`` (robot-qpos composition, mimic fill, low-pass filter for all T x B frames).
        inputs_seq: (T, B, 21, 3) float32 keypoints (or (T, B, n_ref, 3) ref_value rows with keypoints=False);
        fixed_seq: (T, B, n_fixed) float32 when the model has caller-supplied fixed joints.
        Returns the (T, B, dof) float64 tensor SeqRetargeting.retarget would have returned frame by frame; the carried
        state (last_qpos, filter, DexPilot bits) continues from / for calls of either kind."""
        torch = self.torch
        if inputs_seq.dtype != torch.float32 or not inputs_seq.is_contiguous() or inputs_seq.device != self.device:
            inputs_seq = inputs_seq.to(device=self.device, dtype=torch.float32).contiguous()
        T, B = int(inputs_seq.shape[0]), self.batch
        if int(inputs_seq.shape[1]) != B:
            raise ValueError(f"inputs_seq must have shape (T, {B}, rows, 3), got {tuple(inputs_seq.shape)}")
        fixed_ptr = 0
        if self.n_fixed:
            if fixed_seq is None:
                raise ValueError(f"Optimizer has {self.n_fixed} joints but non_target_qpos None is given")
            fixed_seq = fixed_seq.to(device=self.device, dtype=torch.float32).contiguous()
            fixed_ptr = fixed_seq.data_ptr()
        n_q = self.robot_qpos.shape[1]
        if raw_out is None:
            raw_out = torch.empty((T, B, self.n_opt), dtype=torch.float32, device=self.device)
        if out is None:
            out = torch.empty((T, B, n_q), dtype=torch.float64, device=self.device)
        stream = torch.cuda.current_stream(self.device).cuda_stream
        self.model.retarget_seq_dev(B, T, inputs_seq.data_ptr(), fixed_ptr, self.last_qpos.data_ptr(),
                                    self.state.data_ptr() if self.dexpilot else 0, raw_out.data_ptr(),
                                    status_ptr=status_out.data_ptr() if status_out is not None else 0,
                                    joint_limit_eps=1e-3, opts=self._opts, stream=stream, keypoints=keypoints)
        kind, idx, mult, off = self._dof_map()
        _lib.seq_compose_dev(B, T, kind, idx, mult, off, self.n_opt, self.n_fixed, raw_out.data_ptr(), fixed_ptr,
                             -1.0 if self.alpha is None else float(self.alpha), self.filtered.data_ptr(),
                             not self._filter_init, out.data_ptr(), stream)
        if self.alpha is not None:
            self._filter_init = True
        self.num_retargeting += T
        return out

    def capture(self, keypoints_seq, out=None, fixed_seq=None):
        """Capture T lock-step frames into ONE HIP graph (``torch.cuda.CUDAGraph``): ``keypoints_seq`` is a persistent
        (T, B, 21, 3) float32 CUDA tensor the caller refills before every ``graph.replay()``; ``out`` (T, B, dof)
        float64 receives the filtered robot qpos of every frame.  The carried state (last_qpos, filter, DexPilot bits)
        lives in HBM, so consecutive replays continue the sequences: T kernel launches + ~10 T element-wise ops cost
        one graph launch.  At least one eager frame must have run before (the low-pass filter's first frame is a
        host-side branch).  Models with caller-supplied fixed joints take ``fixed_seq``, a persistent (T, B, n_fixed)
        float32 CUDA tensor refilled the same way.  Returns (graph, out)."""
        torch = self.torch
        if self.alpha is not None and not self._filter_init:
            raise RuntimeError("run one eager frame first: the first frame initialises the low-pass filter")
        T = int(keypoints_seq.shape[0])
        if self.n_fixed:
            if fixed_seq is None:
                raise ValueError(f"Optimizer has {self.n_fixed} joints but non_target_qpos None is given")
            if fixed_seq.dtype != torch.float32 or not fixed_seq.is_contiguous() or fixed_seq.device != self.device or \
                    tuple(fixed_seq.shape) != (T, self.batch, self.n_fixed):
                raise ValueError(f"fixed_seq must be a contiguous float32 tensor of shape ({T}, {self.batch}, {self.n_fixed}) "
                                 f"on this device")
        if keypoints_seq.dtype != torch.float32 or not keypoints_seq.is_contiguous() or keypoints_seq.device != self.device:
            raise ValueError("keypoints_seq must be a contiguous float32 tensor on this device")
        if out is None:
            out = torch.empty((T, self.batch, self.robot_qpos.shape[1]), dtype=torch.float64, device=self.device)
        graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(graph):
            for t in range(T):
                out[t].copy_(self.retarget(keypoints_seq[t], fixed_seq[t] if self.n_fixed else None, _keypoints=True))
        self.num_retargeting -= T  # capture itself solves nothing
        return graph, out

    def retarget(self, ref_value, fixed_qpos=None, _keypoints=False):
        """ref_value: (B, n_ref, 3) float32 CUDA tensor (contiguous).  Returns the (B, dof) float64 CUDA tensor of
        filtered robot qpos in pinocchio dof order (a view of an internal buffer, overwritten by the next call)."""
        torch = self.torch
        B = self.batch
        if ref_value.dtype != torch.float32 or not ref_value.is_contiguous() or ref_value.device != self.device:
            ref_value = ref_value.to(device=self.device, dtype=torch.float32).contiguous()
        fixed = self._no_fixed
        if self.n_fixed:
            if fixed_qpos is None:
                raise ValueError(f"Optimizer has {self.n_fixed} joints but non_target_qpos None is given")
            fixed = fixed_qpos.to(device=self.device, dtype=torch.float32).contiguous()
        torch.clamp(self.last_qpos, min=self._lo, max=self._hi, out=self._last_clipped)
        stream = torch.cuda.current_stream(self.device).cuda_stream
        self.model.retarget_dev(B, ref_value.data_ptr(), fixed.data_ptr() if self.n_fixed else 0,
                                self._last_clipped.data_ptr(), self.state.data_ptr() if self.dexpilot else 0,
                                self._q.data_ptr(), status_ptr=self._status.data_ptr(), opts=self._opts, stream=stream,
                                keypoints=_keypoints)
        # non-finite solve -> keep last_qpos, like the reference's RuntimeError branch (optimizer.py:100-102)
        bad = (self._status == 2).unsqueeze(1)
        self.last_qpos.copy_(torch.where(bad, self._last_clipped, self._q))
        rq = self.robot_qpos
        rq.zero_()
        if self.n_fixed:
            rq.index_copy_(1, self._idx_f, fixed.to(torch.float64))
        rq.index_copy_(1, self._idx_t, self.last_qpos.to(torch.float64))
        if self._mimic is not None:  # kinematics_adaptor.py:102-105
            im, isrc, mul, off = self._mimic
            rq.index_copy_(1, im, rq.index_select(1, isrc) * mul + off)
        self.num_retargeting += 1
        if self.alpha is None:
            return rq
        if not self._filter_init:
            self.filtered.copy_(rq)
            self._filter_init = True
        else:
            self.filtered.add_(rq - self.filtered, alpha=float(self.alpha))
        return self.filtered
