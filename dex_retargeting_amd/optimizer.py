"""Drop-in optimizers: same class names, constructor arguments, attributes and ``retarget`` contract as the
reference's ``VectorOptimizer`` / ``PositionOptimizer`` / ``DexPilotOptimizer``
(/root/reference/src/dex_retargeting/optimizer.py), with the per-frame nlopt+pinocchio+torch solve replaced by
one call into libdexr (HIP, gfx950) through ctypes.

Beyond the reference API every optimizer also offers the batched form the GPU is built for:

    qpos = optimizer.retarget_batch(ref_value (B,n_ref,3), fixed_qpos (B,n_fixed), last_qpos (B,n_opt))

``retarget`` is ``retarget_batch`` with B = 1.  ``get_objective_function`` returns the nlopt-style closure
``objective(x, grad) -> float`` (value WITHOUT, gradient WITH the norm_delta term, exactly like
optimizer.py:146-198, 249-304, 510-575) evaluated by the ``dexr_eval`` kernel: it is the function-level parity
hook.  The minimiser itself is not SLSQP: the kernel drives F(x) = f(x) + norm_delta*|x - last|^2 (the function
whose gradient the reference supplies) to its box-constrained minimum; see DESIGN.md for why parity is defined
against that minimum.
"""
from __future__ import annotations

from abc import abstractmethod
from typing import List, Optional

import numpy as np

from . import _lib
from . import model_compiler as mc
from .kinematics_adaptor import KinematicAdaptor, MimicJointKinematicAdaptor
from .robot_wrapper import RobotWrapper


class _OptHandle:
    """Stand-in for the ``nlopt.opt`` attribute callers poke at (seq_retarget.py:148)."""

    def __init__(self):
        self._last = float("nan")
        self.lower = None
        self.upper = None
        self.ftol_abs = None

    def last_optimum_value(self):
        p = getattr(self, "_pending", None)
        if p is not None:
            delta, q, last, fval = p
            self._last = float(fval[0]) - delta * float(((q.astype(np.float64) - last.astype(np.float64)) ** 2).sum())
            self._pending = None
        return self._last

    def set_ftol_abs(self, v):
        self.ftol_abs = v


class Optimizer:
    retargeting_type = "BASE"

    def __init__(self, robot: RobotWrapper, target_joint_names: List[str], target_link_human_indices: np.ndarray):
        self.robot = robot
        self.num_joints = robot.dof

        joint_names = robot.dof_joint_names
        idx_pin2target = []
        for target_joint_name in target_joint_names:
            if target_joint_name not in joint_names:
                raise ValueError(f"Joint {target_joint_name} given does not appear to be in robot XML.")
            idx_pin2target.append(joint_names.index(target_joint_name))
        self.target_joint_names = target_joint_names
        self.idx_pin2target = np.array(idx_pin2target)
        self.idx_pin2fixed = np.array([i for i in range(robot.dof) if i not in idx_pin2target], dtype=int)
        self.opt = _OptHandle()
        self.opt_dof = len(idx_pin2target)

        self.target_link_human_indices = target_link_human_indices
        link_names = robot.link_names
        self.has_free_joint = len([name for name in link_names if "dummy" in name]) >= 6
        self.adaptor: Optional[KinematicAdaptor] = None

        # solver state
        self._lower = np.full(self.opt_dof, -np.inf)
        self._upper = np.full(self.opt_dof, np.inf)
        self._model: Optional[_lib.Model] = None
        self._compiled: Optional[mc.CompiledModel] = None
        # True: compile to the generic table format even when the model fits the fixed-size records (the general kernel
        # then serves it; used by the tests that cross-check that kernel on the shipped robots)
        self.use_generic_tables = False
        self.solve_options = dict(max_iter=None, tol=None, lambda0=None, newton=None, precision=None, polish=None, strict=None)
        self.last_info: dict = {}

    # ---- configuration (optimizer.py:54-75) --------------------------------------------------------
    def set_joint_limit(self, joint_limits: np.ndarray, epsilon=1e-3):
        if joint_limits.shape != (self.opt_dof, 2):
            raise ValueError(f"Expect joint limits have shape: {(self.opt_dof, 2)}, but get {joint_limits.shape}")
        self._lower = joint_limits[:, 0] - epsilon
        self._upper = joint_limits[:, 1] + epsilon
        self.opt.lower, self.opt.upper = self._lower.tolist(), self._upper.tolist()
        self._model = None

    def get_link_indices(self, target_link_names):
        return [self.robot.get_link_index(link_name) for link_name in target_link_names]

    def set_kinematic_adaptor(self, adaptor: KinematicAdaptor):
        self.adaptor = adaptor
        if isinstance(adaptor, MimicJointKinematicAdaptor):
            fixed_idx = self.idx_pin2fixed
            mimic_idx = adaptor.idx_pin2mimic
            self.idx_pin2fixed = np.array([x for x in fixed_idx if x not in mimic_idx], dtype=int)
        self._model = None

    @property
    def fixed_joint_names(self):
        joint_names = self.robot.dof_joint_names
        return [joint_names[i] for i in self.idx_pin2fixed]

    # ---- table compilation ---------------------------------------------------------------------------
    @abstractmethod
    def _terms(self) -> List[mc.TermSpec]:
        pass

    def _kind(self) -> int:
        return {"VECTOR": mc.KIND_VECTOR, "POSITION": mc.KIND_POSITION, "DEXPILOT": mc.KIND_DEXPILOT}[
            self.retargeting_type]

    def _compile_kwargs(self) -> dict:
        return {}

    def compiled_model(self) -> mc.CompiledModel:
        if self._compiled is None or self._model is None:
            mimic = []
            if isinstance(self.adaptor, MimicJointKinematicAdaptor):
                a = self.adaptor
                mimic = [(int(m), int(s), float(mu), float(of)) for m, s, mu, of in
                         zip(a.idx_pin2mimic, a.idx_pin2source, a.multipliers, a.offsets)]
            self._compiled = mc.compile_model(
                self.robot.kin, self._kind(), self.idx_pin2target.tolist(), self.idx_pin2fixed.tolist(), self._terms(),
                lower=self._lower, upper=self._upper, mimic=mimic,
                human_indices=self.target_link_human_indices, force_generic=self.use_generic_tables,
                **self._compile_kwargs())
        return self._compiled

    def device_model(self) -> _lib.Model:
        if self._model is None:
            self._compiled = None
            self._model = _lib.Model(self.compiled_model().to_blob())
        return self._model

    def _options(self) -> Optional[_lib.SolveOptions]:
        if all(v is None for v in self.solve_options.values()):
            return None
        return _lib.default_options(**self.solve_options)

    # ---- the hot path ---------------------------------------------------------------------------------
    def _state_in(self, B: int):
        return None

    def _state_out(self, state):
        pass

    def retarget_batch(self, ref_value: np.ndarray, fixed_qpos: Optional[np.ndarray], last_qpos: np.ndarray,
                       state: Optional[np.ndarray] = None) -> np.ndarray:
        """B independent frames: ref_value (B,n_ref,3), fixed_qpos (B,n_fixed) or None, last_qpos (B,n_opt)
        -> float32 (B,n_opt) in target_joint_names order.  `state` (B,) uint32 carries the DexPilot projection
        bits per item (updated in place)."""
        last = np.ascontiguousarray(last_qpos, dtype=np.float32)
        if last.ndim != 2 or last.shape[1] != self.opt_dof:
            raise ValueError(f"last_qpos must have shape (B, {self.opt_dof}), got {last.shape}")
        B = last.shape[0]
        n_fixed = len(self.idx_pin2fixed)
        if fixed_qpos is None:
            fixed = np.zeros((B, 0), dtype=np.float32)
        else:
            fixed = np.ascontiguousarray(fixed_qpos, dtype=np.float32).reshape(B, -1)
        if fixed.shape[1] != n_fixed:
            raise ValueError(f"Optimizer has {n_fixed} joints but non_target_qpos {fixed_qpos} is given")
        model = self.device_model()
        n_ref = self._compiled.n_ref
        ref = np.ascontiguousarray(ref_value, dtype=np.float32)
        if ref.size != B * n_ref * 3:
            raise ValueError(f"ref_value must have shape ({B}, {n_ref}, 3), got {np.shape(ref_value)}")
        ref = ref.reshape(B, n_ref, 3)
        q, info = model.retarget(ref, fixed, last, state=self._check_state(state, B), opts=self._options(), want_info=True)
        self.last_info = info
        return q

    @staticmethod
    def _check_state(state: Optional[np.ndarray], B: int) -> Optional[np.ndarray]:
        """The kernels read and write `state` through a raw pointer: insist on the exact layout."""
        if state is None:
            return None
        if not isinstance(state, np.ndarray) or state.dtype != np.uint32 or state.shape != (B,) or \
                not state.flags["C_CONTIGUOUS"]:
            raise ValueError(f"state must be a C-contiguous uint32 array of shape ({B},)")
        return state

    def retarget_keypoints_batch(self, keypoints: np.ndarray, fixed_qpos: Optional[np.ndarray], last_qpos: np.ndarray,
                                 state: Optional[np.ndarray] = None) -> np.ndarray:
        """Like retarget_batch but fed with raw hand keypoints (B, 21, 3): the kernel forms ref_value itself from
        ``target_link_human_indices`` (joint_pos[task] - joint_pos[origin] for vector/DexPilot, joint_pos[idx] for
        position -- what every caller of the reference does first, profile_online_retargeting.py:24-30)."""
        last = np.ascontiguousarray(last_qpos, dtype=np.float32)
        B = last.shape[0]
        fixed = None if fixed_qpos is None else np.ascontiguousarray(fixed_qpos, dtype=np.float32).reshape(B, -1)
        if (0 if fixed is None else fixed.shape[1]) != len(self.idx_pin2fixed):
            raise ValueError(f"Optimizer has {len(self.idx_pin2fixed)} joints but non_target_qpos {fixed_qpos} is given")
        model = self.device_model()
        n_kp = int(self._compiled.header["n_keypoints"])
        kp = np.ascontiguousarray(keypoints, dtype=np.float32)
        if n_kp <= 0:
            raise ValueError("this optimizer carries no target_link_human_indices: keypoint input is not available")
        if kp.size != B * n_kp * 3:  # the kernel indexes n_keypoints rows per frame
            raise ValueError(f"keypoints must have shape ({B}, {n_kp}, 3), got {np.shape(keypoints)}")
        kp = kp.reshape(B, n_kp, 3)
        q, info = model.retarget(kp, fixed, last, state=self._check_state(state, B), opts=self._options(),
                                 want_info=True, keypoints=True)
        self.last_info = info
        return q

    def retarget(self, ref_value, fixed_qpos, last_qpos):
        """Same contract as optimizer.py:77-102 (single frame)."""
        if len(fixed_qpos) != len(self.idx_pin2fixed):
            raise ValueError(
                f"Optimizer has {len(self.idx_pin2fixed)} joints but non_target_qpos {fixed_qpos} is given")
        state = self._state_in(1)
        last = np.asarray(last_qpos, dtype=np.float32)[None]
        q = self.retarget_batch(np.asarray(ref_value)[None], np.asarray(fixed_qpos, dtype=np.float32)[None], last,
                                state=state)
        self._state_out(state)
        if self.last_info["status"][0] == 2:  # non-finite: same fallback as the reference's RuntimeError path
            print("dexr: solver produced non-finite values, returning last_qpos")
            return np.array(last_qpos, dtype=np.float32)
        # nlopt's last_optimum_value() is the closure's return value, i.e. the data term WITHOUT the regulariser
        # (optimizer.py:198,304,575; printed by SeqRetargeting.verbose as "Last distance"); the kernels report
        # F = f + norm_delta |x - last|^2
        # (formed when somebody asks -- SeqRetargeting.verbose() -- not on every frame: three float64 numpy operations per call
        # of the one-frame-per-call loop)
        self.opt._pending = (float(self.norm_delta), q[0], last[0], self.last_info["fval"])
        return q[0]

    def get_objective_function(self, ref_value: np.ndarray, fixed_qpos: np.ndarray, last_qpos: np.ndarray):
        """nlopt-style closure evaluated on the GPU (parity hook).  Like the reference, building the closure
        advances the DexPilot projection state once."""
        model = self.device_model()
        ref = np.ascontiguousarray(ref_value, dtype=np.float32).reshape(1, self._compiled.n_ref, 3)
        fixed = np.ascontiguousarray(fixed_qpos, dtype=np.float32).reshape(1, -1)
        last = np.ascontiguousarray(last_qpos, dtype=np.float32).reshape(1, -1)
        state_before = self._state_in(1)
        if state_before is not None:  # advance the state exactly once, evaluate closures against the old one
            st = state_before.copy()
            model.eval(ref, fixed, last, last.astype(np.float64), state=st)
            self._state_out(st)

        def objective(x: np.ndarray, grad: np.ndarray) -> float:
            st = None if state_before is None else state_before.copy()
            f, g = model.eval(ref, fixed, last, np.asarray(x, dtype=np.float64)[None], state=st)
            if grad.size > 0:
                grad[:] = g[0]
            return float(f[0])

        return objective


class PositionOptimizer(Optimizer):
    retargeting_type = "POSITION"

    def __init__(self, robot: RobotWrapper, target_joint_names: List[str], target_link_names: List[str],
                 target_link_human_indices: np.ndarray, huber_delta=0.02, norm_delta=4e-3):
        super().__init__(robot, target_joint_names, target_link_human_indices)
        self.body_names = target_link_names
        self.huber_delta = huber_delta
        self.norm_delta = norm_delta
        self.target_link_indices = self.get_link_indices(target_link_names)
        self.opt.set_ftol_abs(1e-5)

    def _terms(self):
        return [mc.TermSpec(name, None, i) for i, name in enumerate(self.body_names)]

    def _compile_kwargs(self):
        return dict(huber_delta=self.huber_delta, norm_delta=self.norm_delta)


class VectorOptimizer(Optimizer):
    retargeting_type = "VECTOR"

    def __init__(self, robot: RobotWrapper, target_joint_names: List[str], target_origin_link_names: List[str],
                 target_task_link_names: List[str], target_link_human_indices: np.ndarray, huber_delta=0.02,
                 norm_delta=4e-3, scaling=1.0):
        super().__init__(robot, target_joint_names, target_link_human_indices)
        self.origin_link_names = target_origin_link_names
        self.task_link_names = target_task_link_names
        self.huber_delta = huber_delta
        self.norm_delta = norm_delta
        self.scaling = scaling

        # same de-duplicated link cache as optimizer.py:226-237 (order made deterministic)
        self.computed_link_names = list(dict.fromkeys(list(target_origin_link_names) + list(target_task_link_names)))
        self.origin_link_indices = np.array([self.computed_link_names.index(n) for n in target_origin_link_names])
        self.task_link_indices = np.array([self.computed_link_names.index(n) for n in target_task_link_names])
        self.computed_link_indices = self.get_link_indices(self.computed_link_names)
        self.opt.set_ftol_abs(1e-6)

    def _terms(self):
        return [mc.TermSpec(t, o, i) for i, (o, t) in enumerate(zip(self.origin_link_names, self.task_link_names))]

    def _compile_kwargs(self):
        return dict(huber_delta=self.huber_delta, norm_delta=self.norm_delta, scaling=self.scaling)


class DexPilotOptimizer(Optimizer):
    """Retargeting optimizer using the method proposed in DexPilot (https://arxiv.org/abs/1910.03135), in the
    generalised 2..5-finger form of the reference (optimizer.py:309-577)."""

    retargeting_type = "DEXPILOT"

    def __init__(self, robot: RobotWrapper, target_joint_names: List[str], finger_tip_link_names: List[str],
                 wrist_link_name: str, target_link_human_indices: Optional[np.ndarray] = None, huber_delta=0.03,
                 norm_delta=4e-3, project_dist=0.03, escape_dist=0.05, eta1=1e-4, eta2=3e-2, scaling=1.0):
        if len(finger_tip_link_names) < 2 or len(finger_tip_link_names) > 5:
            raise ValueError(f"DexPilot optimizer can only be applied to hands with 2 to 5 fingers, but got "
                             f"{len(finger_tip_link_names)} fingers.")
        self.num_fingers = len(finger_tip_link_names)
        origin_link_index, task_link_index = self.generate_link_indices(self.num_fingers)
        if target_link_human_indices is None:
            target_link_human_indices = (np.stack([origin_link_index, task_link_index], axis=0) * 4).astype(int)
        link_names = [wrist_link_name] + finger_tip_link_names
        target_origin_link_names = [link_names[index] for index in origin_link_index]
        target_task_link_names = [link_names[index] for index in task_link_index]

        super().__init__(robot, target_joint_names, target_link_human_indices)
        self.origin_link_names = target_origin_link_names
        self.task_link_names = target_task_link_names
        self.scaling = scaling
        self.huber_delta = huber_delta
        self.norm_delta = norm_delta
        self.project_dist = project_dist
        self.escape_dist = escape_dist
        self.eta1 = eta1
        self.eta2 = eta2

        self.computed_link_names = list(dict.fromkeys(target_origin_link_names + target_task_link_names))
        self.origin_link_indices = np.array([self.computed_link_names.index(n) for n in target_origin_link_names])
        self.task_link_indices = np.array([self.computed_link_names.index(n) for n in target_task_link_names])
        self.computed_link_indices = self.get_link_indices(self.computed_link_names)
        self.opt.set_ftol_abs(1e-6)

        (self.projected, self.s2_project_index_origin, self.s2_project_index_task,
         self.projected_dist) = self.set_dexpilot_cache(self.num_fingers, eta1, eta2)

    @staticmethod
    def generate_link_indices(num_fingers):
        """
        Example:
        >>> generate_link_indices(4)
        ([2, 3, 4, 3, 4, 4, 0, 0, 0, 0], [1, 1, 1, 2, 2, 3, 1, 2, 3, 4])
        """
        origin_link_index = []
        task_link_index = []
        for i in range(1, num_fingers):
            for j in range(i + 1, num_fingers + 1):
                origin_link_index.append(j)
                task_link_index.append(i)
        for i in range(1, num_fingers + 1):
            origin_link_index.append(0)
            task_link_index.append(i)
        return origin_link_index, task_link_index

    @staticmethod
    def set_dexpilot_cache(num_fingers, eta1, eta2):
        """
        Example:
        >>> set_dexpilot_cache(4, 0.1, 0.2)
        (array([False, False, False, False, False, False]),
        [1, 2, 2],
        [0, 0, 1],
        array([0.1, 0.1, 0.1, 0.2, 0.2, 0.2]))
        """
        projected = np.zeros(num_fingers * (num_fingers - 1) // 2, dtype=bool)
        s2_project_index_origin = []
        s2_project_index_task = []
        for i in range(0, num_fingers - 2):
            for j in range(i + 1, num_fingers - 1):
                s2_project_index_origin.append(j)
                s2_project_index_task.append(i)
        projected_dist = np.array([eta1] * (num_fingers - 1) + [eta2] * ((num_fingers - 1) * (num_fingers - 2) // 2))
        return projected, s2_project_index_origin, s2_project_index_task, projected_dist

    def _terms(self):
        return [mc.TermSpec(t, o, i) for i, (o, t) in enumerate(zip(self.origin_link_names, self.task_link_names))]

    def _compile_kwargs(self):
        return dict(huber_delta=self.huber_delta, norm_delta=self.norm_delta, scaling=self.scaling,
                    num_fingers=self.num_fingers, project_dist=self.project_dist, escape_dist=self.escape_dist,
                    eta1=self.eta1, eta2=self.eta2)

    # the reference keeps `self.projected` (bool array) across frames; the kernel carries it as a bit mask
    def _state_in(self, B: int):
        bits = 0
        for i, b in enumerate(self.projected):
            bits |= int(bool(b)) << i
        return np.full(B, bits, dtype=np.uint32)

    def _state_out(self, state):
        if state is None:
            return
        bits = int(state[0])
        self.projected = np.array([(bits >> i) & 1 for i in range(len(self.projected))], dtype=bool)
