"""RetargetingConfig: same dataclass fields, validation, loaders and ``build()`` factory as the reference
(/root/reference/src/dex_retargeting/retargeting_config.py:18-285).  ``build()`` returns a ``SeqRetargeting`` whose
optimizer runs on the GPU; ``build_batched(B)`` returns the lock-step batched form."""
from __future__ import annotations

from dataclasses import dataclass
from pathlib import Path
from typing import Any, Dict, List, Optional, Tuple, Union

import numpy as np
import yaml

from .kinematics_adaptor import MimicJointKinematicAdaptor
from .optimizer_utils import LPFilter
from .robot_wrapper import RobotWrapper
from .seq_retarget import BatchedSeqRetargeting, SeqRetargeting
from .urdf import DUMMY_JOINT_NAMES


@dataclass
class RetargetingConfig:
    type: str
    urdf_path: str

    add_dummy_free_joint: bool = False
    target_link_human_indices: Optional[np.ndarray] = None
    wrist_link_name: Optional[str] = None
    target_link_names: Optional[List[str]] = None
    target_joint_names: Optional[List[str]] = None
    target_origin_link_names: Optional[List[str]] = None
    target_task_link_names: Optional[List[str]] = None
    finger_tip_link_names: Optional[List[str]] = None
    scaling_factor: float = 1.0
    normal_delta: float = 4e-3
    huber_delta: float = 2e-2
    project_dist: float = 0.03
    escape_dist: float = 0.05
    has_joint_limits: bool = True
    ignore_mimic_joint: bool = False
    low_pass_alpha: float = 0.1

    _TYPE = ["vector", "position", "dexpilot"]
    _DEFAULT_URDF_DIR = "./"

    def __post_init__(self):
        self.type = self.type.lower()
        if self.type not in self._TYPE:
            raise ValueError(f"Retargeting type must be one of {self._TYPE}")

        if self.type == "vector":
            if self.target_origin_link_names is None or self.target_task_link_names is None:
                raise ValueError("Vector retargeting requires: target_origin_link_names + target_task_link_names")
            if len(self.target_task_link_names) != len(self.target_origin_link_names):
                raise ValueError("Vector retargeting origin and task links dim mismatch")
            if self.target_link_human_indices is None:
                raise ValueError("Vector retargeting requires: target_link_human_indices")
            if self.target_link_human_indices.shape != (2, len(self.target_origin_link_names)):
                raise ValueError("Vector retargeting link names and link indices dim mismatch")
        elif self.type == "position":
            if self.target_link_names is None:
                raise ValueError("Position retargeting requires: target_link_names")
            if self.target_link_human_indices is None:
                raise ValueError("Position retargeting requires: target_link_human_indices")
            self.target_link_human_indices = self.target_link_human_indices.squeeze()
            if self.target_link_human_indices.shape != (len(self.target_link_names),):
                raise ValueError("Position retargeting link names and link indices dim mismatch")
        elif self.type == "dexpilot":
            if self.finger_tip_link_names is None or self.wrist_link_name is None:
                raise ValueError("Position retargeting requires: finger_tip_link_names + wrist_link_name")
            if self.target_link_human_indices is not None:
                print("\033[33m",
                      "Target link human indices is provided in the DexPilot retargeting config, which is uncommon.\n"
                      "If you do not know exactly how it is used, please leave it to None for default.\n"
                      "\033[00m")

        urdf_path = Path(self.urdf_path)
        if not urdf_path.is_absolute():
            urdf_path = Path(self._DEFAULT_URDF_DIR) / urdf_path
            urdf_path = urdf_path.absolute()
        if not urdf_path.exists():
            raise ValueError(f"URDF path {urdf_path} does not exist")
        self.urdf_path = str(urdf_path)

    @classmethod
    def set_default_urdf_dir(cls, urdf_dir: Union[str, Path]):
        path = Path(urdf_dir)
        if not path.exists():
            raise ValueError(f"URDF dir {urdf_dir} not exists.")
        cls._DEFAULT_URDF_DIR = urdf_dir

    @classmethod
    def load_from_file(cls, config_path: Union[str, Path], override: Optional[Dict] = None):
        path = Path(config_path)
        if not path.is_absolute():
            path = path.absolute()
        with path.open("r") as f:
            yaml_config = yaml.load(f, Loader=yaml.FullLoader)
            cfg = yaml_config["retargeting"]
            return cls.from_dict(cfg, override)

    @classmethod
    def from_dict(cls, cfg: Dict[str, Any], override: Optional[Dict] = None):
        if "target_link_human_indices" in cfg:
            cfg["target_link_human_indices"] = np.array(cfg["target_link_human_indices"])
        if override is not None:
            for key, value in override.items():
                cfg[key] = value
        config = RetargetingConfig(**cfg)
        return config

    # ------------------------------------------------------------------------------------------------
    def _build_optimizer(self):
        from .optimizer import DexPilotOptimizer, PositionOptimizer, VectorOptimizer

        # The reference round-trips the URDF through yourdfpy into a temp file for pinocchio
        # (retargeting_config.py:175-187); here the same URDF is flattened straight into kinematic tables.
        robot = RobotWrapper(self.urdf_path, add_dummy_free_joints=self.add_dummy_free_joint)

        if self.add_dummy_free_joint and self.target_joint_names is not None:
            self.target_joint_names = DUMMY_JOINT_NAMES + self.target_joint_names
        joint_names = self.target_joint_names if self.target_joint_names is not None else robot.dof_joint_names

        if self.type == "position":
            optimizer = PositionOptimizer(robot, joint_names, target_link_names=self.target_link_names,
                                          target_link_human_indices=self.target_link_human_indices,
                                          norm_delta=self.normal_delta, huber_delta=self.huber_delta)
        elif self.type == "vector":
            optimizer = VectorOptimizer(robot, joint_names, target_origin_link_names=self.target_origin_link_names,
                                        target_task_link_names=self.target_task_link_names,
                                        target_link_human_indices=self.target_link_human_indices,
                                        scaling=self.scaling_factor, norm_delta=self.normal_delta,
                                        huber_delta=self.huber_delta)
        elif self.type == "dexpilot":
            # NOTE: like the reference (retargeting_config.py:218-228) huber/normal delta are NOT forwarded here
            optimizer = DexPilotOptimizer(robot, joint_names, finger_tip_link_names=self.finger_tip_link_names,
                                          wrist_link_name=self.wrist_link_name,
                                          target_link_human_indices=self.target_link_human_indices,
                                          scaling=self.scaling_factor, project_dist=self.project_dist,
                                          escape_dist=self.escape_dist)
        else:
            raise RuntimeError()

        has_mimic_joints, source_names, mimic_names, multipliers, offsets = parse_mimic_joint(robot)
        if has_mimic_joints and not self.ignore_mimic_joint:
            adaptor = MimicJointKinematicAdaptor(robot, target_joint_names=joint_names,
                                                 source_joint_names=source_names, mimic_joint_names=mimic_names,
                                                 multipliers=multipliers, offsets=offsets)
            optimizer.set_kinematic_adaptor(adaptor)
        return optimizer

    def build(self) -> SeqRetargeting:
        optimizer = self._build_optimizer()
        lp_filter = LPFilter(self.low_pass_alpha) if 0 <= self.low_pass_alpha <= 1 else None
        return SeqRetargeting(optimizer, has_joint_limits=self.has_joint_limits, lp_filter=lp_filter)

    def build_batched(self, batch: int) -> BatchedSeqRetargeting:
        optimizer = self._build_optimizer()
        alpha = self.low_pass_alpha if 0 <= self.low_pass_alpha <= 1 else None
        return BatchedSeqRetargeting(optimizer, batch, has_joint_limits=self.has_joint_limits, low_pass_alpha=alpha)


    def build_device(self, batch: int, device: str = "cuda:0"):
        """B lock-step sequences with all per-sequence state resident on the GPU (torch tensors in / out)."""
        from .device_seq import DeviceSeqRetargeting

        optimizer = self._build_optimizer()
        alpha = self.low_pass_alpha if 0 <= self.low_pass_alpha <= 1 else None
        return DeviceSeqRetargeting(optimizer, batch, has_joint_limits=self.has_joint_limits, low_pass_alpha=alpha,
                                    device=device)


def get_retargeting_config(config_path: Union[str, Path]) -> RetargetingConfig:
    return RetargetingConfig.load_from_file(config_path)


def parse_mimic_joint(robot: RobotWrapper) -> Tuple[bool, List[str], List[str], List[float], List[float]]:
    source_joint_names, mimic_joint_names, multipliers, offsets = robot.kin.mimic_joints()
    return len(mimic_joint_names) > 0, source_joint_names, mimic_joint_names, multipliers, offsets
