"""ORACLE (test infrastructure only): problem construction from the YAML specs + seeded synthetic inputs.

Input recipes follow SURVEY.md section 8d, which mirrors the reference's own test fixtures:
``sample_qpos`` (/root/reference/tests/test_optimizer.py:27-42) for the reachable sets and
``profile_retargeting`` (/root/reference/example/profiling/profile_online_retargeting.py:18-36) for the
human-keypoint set.
"""
from __future__ import annotations

import os
from typing import Optional

import numpy as np
import yaml

from .kin import OracleRobot
from .objectives import OracleProblem

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CONFIG_DIR = os.path.join(REPO, "dex_retargeting_amd", "configs")
URDF_DIR = os.path.join(REPO, "dex_retargeting_amd", "assets", "robots", "hands")
HUMAN_FIXTURE = os.path.join(REPO, "tests", "golden", "human_joint_right_f32.npy")
SEED = 20250614


def load_cfg(rel: str) -> dict:
    with open(os.path.join(CONFIG_DIR, rel)) as f:
        return yaml.safe_load(f)["retargeting"]


def problem_from_config(rel: str, **override) -> OracleProblem:
    """rel e.g. 'teleop/allegro_hand_right.yml'.  Mirrors RetargetingConfig.build()
    (/root/reference/src/dex_retargeting/retargeting_config.py:167-257) incl. quirk Q2 (DexPilot ignores the
    config's huber/normal delta)."""
    cfg = load_cfg(rel)
    cfg.update(override)
    kind = cfg["type"].lower()
    free = bool(cfg.get("add_dummy_free_joint", False))
    robot = OracleRobot(os.path.join(URDF_DIR, cfg["urdf_path"]), add_dummy_free_joints=free)
    tj = cfg.get("target_joint_names")
    if free and tj is not None:
        tj = [f"dummy_{n}_translation_joint" for n in "xyz"] + [f"dummy_{n}_rotation_joint" for n in "xyz"] + tj
    common = dict(use_mimic=not cfg.get("ignore_mimic_joint", False),
                  has_joint_limits=cfg.get("has_joint_limits", True))
    if kind == "position":
        p = OracleProblem(robot, kind, tj, target_link_names=cfg["target_link_names"],
                          huber_delta=cfg.get("huber_delta", 0.02), norm_delta=cfg.get("normal_delta", 4e-3), **common)
        p.target_link_human_indices = np.array(cfg["target_link_human_indices"]).squeeze()
    elif kind == "vector":
        p = OracleProblem(robot, kind, tj, target_origin_link_names=cfg["target_origin_link_names"],
                          target_task_link_names=cfg["target_task_link_names"],
                          huber_delta=cfg.get("huber_delta", 0.02), norm_delta=cfg.get("normal_delta", 4e-3),
                          scaling=cfg.get("scaling_factor", 1.0), **common)
        p.target_link_human_indices = np.array(cfg["target_link_human_indices"])
    else:
        p = OracleProblem(robot, kind, tj, wrist_link_name=cfg["wrist_link_name"],
                          finger_tip_link_names=cfg["finger_tip_link_names"], scaling=cfg.get("scaling_factor", 1.0),
                          project_dist=cfg.get("project_dist", 0.03), escape_dist=cfg.get("escape_dist", 0.05),
                          **common)
    p.cfg = cfg
    p.low_pass_alpha = cfg.get("low_pass_alpha", 0.1)
    return p


def fk_reference_values(prob: OracleProblem, q_full: np.ndarray) -> np.ndarray:
    """Targets that are exactly reachable at q_full (tests/test_optimizer.py:56-81), BEFORE scaling."""
    pos = prob.robot.link_positions(q_full, prob.computed_links)
    if prob.kind == "position":
        return pos
    return pos[:, prob.task_idx] - pos[:, prob.origin_idx]


def reachable_set(prob: OracleProblem, B: int, sigma: float, seed: int = SEED, divide_scaling: bool = True):
    """q* ~ U(lo,hi) (mimic-forwarded); ref = FK-derived values (divided by scaling so that ref*s is reachable);
    last = clip(q* + sigma N(0,1), lo+1e-5, hi-1e-5).  Returns dict(ref f32 (B,n_ref,3), fixed f32, last f32,
    q_star (B,n_opt))."""
    rng = np.random.default_rng(seed)
    lim = prob.robot.joint_limits
    q = rng.uniform(lim[:, 0], lim[:, 1], size=(B, prob.robot.dof))
    if len(prob.mimic):
        q = prob.robot.mimic_forward(q)
    init = np.clip(q + sigma * rng.standard_normal(q.shape), lim[:, 0] + 1e-5, lim[:, 1] - 1e-5)
    ref = fk_reference_values(prob, q)
    if prob.kind != "position" and divide_scaling:
        ref = ref / prob.scaling
    c = np.ascontiguousarray  # fancy indexing may hand back F-ordered memory; device pointers need C order
    return dict(ref=c(ref, dtype=np.float32), fixed=c(q[:, prob.idx_pin2fixed], dtype=np.float32),
                last=c(init[:, prob.idx_pin2target], dtype=np.float32), q_star=c(q[:, prob.idx_pin2target]))


def human_keypoints(B: int, seed: int = SEED, noise: float = 2e-3) -> np.ndarray:
    """(B,21,3) f32: fixture frame b mod 621 + N(0, 2 mm).  One definition, shared with the
    measurement harness (bench_data.py) so that tests and bench feed identical frames."""
    import sys

    if REPO not in sys.path:
        sys.path.insert(0, REPO)
    import bench_data

    return bench_data.human_keypoints(B, seed, noise)


def ref_from_keypoints(prob: OracleProblem, kp: np.ndarray) -> np.ndarray:
    """profile_online_retargeting.py:24-30."""
    idx = prob.target_link_human_indices
    if prob.kind == "position":
        return kp[:, idx, :]
    return kp[:, idx[1], :] - kp[:, idx[0], :]


def human_set(prob: OracleProblem, B: int, seed: int = SEED, sigma: float = 0.05, last: Optional[np.ndarray] = None):
    """Human-keypoint references with a 'tracking' start: last = a tight solution neighbourhood is not
    available a priori, so start from the limit midpoint (seq_retarget.py:33-35) plus sigma noise."""
    kp = human_keypoints(B, seed)
    ref = ref_from_keypoints(prob, kp)
    lim = prob.joint_limits
    rng = np.random.default_rng(seed + 2)
    if last is None:
        mid = lim.mean(1)[None].repeat(B, 0)
        last = np.clip(mid + sigma * rng.standard_normal(mid.shape), lim[:, 0], lim[:, 1])
    fixed = np.zeros((B, len(prob.idx_pin2fixed)), dtype=np.float32)
    return dict(ref=np.ascontiguousarray(ref, dtype=np.float32), fixed=fixed,
                last=np.ascontiguousarray(last, dtype=np.float32), kp=kp)
