"""Seeded synthetic inputs of the measurement harness (bench.py, tools/): NOT part of the product and not part of the
oracle -- just the workload recipes of SURVEY.md section 8d, which mirror the reference's own profiling script
(/root/reference/example/profiling/profile_online_retargeting.py:18-36: 21 hand keypoints per frame taken from the
reference's only data fixture, a float32 copy of which lives in tests/golden/, plus Gaussian noise) and the reference's
own test fixture for cold starts (/root/reference/tests/test_optimizer.py:27-81: random joint positions inside the
limits, targets = forward kinematics of them, start = another random draw)."""
import os

import numpy as np

REPO = os.path.dirname(os.path.abspath(__file__))
CONFIG_DIR = os.path.join(REPO, "dex_retargeting_amd", "configs")
HUMAN_FIXTURE = os.path.join(REPO, "tests", "golden", "human_joint_right_f32.npy")
SEED = 20250614


def human_keypoints(B: int, seed: int = SEED, noise: float = 2e-3, offset: int = 0) -> np.ndarray:
    """(B,21,3) f32: fixture frame (offset + b) mod 621 + N(0, 2 mm), wrist kept at the origin."""
    kp = np.load(HUMAN_FIXTURE)
    rng = np.random.default_rng(seed + 1)
    out = kp[(np.arange(B) + offset) % kp.shape[0]].astype(np.float64)
    if noise > 0:
        out = out + noise * rng.standard_normal(out.shape)
        out[:, 0] = 0.0
    return out.astype(np.float32)


def reachable_batch(seq, B: int, sigma: float, seed: int = SEED):
    """Cold-start regime of the reference's own tests (tests/test_optimizer.py:27-81): q* ~ U(lo, hi) per frame (mimic
    joints forwarded), ref_value = the robot's own link vectors / positions at q* (divided by the scaling factor so
    that ref * scaling is reachable), start = clip(q* + sigma N(0,1), limits).  The forward kinematics is the product's
    own (RobotWrapper.link_positions -> dexr_fk on the GPU): no oracle code is involved.
    Returns ref (B,n_ref,3) f32, last (B,n_opt) f32."""
    opt = seq.optimizer
    robot = opt.robot
    rng = np.random.default_rng(seed)
    lim = robot.joint_limits
    q = rng.uniform(lim[:, 0], lim[:, 1], size=(B, robot.dof))
    if opt.adaptor is not None:
        q = opt.adaptor.forward_qpos(q)
    start = np.clip(q + sigma * rng.standard_normal(q.shape), lim[:, 0] + 1e-5, lim[:, 1] - 1e-5)
    kind = opt.retargeting_type
    if kind == "POSITION":
        ref = robot.link_positions(q, opt.target_link_indices)
    else:
        pos = robot.link_positions(q, opt.computed_link_indices)
        ref = (pos[:, opt.task_link_indices] - pos[:, opt.origin_link_indices]) / float(opt.scaling)
    return (np.ascontiguousarray(ref, dtype=np.float32),
            np.ascontiguousarray(start[:, opt.idx_pin2target], dtype=np.float32))
