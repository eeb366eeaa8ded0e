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


def world_tracks(B: int, T: int, seed: int = 5):
    """B hand tracks x T frames of (21, 3) WORLD-frame keypoints, the way the reference's offline viewer feeds its position
    configs (/root/reference/example/position_retargeting/hand_robot_viewer.py:143-176: MANO joints in the camera frame),
    + the wrist pose of frame 0 as that viewer hands it to warm_start (wrist_pos = joint[0], wrist_quat = the MANO global
    orientation R_q).  Track b plays the human fixture from a random phase; its (MANO-convention) wrist frame sits in the
    world at rotation R_q @ OPERATOR2MANO[right]^T (seq_retarget.py:70-75) and translation p0_b + t v_b (|v| <= 2 mm/frame).
    Returns kp (T, B, 21, 3) f32, wrist_pos (B, 3) f64, wrist_quat (B, 4) f64 (w, x, y, z)."""
    from dex_retargeting_amd.constants import OPERATOR2MANO, HandType

    rng = np.random.default_rng(seed)
    fixture = np.load(HUMAN_FIXTURE).astype(np.float64)
    op = np.asarray(OPERATOR2MANO[HandType.right], dtype=np.float64)
    q = rng.standard_normal((B, 4))
    q /= np.linalg.norm(q, axis=1, keepdims=True)
    w, x, y, z = q[:, 0], q[:, 1], q[:, 2], q[:, 3]
    Rq = np.stack([np.stack([1 - 2 * (y * y + z * z), 2 * (x * y - z * w), 2 * (x * z + y * w)], -1),
                   np.stack([2 * (x * y + z * w), 1 - 2 * (x * x + z * z), 2 * (y * z - x * w)], -1),
                   np.stack([2 * (x * z - y * w), 2 * (y * z + x * w), 1 - 2 * (x * x + y * y)], -1)], -2)
    Rw = Rq @ op.T                                     # (B, 3, 3)
    p0 = rng.uniform(-0.3, 0.3, (B, 3))
    vel = rng.uniform(-2e-3, 2e-3, (B, 3))
    phase = rng.integers(0, fixture.shape[0], B)
    idx = (phase[None, :] + np.arange(T)[:, None]) % fixture.shape[0]          # (T, B)
    hand = fixture[idx]                                                        # (T, B, 21, 3)
    kp = np.einsum("tbkj,bij->tbki", hand, Rw) + (p0[None] + vel[None] * np.arange(T)[:, None, None])[:, :, None, :]
    return kp.astype(np.float32), p0, q


ARM_HAND_URDF = os.path.join(REPO, "tests", "urdf", "arm_shadow_hand_right.urdf")


def arm_hand_position_config() -> dict:
    """An arm + Shadow hand URDF with 6 dummy free joints under the position objective: 37 movable joints in ONE component,
    21 reference rows (every MANO keypoint matched to a link) -- beyond the fixed-size tables (32 joints, 16 rows), served by
    the general kernel (the same problem as tests/test_generic_tables.arm_hand_config("position"))."""
    tips = ["thtip", "fftip", "mftip", "rftip", "lftip"]
    mid = ["thmiddle", "ffmiddle", "mfmiddle", "rfmiddle", "lfmiddle"]
    prox = ["thproximal", "ffproximal", "mfproximal", "rfproximal", "lfproximal"]
    dist = ["thdistal", "ffdistal", "mfdistal", "rfdistal", "lfdistal"]
    links = ["palm"] + [l for f in range(5) for l in (prox[f], mid[f], dist[f], tips[f])]
    return dict(type="position", urdf_path=ARM_HAND_URDF, add_dummy_free_joint=True, target_link_names=links,
                target_link_human_indices=list(range(21)), low_pass_alpha=1.0)
