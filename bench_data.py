"""Seeded synthetic inputs of the measurement harness (bench.py, tools/): NOT part of the product and not part of the
oracle -- just the workload recipe of SURVEY.md section 8d, which mirrors the reference's own profiling script
(/root/reference/example/profiling/profile_online_retargeting.py:18-36): 21 hand keypoints per frame taken from the
reference's only data fixture (a float32 copy lives in tests/golden/) plus Gaussian noise."""
import os

import numpy as np

REPO = os.path.dirname(os.path.abspath(__file__))
CONFIG_DIR = os.path.join(REPO, "dex_retargeting_amd", "configs")
HUMAN_FIXTURE = os.path.join(REPO, "tests", "golden", "human_joint_right_f32.npy")
SEED = 20250614


def human_keypoints(B: int, seed: int = SEED, noise: float = 2e-3) -> np.ndarray:
    """(B,21,3) f32: fixture frame b mod 621 + N(0, 2 mm), wrist kept at the origin."""
    kp = np.load(HUMAN_FIXTURE)
    rng = np.random.default_rng(seed + 1)
    out = kp[np.arange(B) % kp.shape[0]].astype(np.float64)
    if noise > 0:
        out = out + noise * rng.standard_normal(out.shape)
        out[:, 0] = 0.0
    return out.astype(np.float32)
