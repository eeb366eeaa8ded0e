"""ORACLE (test infrastructure only -- never imported by the product path).

CPU restatement, in float64 numpy, of the keypoint pre-processing the reference's MediaPipe wrapper applies before
retargeting (example/vector_retargeting/single_hand_detector.py):

* :func:`estimate_frame_from_hand_points`  -- :129-158, SVD plane fit through keypoints (0, 5, 9) + Gram-Schmidt;
* :func:`mano_joint_pos`                   -- :102-104, centre on the wrist, rotate into the MANO frame.

Pinned against the reference's own static method by tests/golden/gen_golden.py -> tests/golden/mano_frame_golden.npz.
"""
from __future__ import annotations

import numpy as np

# constants.py:7-21 (also single_hand_detector.py:9-23)
OPERATOR2MANO_RIGHT = np.array([[0, 0, -1], [-1, 0, 0], [0, 1, 0]], dtype=np.float64)
OPERATOR2MANO_LEFT = np.array([[0, 0, -1], [1, 0, 0], [0, -1, 0]], dtype=np.float64)


def estimate_frame_from_hand_points(kp: np.ndarray) -> np.ndarray:
    """kp: (21, 3) wrist-centred keypoints -> (3, 3) wrist frame, columns (x, normal, z)."""
    assert kp.shape == (21, 3)
    pts = kp[[0, 5, 9], :].astype(np.float64)
    x_vec = pts[0] - pts[2]                                    # :140
    pts = pts - pts.mean(axis=0, keepdims=True)                # :143
    _, _, v = np.linalg.svd(pts)                               # :144
    normal = v[2, :].copy()                                    # :146
    x = x_vec - np.sum(x_vec * normal) * normal                # :149
    x = x / np.linalg.norm(x)
    z = np.cross(x, normal)                                    # :151
    if np.sum(z * (pts[1] - pts[2])) < 0:                      # :154-156
        normal *= -1
        z *= -1
    return np.stack([x, normal, z], axis=1)                    # :157


def mano_joint_pos(kp_raw: np.ndarray, right: bool = True):
    """kp_raw: (B, 21, 3) raw detector keypoints -> (joint_pos (B,21,3), wrist_rot (B,3,3)), float64."""
    op = OPERATOR2MANO_RIGHT if right else OPERATOR2MANO_LEFT
    kp_raw = np.asarray(kp_raw, dtype=np.float64)
    out = np.empty_like(kp_raw)
    rot = np.empty((kp_raw.shape[0], 3, 3))
    for b in range(kp_raw.shape[0]):
        c = kp_raw[b] - kp_raw[b, 0:1, :]                      # :102
        rot[b] = estimate_frame_from_hand_points(c)            # :103
        out[b] = c @ rot[b] @ op                               # :104
    return out, rot
