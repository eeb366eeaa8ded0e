"""Raw detector keypoints -> MANO-frame joint positions, batched on the GPU.

The reference does this per frame on the host inside its MediaPipe wrapper
(example/vector_retargeting/single_hand_detector.py:102-104):

    keypoint_3d_array = keypoint_3d_array - keypoint_3d_array[0:1, :]
    mediapipe_wrist_rot = self.estimate_frame_from_hand_points(keypoint_3d_array)     # :129-158
    joint_pos = keypoint_3d_array @ mediapipe_wrist_rot @ self.operator2mano

Here the same three lines run for B frames in one HBM-bound HIP kernel (csrc/dexr_prep.hip); its output feeds
``dexr_retarget_kp_dev`` without leaving the device.
"""
from __future__ import annotations

import ctypes as C
from typing import Tuple, Union

import numpy as np

from . import _lib
from .constants import OPERATOR2MANO, HandType

N_KEYPOINTS = 21


def _operator2mano(hand_type: Union[str, HandType]) -> np.ndarray:
    if isinstance(hand_type, str):
        key = hand_type.lower()
        if key not in ("right", "left"):
            raise ValueError(f"hand_type must be 'Right' or 'Left', got {hand_type!r}")
        hand_type = HandType.right if key == "right" else HandType.left
    return np.ascontiguousarray(OPERATOR2MANO[hand_type], dtype=np.float32)


def mano_keypoints(keypoints: np.ndarray, hand_type: Union[str, HandType] = "Right") -> Tuple[np.ndarray, np.ndarray]:
    """keypoints: (B, 21, 3) raw detector output.  Returns (joint_pos (B,21,3) f32, wrist_rot (B,3,3) f32) -- the
    ``joint_pos`` / ``mediapipe_wrist_rot`` pair ``SingleHandDetector.detect`` returns, for every frame."""
    kp = np.ascontiguousarray(keypoints, dtype=np.float32)
    if kp.ndim != 3 or kp.shape[1:] != (N_KEYPOINTS, 3):
        raise ValueError(f"keypoints must have shape (B, {N_KEYPOINTS}, 3), got {kp.shape}")
    op = _operator2mano(hand_type)
    out = np.empty_like(kp)
    rot = np.empty((kp.shape[0], 3, 3), dtype=np.float32)
    f32p = C.POINTER(C.c_float)
    _lib.check(_lib.load().dexr_mano_keypoints(kp.shape[0], kp.ctypes.data_as(f32p), op.ctypes.data_as(f32p),
                                               out.ctypes.data_as(f32p), rot.ctypes.data_as(f32p)))
    return out, rot


def mano_keypoints_dev(batch: int, keypoints_ptr: int, joint_pos_ptr: int, wrist_rot_ptr: int = 0,
                       hand_type: Union[str, HandType] = "Right", stream: int = 0) -> None:
    """Device-pointer variant: enqueue on ``stream`` and return (no synchronisation)."""
    op = _operator2mano(hand_type)
    _lib.check(_lib.load().dexr_mano_keypoints_dev(int(batch), C.c_void_p(keypoints_ptr),
                                                   op.ctypes.data_as(C.POINTER(C.c_float)),
                                                   C.c_void_p(joint_pos_ptr), C.c_void_p(wrist_rot_ptr or None),
                                                   C.c_void_p(stream or None)))
