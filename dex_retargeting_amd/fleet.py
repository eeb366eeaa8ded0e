"""Mixed-fleet batches (BASELINE.json config 5): frames for several robots / optimizer types in one batch.

One call = ``dexr_retarget_multi_dev`` (include/dexr.h): the frames are bucketed by model ON THE DEVICE (wavefronts must
be model-uniform: the kinematic tables are wave-uniform scalar operands), every model's solve kernel runs over its
index list reading its bucket size from device memory and reading / writing the caller's rows in place.  No gather /
scatter copies, no host synchronisation, no torch index ops.  Every model consumes the same raw input -- 21 hand
keypoints per frame -- and forms its own ``ref_value`` from its ``target_link_human_indices`` inside the kernel.
"""
from __future__ import annotations

from typing import List, Optional, Sequence

from . import _lib
from .optimizer import Optimizer


class MixedFleet:
    def __init__(self, optimizers: Sequence[Optimizer], device: str = "cuda:0"):
        import torch

        self.torch = torch
        self.device = torch.device(device)
        self.optimizers: List[Optimizer] = list(optimizers)
        for o in self.optimizers:
            if len(o.idx_pin2fixed):
                raise ValueError("MixedFleet serves models whose non-target joints are all mimic joints")
        self.models = [o.device_model() for o in self.optimizers]
        self.n_opt = [o.opt_dof for o in self.optimizers]
        self.n_max = max(self.n_opt)
        self._ws = None
        self._opts = None

    def retarget(self, model_id, keypoints, last_qpos, state: Optional["object"] = None, out=None, status=None):
        """model_id (B,) int32, keypoints (B,21,3) f32, last_qpos (B,n_max) f32 (columns >= n_opt[m] ignored),
        state (B,) int32 DexPilot bits (updated in place) or None  ->  (B, n_max) f32; a frame's row holds its model's
        n_opt joints, the remaining columns are zero (or whatever `out` held).  Enqueues on the current stream."""
        torch = self.torch
        B = int(model_id.shape[0])
        if model_id.dtype != torch.int32:
            model_id = model_id.to(torch.int32)
        for t, name in ((model_id, "model_id"), (keypoints, "keypoints"), (last_qpos, "last_qpos")):
            if not t.is_contiguous() or t.device != self.device:
                raise ValueError(f"{name} must be a contiguous tensor on {self.device}")
        if keypoints.dtype != torch.float32 or last_qpos.dtype != torch.float32 or tuple(last_qpos.shape) != (B, self.n_max):
            raise ValueError(f"keypoints / last_qpos must be float32, last_qpos of shape ({B}, {self.n_max})")
        if out is None:
            out = torch.zeros((B, self.n_max), dtype=torch.float32, device=self.device)
        need = _lib.fleet_workspace_bytes(B)
        if self._ws is None or self._ws.numel() < need:
            self._ws = torch.empty(need, dtype=torch.uint8, device=self.device)
        _lib.retarget_multi_dev(self.models, B, model_id.data_ptr(), keypoints.data_ptr(), last_qpos.data_ptr(), self.n_max,
                                state.data_ptr() if state is not None else 0, out.data_ptr(),
                                status.data_ptr() if status is not None else 0, self._ws.data_ptr(), self._ws.numel(),
                                opts=self._opts, stream=torch.cuda.current_stream(self.device).cuda_stream)
        return out
