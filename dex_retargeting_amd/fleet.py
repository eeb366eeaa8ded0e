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
        self.n_fixed = [len(o.idx_pin2fixed) for o in self.optimizers]
        self.n_fixed_max = max(self.n_fixed)
        self.models = [o.device_model() for o in self.optimizers]
        self.n_opt = [o.opt_dof for o in self.optimizers]
        self.n_max = max(self.n_opt)
        self._ws = None
        # dexr_retarget_multi_dev takes ONE dexr_solve_options for the whole batch: a fleet whose optimizers carry
        # different solve_options would silently solve some of them differently from their own retarget_batch
        so = [dict(o.solve_options) for o in self.optimizers]
        if any(x != so[0] for x in so[1:]):
            raise ValueError("MixedFleet: the optimizers carry different solve_options; a fleet batch is solved with one "
                             "set of options (set them equal, or run the models in separate calls)")
        self._opts = self.optimizers[0]._options()

    def retarget(self, model_id, keypoints, last_qpos, state: Optional["object"] = None, out=None, status=None, fixed=None):
        """model_id (B,) int32, keypoints (B,21,3) f32, last_qpos (B,n_max) f32 (columns >= n_opt[m] ignored),
        state (B,) int32 DexPilot bits (updated in place) or None, fixed (B, n_fixed_max) f32 caller-supplied fixed-joint
        values (a frame's row holds its model's fixed_qpos; required when a model has fixed joints)  ->  (B, n_max) f32; a frame's row holds its model's
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
        for t, name in ((state, "state"), (status, "status")):  # raw pointers go to the kernels: insist on the layout
            if t is not None and (t.dtype not in (torch.int32, torch.uint32) or tuple(t.shape) != (B,) or
                                  not t.is_contiguous() or t.device != self.device):
                raise ValueError(f"{name} must be a contiguous int32 tensor of shape ({B},) on {self.device}")
        if state is None and any(o.retargeting_type == "DEXPILOT" for o in self.optimizers):
            raise ValueError("the fleet contains a DexPilot model: state (B,) int32 is required")
        if self.n_fixed_max:
            if fixed is None or fixed.dtype != torch.float32 or not fixed.is_contiguous() or fixed.device != self.device or \
                    fixed.dim() != 2 or fixed.shape[0] != B or fixed.shape[1] < self.n_fixed_max:
                raise ValueError(f"fixed must be a contiguous float32 tensor of shape ({B}, >= {self.n_fixed_max}) on {self.device}")
        if out is None:
            out = torch.zeros((B, self.n_max), dtype=torch.float32, device=self.device)
        elif out.dtype != torch.float32 or tuple(out.shape) != (B, self.n_max) or not out.is_contiguous() or out.device != self.device:
            raise ValueError(f"out must be a contiguous float32 tensor of shape ({B}, {self.n_max}) on {self.device}")
        need = _lib.fleet_workspace_bytes(B)
        if self._ws is None or self._ws.numel() < need:
            self._ws = torch.empty(need, dtype=torch.uint8, device=self.device)
        _lib.retarget_multi_dev(self.models, B, model_id.data_ptr(), keypoints.data_ptr(), last_qpos.data_ptr(), self.n_max,
                                state.data_ptr() if state is not None else 0, out.data_ptr(),
                                status.data_ptr() if status is not None else 0, self._ws.data_ptr(), self._ws.numel(),
                                opts=self._opts, stream=torch.cuda.current_stream(self.device).cuda_stream,
                                fixed_ptr=fixed.data_ptr() if self.n_fixed_max else 0,
                                ld_fixed=int(fixed.shape[1]) if self.n_fixed_max else 0)
        return out
