"""Mixed-fleet batches (BASELINE.json config 5): frames for several robots / optimizer types in one batch.

Items are bucketed by model so that every wavefront stays model-uniform (kinematic tables are wave-uniform scalar
loads); each model's bucket is solved by its own ``dexr_retarget_kp_dev`` enqueue on its own HIP stream, so the
kernels of different models overlap on the GPU; results are scattered back into the caller's order.  All tensors stay
on the device.  Every model consumes the same raw input -- 21 hand keypoints per frame -- and forms its own
``ref_value`` from its ``target_link_human_indices`` inside the kernel.
"""
from __future__ import annotations

from typing import List, Optional, Sequence

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
        self.streams = [torch.cuda.Stream(device=self.device) for _ in self.optimizers]
        self._opts = [o._options() for o in self.optimizers]

    def retarget(self, model_id, keypoints, last_qpos, state: Optional["object"] = None):
        """model_id (B,) int64, keypoints (B,21,3) f32, last_qpos (B,n_max) f32 (columns >= n_opt[m] ignored),
        state (B,) int32 DexPilot bits (updated in place) or None  ->  (B, n_max) f32, zero-padded per model."""
        torch = self.torch
        B = model_id.shape[0]
        out = torch.zeros((B, self.n_max), dtype=torch.float32, device=self.device)
        cur = torch.cuda.current_stream(self.device)
        keep = []  # keep the gathered buffers alive until the side streams are done
        for m, (model, n, st) in enumerate(zip(self.models, self.n_opt, self.streams)):
            idx = torch.nonzero(model_id == m, as_tuple=False).squeeze(1)
            if idx.numel() == 0:
                continue
            kp_m = keypoints.index_select(0, idx).contiguous()
            last_m = last_qpos.index_select(0, idx)[:, :n].contiguous()
            q_m = torch.empty_like(last_m)
            dex = self.optimizers[m].retargeting_type == "DEXPILOT"
            s_m = state.index_select(0, idx).contiguous() if (dex and state is not None) else None
            st.wait_stream(cur)
            with torch.cuda.stream(st):
                model.retarget_dev(int(idx.numel()), kp_m.data_ptr(), 0, last_m.data_ptr(),
                                   s_m.data_ptr() if s_m is not None else 0, q_m.data_ptr(), opts=self._opts[m],
                                   stream=st.cuda_stream, keypoints=True)
                out[idx, :n] = q_m
                if s_m is not None:
                    state[idx] = s_m
            keep.append((kp_m, last_m, q_m, s_m, idx))
        for st in self.streams:
            cur.wait_stream(st)
        self._keep = keep
        return out
