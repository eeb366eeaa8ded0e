"""Multi-GPU sharding of a batch (one process per GPU, torch.distributed; backend "nccl" is RCCL on ROCm).

Every frame is an independent solve, so a batch shards with no data-path exchange: rank r solves the contiguous
slice [lo_r, hi_r) and ONE all-gather reassembles the (B, n_opt) qpos tensor on every rank (BASELINE.json
north_star; the reference has no distributed mode)."""
from __future__ import annotations

from typing import Callable, Optional, Tuple

import numpy as np


def shard_bounds(B: int, rank: int, world: int) -> Tuple[int, int]:
    """Contiguous balanced slices: the first B % world ranks get one extra item."""
    if not (0 <= rank < world):
        raise ValueError(f"rank {rank} outside world of {world}")
    base, extra = divmod(B, world)
    lo = rank * base + min(rank, extra)
    return lo, lo + base + (1 if rank < extra else 0)


class ShardedRetargeter:
    """retarget(ref, fixed, last[, state]) -> full (B, n_opt) float32 on every rank.

    `solve` is the per-shard solver (default: ``optimizer.retarget_batch`` = the HIP path).  Tensors live on
    `device` for the collective ("cuda:<local_rank>" with nccl/RCCL, "cpu" with gloo)."""

    def __init__(self, optimizer=None, solve: Optional[Callable] = None, device: str = "cpu", group=None):
        import torch.distributed as dist

        if not dist.is_initialized():
            raise RuntimeError("torch.distributed is not initialised")
        self.dist = dist
        self.group = group
        self.rank = dist.get_rank(group)
        self.world = dist.get_world_size(group)
        self.device = device
        self.solve = solve if solve is not None else optimizer.retarget_batch

    def retarget(self, ref: np.ndarray, fixed: Optional[np.ndarray], last: np.ndarray,
                 state: Optional[np.ndarray] = None) -> np.ndarray:
        import torch

        B, n_opt = last.shape
        lo, hi = shard_bounds(B, self.rank, self.world)
        st = None if state is None else state[lo:hi]
        q = self.solve(ref[lo:hi], None if fixed is None else fixed[lo:hi], last[lo:hi], st)
        if state is not None:
            state[lo:hi] = st
        per = -(-B // self.world)  # equal-size slots for the all-gather (last slots padded)
        mine = torch.zeros((per, n_opt), dtype=torch.float32, device=self.device)
        if hi > lo:
            mine[: hi - lo] = torch.from_numpy(np.ascontiguousarray(q, dtype=np.float32)).to(self.device)
        full = torch.empty((self.world * per, n_opt), dtype=torch.float32, device=self.device)
        self.dist.all_gather_into_tensor(full, mine, group=self.group)
        full = full.cpu().numpy().reshape(self.world, per, n_opt)
        out = np.empty((B, n_opt), dtype=np.float32)
        for r in range(self.world):
            a, b = shard_bounds(B, r, self.world)
            out[a:b] = full[r, : b - a]
        if state is not None:  # DexPilot bits travel the same way
            smine = torch.zeros(per, dtype=torch.int64, device=self.device)
            smine[: hi - lo] = torch.from_numpy(state[lo:hi].astype(np.int64)).to(self.device)
            sfull = torch.empty(self.world * per, dtype=torch.int64, device=self.device)
            self.dist.all_gather_into_tensor(sfull, smine, group=self.group)
            sfull = sfull.cpu().numpy().reshape(self.world, per)
            for r in range(self.world):
                a, b = shard_bounds(B, r, self.world)
                state[a:b] = sfull[r, : b - a].astype(np.uint32)
        return out
