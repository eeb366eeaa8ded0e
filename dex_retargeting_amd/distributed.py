"""Multi-GPU sharding of a batch (one process per GPU, torch.distributed; backend "nccl" is RCCL on ROCm).

Every frame is an independent solve, so a batch shards with no data-path exchange: rank r solves the contiguous
slice [lo_r, hi_r) and ONE all-gather reassembles the (B, n_opt) qpos tensor on every rank (BASELINE.json
north_star; the reference has no distributed mode).

Two ways to issue that collective:

* ``NativeGather`` -- the data path: ``dexr_allgather`` of the C-ABI (RCCL bound inside libdexr.so), enqueued on a HIP
  stream right behind the solve, no host round trip, hipGraph-capturable.  ``native_comm`` builds the communicator; the
  128-byte RCCL unique id travels through a ``torch.distributed`` TCP store (the one ``torch.distributed.run`` already
  hosts, when launched by it).
* ``ShardedRetargeter`` / ``PipelinedAllGather`` -- the same partition over ``torch.distributed`` process groups (gloo on
  CPU: what the world-size-2 tests run; nccl = RCCL through torch)."""
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


def shard_by_model(model_id, world: int, n_models: Optional[int] = None):
    """Skew-proof partition of a mixed-fleet batch ("bucket by model id first, then slice each bucket
    N ways"): every rank receives 1/N of EVERY model's frames, whatever the order of the batch.

    A contiguous B/N slice (``shard_bounds``) is balanced only while model ids are spread evenly over the batch; a batch
    sorted by robot would put all frames of the 24-DoF DexPilot model (10-20x the per-frame cost of a per-finger vector
    model) on one or two ranks.  Here the frames of model m, in batch order, are cut into `world` contiguous balanced
    runs; the rank that receives the first run rotates from model to model by the remainders handed out so far, so the
    TOTAL number of frames per rank also differs by at most one.

    Returns a list of `world` int64 index arrays (ascending batch positions; disjoint; their union is range(B)).  Pure
    index arithmetic -- every rank computes the same partition from the same ids, nothing is exchanged."""
    mid = np.asarray(model_id).astype(np.int64).ravel()
    if world < 1:
        raise ValueError("world must be >= 1")
    if mid.size and mid.min() < 0:
        raise ValueError("model ids must be >= 0")
    M = int(n_models) if n_models is not None else (int(mid.max()) + 1 if mid.size else 0)
    if mid.size and mid.max() >= M:
        raise ValueError(f"model id {int(mid.max())} outside [0, {M})")
    parts = [[] for _ in range(world)]
    rot = 0
    order = np.argsort(mid, kind="stable")
    counts = np.bincount(mid, minlength=M) if mid.size else np.zeros(M, np.int64)
    start = 0
    for m in range(M):
        idx = order[start:start + int(counts[m])]  # positions of model m, ascending
        start += int(counts[m])
        for r in range(world):
            lo, hi = shard_bounds(idx.size, r, world)
            parts[(r + rot) % world].append(idx[lo:hi])
        rot = (rot + idx.size % world) % world
    return [np.sort(np.concatenate(p)) if p else np.zeros(0, np.int64) for p in parts]


def unshard_by_model(full: np.ndarray, shards, B: int) -> np.ndarray:
    """Inverse of the partition after an equal-slot all-gather: `full` is (world, per, ...) with rank r's rows in
    full[r, :len(shards[r])] (the rest padding); returns the (B, ...) array in batch order."""
    out = np.empty((B,) + tuple(full.shape[2:]), dtype=full.dtype)
    seen = 0
    for r, idx in enumerate(shards):
        out[idx] = full[r, : idx.size]
        seen += idx.size
    if seen != B:
        raise ValueError(f"shards cover {seen} of {B} frames")
    return out


def mixed_fleet_solve(fleet) -> Callable:
    """The per-shard `solve` of :class:`ShardedFleet` on the GPU (``MixedFleet.retarget`` itself takes and returns
    device tensors with int32 state words, ShardedFleet hands numpy arrays with uint32 ones): host arrays -> the fleet's device
    -> ``dexr_retarget_multi_dev`` -> host arrays, `state` updated in place.  Tested on the GPU against the unsharded fleet
    call (tests/test_gpu_parity.py::test_sharded_fleet_with_the_mixed_fleet_adapter)."""

    def solve(model_id, keypoints, last, state):
        torch = fleet.torch
        dev = fleet.device
        b = int(model_id.shape[0])
        if b == 0:
            return np.zeros((0, fleet.n_max), np.float32)
        t_state = None
        if state is not None:
            t_state = torch.from_numpy(np.ascontiguousarray(state).view(np.int32).copy()).to(dev)
        elif any(o.retargeting_type == "DEXPILOT" for o in fleet.optimizers):
            t_state = torch.zeros(b, dtype=torch.int32, device=dev)
        out = fleet.retarget(torch.from_numpy(np.ascontiguousarray(model_id, dtype=np.int32)).to(dev),
                             torch.from_numpy(np.ascontiguousarray(keypoints, dtype=np.float32)).to(dev),
                             torch.from_numpy(np.ascontiguousarray(last, dtype=np.float32)).to(dev), t_state)
        if state is not None:
            state[:] = t_state.cpu().numpy().view(np.uint32)
        return out.cpu().numpy()

    return solve


class ShardedFleet:
    """Mixed-fleet batch over N ranks: rank r solves ``shard_by_model(model_id, N)[r]`` (an equal share of every robot's
    frames) and ONE all-gather reassembles the (B, n_max) rows and the DexPilot state words on every rank.

    `solve(model_id, keypoints, last, state) -> (b, n_max) float32` is the per-shard fleet call on HOST arrays
    (``mixed_fleet_solve(MixedFleet(...))`` on the GPU; the world-size-2 gloo tests inject a CPU interpreter of the same
    tables); `state` (uint32) is updated in place.  (Host arrays in and out: this wrapper is the convenience path; the
    device-resident N > 1 path of the bench is bench_fleet.py -- shard_by_model on ids, MixedFleet on tensors, dexr_allgather.)  `work(model_id) -> float` (optional) is recorded per call in ``last_work`` so a test / a bench can show the
    solve work each rank received."""

    def __init__(self, solve: Callable, n_max: int, device: str = "cpu", group=None, n_models: Optional[int] = None):
        import torch.distributed as dist

        if not dist.is_initialized():
            raise RuntimeError("torch.distributed is not initialised")
        self.dist, self.group, self.device, self.solve, self.n_max, self.n_models = dist, group, device, solve, n_max, n_models
        self.rank, self.world = dist.get_rank(group), dist.get_world_size(group)
        self.last_shards = None

    def retarget(self, model_id: np.ndarray, keypoints: np.ndarray, last: np.ndarray, state: Optional[np.ndarray] = None) -> np.ndarray:
        import torch

        B = int(model_id.shape[0])
        shards = shard_by_model(model_id, self.world, self.n_models)
        self.last_shards = shards
        idx = shards[self.rank]
        st = None if state is None else np.ascontiguousarray(state[idx])
        q = self.solve(np.ascontiguousarray(model_id[idx]), np.ascontiguousarray(keypoints[idx]), np.ascontiguousarray(last[idx]), st)
        per = max(1, -(-B // self.world))
        mine = torch.zeros((per, self.n_max + 1), dtype=torch.float64, device=self.device)  # last column: the state word
        if idx.size:
            mine[: idx.size, : self.n_max] = torch.from_numpy(np.asarray(q, dtype=np.float64)).to(self.device)
            if st is not None:
                mine[: idx.size, self.n_max] = torch.from_numpy(st.astype(np.float64)).to(self.device)
        full = torch.empty((self.world * per, self.n_max + 1), dtype=torch.float64, device=self.device)
        self.dist.all_gather_into_tensor(full, mine, group=self.group)
        full = unshard_by_model(full.cpu().numpy().reshape(self.world, per, self.n_max + 1), shards, B)
        if state is not None:
            state[:] = full[:, self.n_max].astype(np.uint32)
        return full[:, : self.n_max].astype(np.float32)


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


class PipelinedAllGather:
    """All-gathers of per-step results overlapped with the following solves (RCCL runs on its own stream; xGMI is
    point-to-point, so at 8 GPUs the gather of a 4 MB shard costs about as much as the solve it follows), and
    batched: ``group`` consecutive steps share ONE collective (fewer, larger collectives -- issuing an async
    collective costs the host ~100 us in torch.distributed, more than one solve takes).

    ``depth`` buffer pairs rotate; a pair holds ``group`` steps.  Per step::

        out = pg.shard(k)      # (per, n) tensor for this rank's result of step k; when step k opens a buffer pair it
                               # first waits until the gather that last read that pair is complete
        ... enqueue the solve that writes `out` on the current stream ...
        pg.gather(k)           # after the last step of a group: enqueue all_gather(full, shards), asynchronously

    and ``pg.finish()`` gathers a partly filled last group, waits for every outstanding gather and returns the last
    full tensor, shaped (world, group, per, n).  With "nccl" (= RCCL) ``wait()`` only orders the current stream after
    the collective; with "gloo" it blocks the host.  ``on_full(first_step, tensor, n_steps)`` -- optional -- is called
    as soon as a group's gather is known complete.
    """

    def __init__(self, per: int, n: int, dtype=None, device="cpu", depth: int = 2, group=None, on_full=None,
                 steps_per_gather: int = 1):
        import torch
        import torch.distributed as dist

        if not dist.is_initialized():
            raise RuntimeError("torch.distributed is not initialised")
        if depth < 1 or steps_per_gather < 1:
            raise ValueError("depth and steps_per_gather must be >= 1")
        self.dist, self.group, self.depth, self.on_full, self.G = dist, group, depth, on_full, steps_per_gather
        self.world = dist.get_world_size(group)
        dtype = dtype or torch.float32
        self._shard = [torch.zeros((self.G, per, n), dtype=dtype, device=device) for _ in range(depth)]
        self._full = [torch.empty((self.world, self.G, per, n), dtype=dtype, device=device) for _ in range(depth)]
        self._work = [None] * depth  # (first step, Work, steps) of the gather in flight on each buffer pair
        self._open = None            # (slot, first step, steps written) of the group being filled
        self._last = None

    def _retire(self, slot: int):
        if self._work[slot] is not None:
            k0, w, cnt = self._work[slot]
            w.wait()
            self._work[slot] = None
            if self.on_full is not None:
                self.on_full(k0, self._full[slot], cnt)

    def _issue(self):
        slot, k0, cnt = self._open
        per, n = self._shard[slot].shape[1:]
        w = self.dist.all_gather_into_tensor(self._full[slot].view(self.world * self.G * per, n),
                                             self._shard[slot].view(self.G * per, n), group=self.group, async_op=True)
        self._work[slot] = (k0, w, cnt)
        self._last = slot
        self._open = None

    def shard(self, k: int):
        slot, g = (k // self.G) % self.depth, k % self.G
        if self._open is None:
            if g != 0:
                raise RuntimeError(f"step {k} does not start a group of {self.G}")
            self._retire(slot)
            self._open = (slot, k, 0)
        elif self._open[0] != slot or self._open[1] + self._open[2] != k:
            raise RuntimeError(f"step {k} out of order")
        return self._shard[slot][g]

    def gather(self, k: int):
        if self._open is None or self._open[1] + self._open[2] != k:
            raise RuntimeError(f"step {k}: call shard(k) first (each step once, in order)")
        slot, k0, cnt = self._open
        self._open = (slot, k0, cnt + 1)
        if cnt + 1 == self.G:
            self._issue()

    def finish(self):
        if self._open is not None and self._open[2] > 0:
            self._issue()  # partly filled last group: the unused rows travel too
        self._open = None
        order = sorted((kw[0], s) for s, kw in enumerate(self._work) if kw is not None)
        for _, slot in order:
            self._retire(slot)
        return None if self._last is None else self._full[self._last]


# ---- native collective (libdexr.so: dexr_comm_* / dexr_allgather) -----------------------------------------------------

def rendezvous_store(rank: int, world: int, timeout_s: float = 300.0):
    """The key-value store the ranks of this job share: the TCP store of ``torch.distributed.run`` when it launched us
    (TORCHELASTIC_USE_AGENT_STORE: every worker is a client), else one hosted by rank 0 on MASTER_ADDR:MASTER_PORT."""
    import datetime
    import os

    from torch.distributed import TCPStore

    addr = os.environ.get("MASTER_ADDR", "127.0.0.1")
    port = int(os.environ.get("MASTER_PORT", "29511"))
    agent = os.environ.get("TORCHELASTIC_USE_AGENT_STORE", "") == "True"
    return TCPStore(addr, port, world, is_master=(rank == 0 and not agent), timeout=datetime.timedelta(seconds=timeout_s),
                    wait_for_workers=False)


def exchange_bytes(store, rank: int, key: str, make) -> bytes:
    """Rank 0 publishes ``make()`` under `key`; every rank returns it (``store.get`` blocks until it is there)."""
    if rank == 0:
        blob = make()
        store.set(key, blob)
        return blob
    return bytes(store.get(key))


_comm_generation = [0]


def comm_key(base: str = "dexr/unique_id") -> str:
    """Store key of the NEXT communicator this process creates: `base`/r<restart>/c<count>.  Every rank of a job creates
    its communicators in the same order, so the count agrees across ranks without another exchange; the elastic restart
    count (TORCHELASTIC_RESTART_COUNT) separates attempts.  A fixed key would hand the id of an earlier communicator --
    or of the attempt before a restart -- to the non-zero ranks the moment they ask (the agent's store outlives both),
    and ncclCommInitRank then hangs on mismatched ids."""
    import os

    k = f"{base}/r{os.environ.get('TORCHELASTIC_RESTART_COUNT', '0')}/c{_comm_generation[0]}"
    _comm_generation[0] += 1
    return k


def native_comm(rank: int, world: int, store=None, key: str = None):
    """An RCCL communicator over all ranks, bound to the CURRENT HIP device (call torch.cuda.set_device first).  `key`
    (optional) names the store entry the unique id travels under; by default every communicator gets its own
    (:func:`comm_key`)."""
    from . import _lib

    if key is None:
        key = comm_key()
    if world == 1:
        return _lib.Comm(_lib.comm_unique_id(), 0, 1)
    if store is None:
        store = rendezvous_store(rank, world)
    uid = exchange_bytes(store, rank, key, _lib.comm_unique_id)
    return _lib.Comm(uid, rank, world)


class NativeGather:
    """All-gather of this rank's (per, n) float32 result rows through ``dexr_allgather``, one collective per
    ``steps_per_gather`` steps.

    ``depth`` (shard, full) buffer pairs rotate; a pair holds ``steps_per_gather`` = G consecutive steps.  Per step k::

        out = ng.shard(k)    # this rank's rows of step k; when step k opens a buffer pair the solve stream first waits
                             # for the gather that last read that pair (an event, no host wait)
        ... enqueue the solve that writes `out` on the current stream ...
        ng.gather(k)         # after the G-th step of a group: ONE collective of G shards.
                             # overlap=True: it runs on a second stream ordered after the solve by an event, so the
                             # following solves overlap it;  overlap=False: on the solve stream itself (strictly serial,
                             # what a single captured hipGraph of [solve, all-gather] does)

    G > 1 is for steps shorter than their gather (Allegro vector at 8 GPUs: 47 us of solve against >= 55 us of xGMI
    ingest): the same bytes travel, in 1/G as many collectives, so the per-collective launch /
    synchronisation cost of RCCL (tens of us) is paid once per G steps and the copy engines see G x larger messages;
    results reach the other ranks up to G - 1 steps later.  ``finish()`` gathers a partly filled last group, makes the
    current stream wait for every gather in flight and returns the last full tensor, shaped (world, G, per, n)
    ((world, per, n) when G = 1).  Nothing here blocks the host."""

    def __init__(self, comm, per: int, n: int, device, depth: int = 4, overlap: bool = True, steps_per_gather: int = 1):
        import torch

        if depth < 1 or steps_per_gather < 1:
            raise ValueError("depth and steps_per_gather must be >= 1")
        self.torch, self.comm, self.depth, self.G = torch, comm, depth, steps_per_gather
        # host tensors (the world-size-2 CPU tests, with a gloo-backed stand-in for the communicator): no streams, every
        # collective completes before gather() returns
        self._gpu = torch.device(device).type == "cuda"
        self.overlap = overlap and self._gpu
        self.bytes_per_rank = per * n * 4  # of ONE step
        G = self.G
        self._shard = [torch.zeros((G, per, n) if G > 1 else (per, n), dtype=torch.float32, device=device) for _ in range(depth)]
        self._full = [torch.empty((comm.world, G, per, n) if G > 1 else (comm.world, per, n), dtype=torch.float32, device=device)
                      for _ in range(depth)]
        self._done = [None] * depth
        self._comm_stream = torch.cuda.Stream(device=device) if self.overlap else None
        self._last = None
        self._filled = 0  # steps written into the open group
        self._open_slot = 0
        self.collectives = 0

    def _slot(self, k: int) -> int:
        return (k // self.G) % self.depth

    def shard(self, k: int):
        slot, g = self._slot(k), k % self.G
        if g == 0 and self._done[slot] is not None:
            self.torch.cuda.current_stream().wait_event(self._done[slot])
        return self._shard[slot][g] if self.G > 1 else self._shard[slot]

    def _issue(self, slot: int):
        torch = self.torch
        # a partly filled last group sends the whole block: the receive layout (world, G, per, n) needs equal counts
        nbytes = self.bytes_per_rank * self.G
        self.collectives += 1
        self._last = slot
        self._filled = 0
        if not self._gpu:
            self.comm.allgather(self._shard[slot].data_ptr(), self._full[slot].data_ptr(), nbytes, 0)
            return
        cur = torch.cuda.current_stream()
        if self.overlap:
            ready = torch.cuda.Event()
            ready.record(cur)
            self._comm_stream.wait_event(ready)
            st = self._comm_stream
        else:
            st = cur
        self.comm.allgather(self._shard[slot].data_ptr(), self._full[slot].data_ptr(), nbytes, st.cuda_stream)
        done = torch.cuda.Event()
        done.record(st)
        self._done[slot] = done

    def gather(self, k: int):
        self._filled += 1
        self._open_slot = self._slot(k)
        if k % self.G == self.G - 1:
            self._issue(self._open_slot)

    def finish(self):
        if self._filled:
            self._issue(self._open_slot)
        if self._gpu:
            cur = self.torch.cuda.current_stream()
            for ev in self._done:
                if ev is not None:
                    cur.wait_event(ev)
        return None if self._last is None else self._full[self._last]


def steps_per_gather_for(step_ms: float, shard_bytes: int, world: int, ingest_GBps: float = 537.0, target: float = 0.6,
                         collective_overhead_us: float = 30.0, max_steps: int = 16) -> int:
    """How many steps should share one all-gather so that the collectives keep up with the solves.

    A gather of G steps moves (world - 1) * G * shard_bytes into every GPU: at `target` of the xGMI ingest that takes
    t_wire(G) = G * (world - 1) * shard_bytes / (target * ingest) and costs one `collective_overhead_us` of launch /
    rendezvous.  The gather stream keeps up when  t_wire(G) + overhead <= G * step_ms, i.e.
    G >= overhead / (step - wire_per_step); when the wire time of ONE step already exceeds the step (the job is
    gather-bound whatever G is) the answer is `max_steps`: fewer, larger collectives are then simply the cheaper way to
    move the same bytes."""
    if world <= 1 or shard_bytes <= 0:
        return 1
    wire_us = (world - 1) * shard_bytes / (target * ingest_GBps * 1e9) * 1e6
    slack_us = step_ms * 1e3 - wire_us
    if slack_us <= 0:
        return max_steps
    import math

    return int(min(max_steps, max(1, math.ceil(collective_overhead_us / slack_us))))
