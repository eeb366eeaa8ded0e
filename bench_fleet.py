"""bench.py --workload mixed_fleet: BASELINE.json configs[4] -- a batch whose robot changes from frame to frame (Allegro
vector + Shadow DexPilot + LEAP vector + Ability vector, 4 URDFs), 1 048 576 frames over 8 GPUs = 131 072 per GPU.

One step = ONE call of ``dexr_retarget_multi_dev`` per rank (device-side bucketing by model id, one solve launch per
model over its index list, rows read and written in place; no host synchronisation) and, for N > 1, one RCCL all-gather
of the (B, n_max) result.  Launched like bench.py (``torch.distributed.run`` for N > 1).
"""
import json
import os
import sys
import time

import numpy as np

REPO = os.path.dirname(os.path.abspath(__file__))
if REPO not in sys.path:
    sys.path.insert(0, REPO)

from bench import FLEET, HBM_PEAK_GBPS, N_BATCHES, WORKLOADS  # noqa: E402


def run(args):
    import torch

    import bench_data
    from dex_retargeting_amd.constants import DEFAULT_URDF_DIR
    from dex_retargeting_amd.fleet import MixedFleet
    from dex_retargeting_amd.retargeting_config import RetargetingConfig

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus and world > 1:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    dist = None
    if world > 1 or (os.environ.get("RANK") is not None and os.environ.get("DEXR_BENCH_DIST", "1") != "0"):
        import torch.distributed as dist

        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29511")
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)

    RetargetingConfig.set_default_urdf_dir(str(DEFAULT_URDF_DIR))
    seqs = [RetargetingConfig.load_from_file(os.path.join(bench_data.CONFIG_DIR, r)).build() for r in FLEET]
    fleet = MixedFleet([q.optimizer for q in seqs], device=str(dev))
    B = args.batch
    seed = bench_data.SEED + 1000 * rank
    batches = []
    for j in range(N_BATCHES):
        rng = np.random.default_rng(seed + 7 * j)
        mid = rng.integers(0, len(FLEET), B).astype(np.int32)
        kp = bench_data.human_keypoints(B + 1, seed=seed + 17 * j, offset=155 * j)
        t_mid = torch.from_numpy(mid).to(dev)
        start = np.zeros((B, fleet.n_max), np.float32)
        for m, sq in enumerate(seqs):
            start[mid == m, : sq.optimizer.opt_dof] = sq.joint_limits.mean(1).astype(np.float32)
        t_state0 = torch.zeros(B, dtype=torch.int32, device=dev)
        # untimed: the previous frame's solution = the warm start a running sequence would carry
        t_last = fleet.retarget(t_mid, torch.from_numpy(np.ascontiguousarray(kp[:-1])).to(dev),
                                torch.from_numpy(start).to(dev), t_state0)
        torch.cuda.synchronize()
        batches.append(dict(mid=mid, kp=np.ascontiguousarray(kp[1:]), t_mid=t_mid, t_last=t_last.clone(),
                            t_kp=torch.from_numpy(np.ascontiguousarray(kp[1:])).to(dev), t_state0=t_state0.clone()))
    t_state = torch.zeros(B, dtype=torch.int32, device=dev)
    t_out = torch.zeros((B, fleet.n_max), dtype=torch.float32, device=dev)
    stream = torch.cuda.current_stream()
    pipe = None
    if dist is not None:
        from dex_retargeting_amd.distributed import PipelinedAllGather

        pipe = PipelinedAllGather(B, fleet.n_max, torch.float32, dev, depth=min(16, args.steps + args.warmup + 2),
                                  steps_per_gather=1)
    n = [0]

    def step():
        k = n[0]
        b = batches[k % N_BATCHES]
        t_state.copy_(b["t_state0"])
        out = t_out if pipe is None else pipe.shard(k)
        fleet.retarget(b["t_mid"], b["t_kp"], b["t_last"], t_state, out=out)
        if pipe is not None:
            pipe.gather(k)
        n[0] += 1
        return b

    for _ in range(args.warmup):
        step()
    if pipe is not None:
        pipe.finish()
    ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    if dist is not None:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    ev0.record(stream)
    for _ in range(args.steps):
        last_b = step()
    ev1.record(stream)
    if pipe is not None:
        pipe.finish()
    torch.cuda.synchronize()
    if dist is not None:
        dist.barrier()
    elapsed = time.perf_counter() - t0
    step_ms = float(ev0.elapsed_time(ev1)) / args.steps
    if dist is not None:
        tt = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        elapsed = float(tt.item())
    if rank != 0:
        dist.destroy_process_group()
        return
    # answers of the last step's batch (re-issued into a plain buffer) for the checker
    t_state.copy_(last_b["t_state0"])
    q = fleet.retarget(last_b["t_mid"], last_b["t_kp"], last_b["t_last"], t_state, out=torch.zeros_like(t_out)).cpu().numpy()
    last = last_b["t_last"].cpu().numpy()
    st_in = last_b["t_state0"].cpu().numpy().astype(np.uint32)
    mid, kp = last_b["mid"], last_b["kp"]
    bpf = 21 * 12 + 2 * 4 * fleet.n_max + 4 + 8  # keypoints + padded last/qpos rows + model id + DexPilot state in/out
    achieved = B * bpf / (step_ms * 1e-3) / 1e9
    out_json = {
        "metric": "retargeted frames/sec, mixed-fleet batch (BASELINE.json configs[4])",
        "value": world * B * args.steps / elapsed, "unit": "frames/s", "n_gpus": world, "steps": args.steps,
        "warmup": args.warmup, "ms_per_step": elapsed / args.steps * 1e3, "higher_is_better": True, "scaling": "weak",
        "vs_baseline": None, "dtype": "f32", "data": "synthetic",
        "config": {"workload": f"{WORKLOADS['mixed_fleet'][1]}; {B} frames/GPU, model id uniform at random per frame, "
                               f"human-keypoint refs, warm start = previous frame's solution; {N_BATCHES} staged batches rotated",
                   "models": FLEET, "batch_per_gpu": B,
                   "collective": "none" if dist is None else "rccl all_gather of the (B, n_max) qpos rows, one per step",
                   "rccl_world_size": world if dist is not None else None},
        "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBPS, "unit": "GB/s",
                     "frac": achieved / HBM_PEAK_GBPS, "traffic": None, "kernel_ms": step_ms,
                     "algorithmic_bytes_per_frame": bpf,
                     "kernel": "dexr_retarget_multi_dev: device-side bucketing (3 small kernels) + one solve launch per model "
                               "over its index list, in-place rows"},
    }
    # ---- checker (oracle) and CPU baseline: only from here on ----------------------------------------------------
    from oracle import cases, solvers

    probs = [cases.problem_from_config(r) for r in FLEET]
    parity, cpu_t, cpu_n = {}, 0.0, 0
    for m, (rel, pr) in enumerate(zip(FLEET, probs)):
        idx = np.nonzero(mid == m)[0][:128]
        ref = cases.ref_from_keypoints(pr, kp[idx]).astype(np.float32)
        kw = {}
        if pr.kind == "dexpilot":
            proj = ((st_in[idx, None] >> np.arange(pr.n_pair, dtype=np.uint32)) & 1).astype(bool)
            w, rv, _ = pr.dexpilot_preamble(ref, proj)
            kw = dict(weights=w, dexpilot_ref=rv)
        la = last[idx][:, : pr.n_opt]
        want = solvers.solve_lm_batched(pr, ref, None, la, newton=True, max_iter=100, **kw)
        dq = np.abs(q[idx][:, : pr.n_opt].astype(np.float64) - want).max(1)
        parity[rel] = {"subset": len(idx), "max_abs_dq_rad": float(dq.max()), "frac_within_1e-4": float((dq < 1e-4).mean())}
        if not args.no_cpu_baseline and world == 1:
            t1 = time.perf_counter()
            solvers.solve_ref_as_configured(pr, ref[:60], None, la[:60], **{k: v[:60] for k, v in kw.items()})
            cpu_t += time.perf_counter() - t1
            cpu_n += 60
    out_json["parity"] = parity
    if cpu_n:
        out_json["cpu_baseline"] = {"value": cpu_n / cpu_t, "unit": "frames/s", "cores": 1, "kind": "port",
                                    "sample": "60 frames of each of the four models, reference-as-configured port"}
    print(json.dumps(out_json))
    if dist is not None:
        dist.destroy_process_group()
