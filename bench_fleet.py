"""bench.py --workload mixed_fleet: BASELINE.json configs[4] (Allegro + Shadow + LEAP + Ability in one batch)."""
import json
import os
import sys
import time

import numpy as np

REPO = os.path.dirname(os.path.abspath(__file__))
if REPO not in sys.path:
    sys.path.insert(0, REPO)

from bench import FLEET, HBM_PEAK_GBPS, WORKLOADS  # noqa: E402


def run(args):
    """BASELINE.json configs[4] on ONE GPU's share: B frames whose robot changes from frame to frame.  A step buckets
    the frames by model (wavefronts must be model-uniform: the kinematic tables are scalar operands), solves the four
    buckets concurrently on four HIP streams and scatters the answers back into the caller's order, all on the device
    (dex_retargeting_amd/fleet.py).  Single-rank only; the 8-GPU run of this config shards the batch like the others."""
    import torch

    import bench_data
    from dex_retargeting_amd.constants import DEFAULT_URDF_DIR
    from dex_retargeting_amd.fleet import MixedFleet
    from dex_retargeting_amd.retargeting_config import RetargetingConfig

    torch.cuda.set_device(0)
    dev = torch.device("cuda", 0)
    RetargetingConfig.set_default_urdf_dir(str(DEFAULT_URDF_DIR))
    seqs = [RetargetingConfig.load_from_file(os.path.join(bench_data.CONFIG_DIR, r)).build() for r in FLEET]
    fleet = MixedFleet([q.optimizer for q in seqs])
    B = args.batch
    rng = np.random.default_rng(bench_data.SEED)
    mid = rng.integers(0, len(FLEET), B)
    kp = bench_data.human_keypoints(B + 1, seed=bench_data.SEED)
    t_mid = torch.from_numpy(mid).to(dev)
    t_prev, t_now = torch.from_numpy(np.ascontiguousarray(kp[:-1])).to(dev), torch.from_numpy(np.ascontiguousarray(kp[1:])).to(dev)
    start = np.zeros((B, fleet.n_max), np.float32)
    for m, sq in enumerate(seqs):
        start[mid == m, : sq.optimizer.opt_dof] = sq.joint_limits.mean(1).astype(np.float32)
    t_state = torch.zeros(B, dtype=torch.int32, device=dev)
    t_last = fleet.retarget(t_mid, t_prev, torch.from_numpy(start).to(dev), t_state)  # untimed warm start
    t_state0 = t_state.clone()
    stream = torch.cuda.current_stream()

    def step():
        t_state.copy_(t_state0)
        return fleet.retarget(t_mid, t_now, t_last, t_state)

    for _ in range(args.warmup):
        step()
    torch.cuda.synchronize()
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(args.steps)]
    t0 = time.perf_counter()
    for a, b in ev:
        a.record(stream)
        out = step()
        b.record(stream)
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    step_ms = float(np.mean([a.elapsed_time(b) for a, b in ev]))
    q = out.cpu().numpy()
    last = t_last.cpu().numpy()
    st_in = t_state0.cpu().numpy().astype(np.uint32)
    bpf = 21 * 12 + 2 * 4 * fleet.n_max + 4 + 8  # keypoints + padded last/qpos rows + model id + DexPilot state in/out
    achieved = B * bpf / (step_ms * 1e-3) / 1e9
    # ---- checker (oracle) and CPU baseline: only from here on ----------------------------------------------------
    from oracle import cases, solvers

    probs = [cases.problem_from_config(r) for r in FLEET]
    parity, cpu_t, cpu_n = {}, 0.0, 0
    for m, (rel, pr) in enumerate(zip(FLEET, probs)):
        idx = np.nonzero(mid == m)[0][:128]
        ref = cases.ref_from_keypoints(pr, kp[1:][idx]).astype(np.float32)
        kw = {}
        if pr.kind == "dexpilot":
            proj = ((st_in[idx, None] >> np.arange(pr.n_pair, dtype=np.uint32)) & 1).astype(bool)
            w, rv, _ = pr.dexpilot_preamble(ref, proj)
            kw = dict(weights=w, dexpilot_ref=rv)
        la = last[idx][:, : pr.n_opt]
        want = solvers.solve_lm_batched(pr, ref, None, la, newton=True, max_iter=100, **kw)
        dq = np.abs(q[idx][:, : pr.n_opt].astype(np.float64) - want).max(1)
        parity[rel] = {"subset": len(idx), "max_abs_dq_rad": float(dq.max()), "frac_within_1e-4": float((dq < 1e-4).mean())}
        if not args.no_cpu_baseline:
            t1 = time.perf_counter()
            solvers.solve_ref_as_configured(pr, ref[:60], None, la[:60], **{k: v[:60] for k, v in kw.items()})
            cpu_t += time.perf_counter() - t1
            cpu_n += 60
    out_json = {
        "metric": "retargeted frames/sec, mixed-fleet batch (BASELINE.json configs[4]), one MI355X",
        "value": B * args.steps / elapsed, "unit": "frames/s", "n_gpus": 1, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": elapsed / args.steps * 1e3, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": "f32", "data": "synthetic",
        "config": {"workload": f"{WORKLOADS['mixed_fleet'][1]}; {B} frames/GPU, model id uniform at random per frame, "
                               f"human-keypoint refs, warm start = previous frame's solution", "models": FLEET,
                   "batch_per_gpu": B, "collective": "none"},
        "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBPS, "unit": "GB/s",
                     "frac": achieved / HBM_PEAK_GBPS, "traffic": None, "kernel_ms": step_ms,
                     "algorithmic_bytes_per_frame": bpf,
                     "kernel": "bucket by model (torch index ops) + 4 solve kernels on 4 streams + scatter"},
        "parity": parity,
    }
    if cpu_n:
        out_json["cpu_baseline"] = {"value": cpu_n / cpu_t, "unit": "frames/s", "cores": 1, "kind": "port",
                                    "sample": "60 frames of each of the four models, reference-as-configured port"}
    print(json.dumps(out_json))


