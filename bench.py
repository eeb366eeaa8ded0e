#!/usr/bin/env python3
"""bench.py -- retargeted frames/s of the batched HIP solver on N MI355X GPUs (one process per GPU).

    python bench.py --gpus 1 --steps 20 --warmup 3
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W

A "step" is one pass of the hot path over one batch: B = 65 536 independent frames per GPU (weak scaling) of the
workload BASELINE.json quotes the metric on -- Allegro right hand, VectorOptimizer -- solved to the tight
tolerance, inputs resident in HBM when the timed region starts, result qpos resident (and, for N > 1,
all-gathered with one RCCL all-gather) when it ends.

Workload (synthetic, seeded): keypoints = frame (b mod 621) of the human fixture
+ N(0, 2 mm); ref_value = kp[task] - kp[origin] (profile_online_retargeting.py:24-30); last_qpos = the solver's own
answer for the neighbouring frame (b-1), i.e. the warm start a running sequence would have (seq_retarget.py:124).

Rank 0 prints ONE JSON line with the driver's contract plus `roofline` and `cpu_baseline`.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

REPO = os.path.dirname(os.path.abspath(__file__))
if REPO not in sys.path:
    sys.path.insert(0, REPO)

HBM_PEAK_GBPS = 8000.0  # MI355X_MICROARCH.md: 8.0 TB/s spec
FP32_VALU_PEAK_TFLOPS = 157.3

FLEET = ["teleop/allegro_hand_right.yml", "teleop/shadow_hand_right_dexpilot.yml", "teleop/leap_hand_right.yml",
         "teleop/ability_hand_right.yml"]  # BASELINE.json configs[4]: 4 URDFs in one batch

WORKLOADS = {
    "allegro_vector": ("teleop/allegro_hand_right.yml", "Allegro right hand, VectorOptimizer"),
    "shadow_dexpilot": ("teleop/shadow_hand_right_dexpilot.yml", "Shadow right hand (24 DoF), DexPilotOptimizer"),
    "leap_position": ("offline/leap_hand_right.yml", "LEAP right hand + 6 free joints, PositionOptimizer"),
    "mixed_fleet": (None, "Mixed fleet: Allegro vector + Shadow DexPilot + LEAP vector + Ability vector, frames interleaved"),
}


def run_mixed_fleet(args):
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


def algorithmic_bytes_per_frame(n_opt: int, dexpilot: bool, n_kp: int = 21) -> int:
    """Compulsory HBM traffic of one frame through dexr_retarget_kp_dev:
    21 raw keypoints in (252 B) + last_qpos in + qpos out (+ 4 B DexPilot state in and out)."""
    return n_kp * 12 + n_opt * 4 + n_opt * 4 + (8 if dexpilot else 0)


def algorithmic_flops_per_pass(compiled) -> float:
    """FP operations of ONE solver pass (FK + value/gradient/Hessian + factorisation + step) over all components of a
    frame, counted from the compiled tables: per component with n joints and T terms
    FK 100 n  +  T (20 + 18 n + 18 n(n+1)/2)  +  n^3/3 + 2 n^2  +  10 n."""
    total = 0.0
    for c in compiled.comps:
        n, t = int(c["n_joint"]), int(c["n_term"])
        total += 100 * n + t * (20 + 18 * n + 18 * n * (n + 1) / 2) + n ** 3 / 3 + 2 * n * n + 10 * n
    return total


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--batch", type=int, default=65536, help="frames per GPU")
    ap.add_argument("--workload", default="allegro_vector", choices=sorted(WORKLOADS))
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-sample", type=int, default=1200, help="frames of the workload timed on the host CPU")
    args = ap.parse_args()
    if args.workload == "mixed_fleet":
        if int(os.environ.get("WORLD_SIZE", "1")) > 1:
            raise SystemExit("--workload mixed_fleet measures one GPU's share; run it with --gpus 1")
        if args.batch == 65536:
            args.batch = 131072  # 1 048 576 frames / 8 GPUs
        return run_mixed_fleet(args)

    import torch

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus and world > 1:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    dist = None
    # launched by torch.distributed.run (RANK set): take the N > 1 path even with one rank, so that the RCCL side of
    # this script can be exercised on a 1-GPU box too
    if world > 1 or (os.environ.get("RANK") is not None and os.environ.get("DEXR_BENCH_DIST", "1") != "0"):
        import torch.distributed as dist

        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29511")
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)

    import bench_data  # seeded synthetic inputs (no oracle code before the checker sections at the end)
    from dex_retargeting_amd.constants import DEFAULT_URDF_DIR
    from dex_retargeting_amd.retargeting_config import RetargetingConfig

    RetargetingConfig.set_default_urdf_dir(str(DEFAULT_URDF_DIR))
    rel, wl_name = WORKLOADS[args.workload]
    seq = RetargetingConfig.load_from_file(os.path.join(bench_data.CONFIG_DIR, rel)).build()
    opt = seq.optimizer
    model = opt.device_model()
    B = args.batch
    n_opt, n_ref = opt.opt_dof, int(opt.compiled_model().header["n_ref"])
    dexpilot = opt.retargeting_type == "DEXPILOT"

    # ---- synthetic inputs, resident in HBM ------------------------------------------------------------------
    seed = bench_data.SEED + 1000 * rank
    kp = bench_data.human_keypoints(B + 1, seed=seed)  # (B+1, 21, 3) float32
    kp_prev, kp_now = np.ascontiguousarray(kp[:-1]), np.ascontiguousarray(kp[1:])
    mid = np.repeat(seq.joint_limits.mean(1)[None], B, 0).astype(np.float32)
    st0 = np.zeros(B, np.uint32) if dexpilot else None
    # untimed: the previous frame's solution = the warm start a running sequence would carry
    last = model.retarget(kp_prev, None, mid, state=st0, keypoints=True)
    t_ref = torch.from_numpy(kp_now).to(dev)  # raw keypoints: ref_value is formed inside the kernel
    t_last = torch.from_numpy(last).to(dev)
    t_state0 = torch.from_numpy(st0.astype(np.int32)).to(dev) if dexpilot else None
    t_state = t_state0.clone() if dexpilot else None
    t_q = torch.empty((B, n_opt), dtype=torch.float32, device=dev)
    t_iters = torch.zeros(B, dtype=torch.int32, device=dev)
    t_status = torch.zeros(B, dtype=torch.int32, device=dev)
    stream = torch.cuda.current_stream()
    # N > 1: one RCCL all-gather of this rank's (B, n_opt) result per step, overlapped with the following steps' solves
    # (rotating buffer pairs; every gather has completed when the timed region ends)
    pipe = None
    if dist is not None:
        from dex_retargeting_amd.distributed import PipelinedAllGather

        # four steps share one collective (issuing an async collective costs the host ~100 us in torch.distributed,
        # more than a solve takes; 4 x 4 MB per rank is still a small message for xGMI), and there are enough buffer
        # pairs (HBM is 288 GB) that the compute stream hardly ever has to wait for an earlier gather before reusing
        # one: the gathers simply trail the solves on RCCL's stream
        G = int(os.environ.get("DEXR_BENCH_GATHER_EVERY", "4"))
        pipe = PipelinedAllGather(B, n_opt, torch.float32, dev, depth=min(16, (args.steps + args.warmup) // G + 3),
                                  steps_per_gather=G)
    n_step, out_last = [0], [None]

    def step(diagnostics=False):
        if dexpilot:
            t_state.copy_(t_state0)
        out = t_q if pipe is None else pipe.shard(n_step[0])
        out_last[0] = out
        model.retarget_dev(B, t_ref.data_ptr(), 0, t_last.data_ptr(), t_state.data_ptr() if dexpilot else 0,
                           out.data_ptr(), status_ptr=t_status.data_ptr() if diagnostics else 0,
                           iters_ptr=t_iters.data_ptr() if diagnostics else 0, stream=stream.cuda_stream,
                           keypoints=True)
        if pipe is not None:
            pipe.gather(n_step[0])
        n_step[0] += 1

    for _ in range(args.warmup):
        step()
    step(diagnostics=True)  # untimed: iteration counts / status of this workload
    if pipe is not None:
        t_q.copy_(out_last[0])
        pipe.finish()
        n_step[0] = 0  # the timed steps start a fresh group
    torch.cuda.synchronize()
    iters_mean = float(t_iters.float().mean())
    iters_max = int(t_iters.max())
    n_conv = int((t_status == 0).sum())

    # HIP events bracket the K launches on the stream they are issued on; average launch duration = span / K.  (An event
    # pair around every single launch puts two extra packets between consecutive kernels and costs ~8 us per step.)
    ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    if dist is not None:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    ev0.record(stream)
    for k in range(args.steps):
        step()
    ev1.record(stream)
    if pipe is not None:
        pipe.finish()
    torch.cuda.synchronize()
    if dist is not None:
        dist.barrier()
    elapsed = time.perf_counter() - t0
    kernel_ms = float(ev0.elapsed_time(ev1)) / args.steps
    if dist is not None:
        tt = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        elapsed = float(tt.item())

    if rank != 0:
        dist.destroy_process_group()
        return

    frames = world * B * args.steps
    value = frames / elapsed
    bpf = algorithmic_bytes_per_frame(n_opt, dexpilot)
    achieved = B * bpf / (kernel_ms * 1e-3) / 1e9
    # HBM bytes per launch as counted by rocprofv3 PMC passes of this same command (tools/profile_round.sh writes the
    # summary, committed under profiles/): 2 x FETCH_SIZE + WRITE_SIZE, gfx950 correction per MI355X_MICROARCH.md
    traffic, valu_frac = None, None
    pmc_path = os.path.join(REPO, "profiles", f"pmc_{args.workload}.json")
    if os.path.exists(pmc_path):
        pmc = json.load(open(pmc_path))
        if pmc.get("batch") == B and world == 1:
            traffic = pmc.get("hbm_bytes_per_launch")
            if "SQ_INSTS_VALU" in pmc:  # wave64 VALU instruction = 4 issue cycles on a 16-lane SIMD; 1024 SIMDs, 2.4 GHz
                valu_frac = pmc["SQ_INSTS_VALU"] * 4.0 / (1024 * kernel_ms * 1e-3 * 2.4e9)
    flops_frame = algorithmic_flops_per_pass(opt.compiled_model()) * (iters_mean + 1.0)  # +1: the start point's model
    valu_tflops = B * flops_frame / (kernel_ms * 1e-3) / 1e12
    out = {
        "metric": json.load(open(os.path.join(REPO, "BASELINE.json")))["metric"],
        "value": value, "unit": "frames/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": elapsed / args.steps * 1e3, "higher_is_better": True, "scaling": "weak",
        "vs_baseline": None, "dtype": "f32", "data": "synthetic",
        "config": {"workload": f"{wl_name}, {B} frames/GPU, human-keypoint refs (fixture frame b mod 621 + 2 mm noise), "
                               f"warm start = previous frame's solution", "config_file": rel, "batch_per_gpu": B,
                   "n_opt": n_opt, "n_ref": n_ref, "collective": "rccl all_gather of qpos (one per 4 steps, 4 x B rows per rank), overlapped with the following solves" if dist is not None else "none"},
        "solver": {"iters_mean": iters_mean, "iters_max": iters_max, "converged_frac": n_conv / B,
                   "tol_rad": 2e-6, "newton": 1},
        "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBPS, "unit": "GB/s",
                     "frac": achieved / HBM_PEAK_GBPS, "traffic": traffic,
                     "traffic_note": "bytes per launch from the rocprofv3 --pmc passes in profiles/ (same command, "
                                     "same batch); null when no matching profile is committed",
                     "valu_issue_frac": valu_frac,
                     "valu": {"achieved": valu_tflops, "peak": FP32_VALU_PEAK_TFLOPS, "unit": "TFLOP/s",
                              "frac": valu_tflops / FP32_VALU_PEAK_TFLOPS, "algorithmic_flops_per_frame": flops_frame,
                              "note": "algorithmic flops per solver pass (bench.py:algorithmic_flops_per_pass) x mean "
                                      "passes per frame of this run"},
                     "kernel": "dexr_kernel<4,float,SOLVE,CHAIN> (Allegro vector) / dexr_quad_kernel<24> (Shadow DexPilot, LEAP position)",
                     "kernel_ms": kernel_ms,
                     "algorithmic_bytes_per_frame": bpf,
                     "note": "path is FP32 VALU/latency bound (n_dof <= 24 per lane, no dense contraction); the HBM "
                             "fraction is reported as north_star asks, see DESIGN.md section 4"},
    }

    # ---- parity on a subset (oracle = checker only; nothing above this line touches oracle/) ---------------------
    from oracle import cases, solvers

    prob = cases.problem_from_config(rel)
    assert (prob.n_opt, prob.n_ref) == (n_opt, n_ref)
    ref_now = cases.ref_from_keypoints(prob, kp).astype(np.float32)[1:]

    n_par = 4096 if args.workload == "allegro_vector" else 512  # 4 096-item subset on the headline
    kw = {}
    if dexpilot:  # same incoming projection state as the timed launches: the bits the previous frame left behind
        proj0 = ((st0[:n_par, None] >> np.arange(prob.n_pair, dtype=np.uint32)) & 1).astype(bool)
        w, rv, _ = prob.dexpilot_preamble(ref_now[:n_par], proj0)
        kw = dict(weights=w, dexpilot_ref=rv)
    want = solvers.solve_lm_batched(prob, ref_now[:n_par], None, last[:n_par], newton=True, max_iter=100, **kw)
    got = t_q[:n_par].cpu().numpy().astype(np.float64)
    dq = np.abs(got - want).max(1)
    # several minima exist on human targets (DexPilot especially): where the two answers are different minima, compare
    # the objective they reach
    last64 = last[:n_par].astype(np.float64)
    F_got = prob.total(got, ref_now[:n_par], None, last64, **kw)
    F_want = prob.total(want, ref_now[:n_par], None, last64, **kw)
    other = dq >= 1e-4
    out["parity"] = {"subset": n_par, "max_abs_dq_rad": float(dq.max()), "p99_abs_dq_rad": float(np.percentile(dq, 99)),
                     "frac_within_1e-4": float((dq < 1e-4).mean()),
                     "other_minimum": {"frames": int(other.sum()),
                                       "gpu_objective_lower_or_equal": int((F_got[other] <= F_want[other] + 1e-9).sum()),
                                       "median_F_gpu_minus_F_oracle": float(np.median(F_got[other] - F_want[other])) if other.any() else 0.0},
                     "max_abs_dq_rad_same_minimum": float(dq[~other].max()) if (~other).any() else None,
                     "oracle": "float64 projected LM/Newton on F (oracle/solvers.py)"}

    # ---- CPU baseline: the reference path as configured (scipy SLSQP stand-in for nlopt), host cores ----------
    if world == 1 and not args.no_cpu_baseline:
        budget_s, done, t_cpu = 15.0, 0, 0.0
        while done < min(args.cpu_sample, B) and t_cpu < budget_s:  # bounded sample: ~15 s of host work
            lo, hi = done, min(done + 50, B)
            kw_c = {}
            if dexpilot:
                proj_c = ((st0[lo:hi, None] >> np.arange(prob.n_pair, dtype=np.uint32)) & 1).astype(bool)
                w, rv, _ = prob.dexpilot_preamble(ref_now[lo:hi], proj_c)
                kw_c = dict(weights=w, dexpilot_ref=rv)
            t1 = time.perf_counter()
            solvers.solve_ref_as_configured(prob, ref_now[lo:hi], None, last[lo:hi], **kw_c)
            t_cpu += time.perf_counter() - t1
            done = hi
        out["cpu_baseline"] = {"value": done / t_cpu, "unit": "frames/s", "cores": 1, "kind": "port",
                               "sample": f"first {done} frames of the same workload, oracle restatement of the reference "
                                         f"objective + scipy SLSQP (ftol {prob.ftol:g}) standing in for nlopt, one process",
                               "host_cpus": os.cpu_count()}
        # the same solve fanned over host processes (frames are independent): what the reference could do on this box
        avail = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
        procs = int(os.environ.get("DEXR_CPU_PROCS", min(avail, 64)))
        if procs > 1:
            from oracle import cpu_worker

            per_proc = max(8, int(out["cpu_baseline"]["value"] * 6.0))  # ~6 s of work per process
            res = cpu_worker.run_all_cores(rel, ref_now, last, procs, per_proc)
            if res is not None:
                out["cpu_baseline_all_cores"] = {
                    "value": res[0] / res[1], "unit": "frames/s", "cores": procs, "kind": "port",
                    "sample": f"first {res[0]} frames, {procs} processes x {per_proc} frames started together, same "
                              f"solver as cpu_baseline; {avail} CPUs available to the process"}
    print(json.dumps(out))
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
