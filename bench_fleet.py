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
    m = fleet_measure(args)
    if m is not None:
        import bench_line

        bench_line.emit(fleet_check(*m, args))


def fleet_measure(args, batch=None, standalone=True):
    """Timed part of the mixed-fleet line: (record dict, checker context) on rank 0, None on the other ranks.  Touches
    nothing under oracle/.  standalone=False: called from bench.py's default command for the sub-record also.mixed_fleet
    (one GPU, no communicator: BASELINE configs[4]'s per-GPU slice)."""
    import torch

    import bench_data
    from dex_retargeting_amd.constants import DEFAULT_URDF_DIR
    from dex_retargeting_amd.fleet import MixedFleet
    from dex_retargeting_amd.retargeting_config import RetargetingConfig

    from bench import gather_records, job_env  # noqa: F401
    from dex_retargeting_amd.distributed import shard_by_model

    rank, local_rank, world, launched = job_env(args) if standalone else (0, int(os.environ.get("LOCAL_RANK", "0")), 1, False)
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    comm = None
    if standalone and (world > 1 or (launched and os.environ.get("DEXR_BENCH_DIST", "1") != "0")):
        from dex_retargeting_amd.distributed import native_comm

        comm = native_comm(rank, world)

    RetargetingConfig.set_default_urdf_dir(str(DEFAULT_URDF_DIR))
    seqs = [RetargetingConfig.load_from_file(os.path.join(bench_data.CONFIG_DIR, r)).build() for r in FLEET]
    fleet = MixedFleet([q.optimizer for q in seqs], device=str(dev))
    B = batch or args.batch
    seed = bench_data.SEED + 1000 * rank
    batches = []
    for j in range(N_BATCHES):
        # the GLOBAL batch of world x B frames (same ids on every rank); this rank's frames are its share under
        # distributed.shard_by_model: 1/N of EVERY robot's frames, so a batch that arrives sorted by robot
        # (--fleet-order sorted) loads the ranks exactly like an interleaved one.  world = 1: the identity.
        rng = np.random.default_rng(bench_data.SEED + 7 * j)
        mid_global = rng.integers(0, len(FLEET), world * B).astype(np.int32)
        if getattr(args, "fleet_order", "iid") == "sorted":
            mid_global = np.sort(mid_global)
        mine = shard_by_model(mid_global, world, len(FLEET))[rank]
        assert mine.size == B
        mid = np.ascontiguousarray(mid_global[mine])
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
    last_b = [None]

    def timed(pipe):
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
            last_b[0] = b

        for _ in range(args.warmup):
            step()
        if pipe is not None:
            pipe.finish()
        n[0] = 0
        ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        if comm is not None:
            comm.barrier(stream.cuda_stream)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        ev0.record(stream)
        for _ in range(args.steps):
            step()
        ev1.record(stream)
        if pipe is not None:
            pipe.finish()
        torch.cuda.synchronize()
        if comm is not None:
            comm.barrier(stream.cuda_stream)
        elapsed = time.perf_counter() - t0
        if comm is not None:
            elapsed = float(comm.max_f64([elapsed], stream.cuda_stream)[0])
        return elapsed, float(ev0.elapsed_time(ev1)) / args.steps

    coll = None
    if comm is not None:
        elapsed, step_ms, coll = gather_records(timed, comm, B, fleet.n_max, dev, args.steps, args.warmup, world, rank=rank)
    else:
        elapsed, step_ms = timed(None)
    if rank != 0:
        comm.close()
        return None
    last_b = last_b[0]
    # answers of the last step's batch (re-issued into a plain buffer) for the checker
    t_state.copy_(last_b["t_state0"])
    q = fleet.retarget(last_b["t_mid"], last_b["t_kp"], last_b["t_last"], t_state, out=torch.zeros_like(t_out)).cpu().numpy()
    last = last_b["t_last"].cpu().numpy()
    st_in = last_b["t_state0"].cpu().numpy().astype(np.uint32)
    mid, kp = last_b["mid"], last_b["kp"]
    bpf = 21 * 12 + 2 * 4 * fleet.n_max + 4 + 8  # keypoints + padded last/qpos rows + model id + DexPilot state in/out
    achieved = B * bpf / (step_ms * 1e-3) / 1e9
    import bench as _bench  # (pmc_counters: the committed rocprofv3 PMC summary of this command, keyed on batch + sources)

    traffic, n_valu, pmc_note = _bench.pmc_counters("mixed_fleet", B) if world == 1 else (None, None, None)
    out_json = {
        "metric": "retargeted frames/sec, mixed-fleet batch (BASELINE.json configs[4])",
        "value": world * B * args.steps / elapsed, "unit": "frames/s", "n_gpus": world, "steps": args.steps,
        "warmup": args.warmup, "ms_per_step": elapsed / args.steps * 1e3, "higher_is_better": True, "scaling": "weak",
        "vs_baseline": None, "dtype": "f32", "data": "synthetic",
        "config": {"workload": f"{WORKLOADS['mixed_fleet'][1]}; {B} frames/GPU, model id uniform at random per frame"
                               f"{' (global batch sorted by robot)' if getattr(args, 'fleet_order', 'iid') == 'sorted' else ''}, "
                               f"ranks take 1/N of every robot's frames (shard_by_model), human-keypoint refs, warm start = previous frame's solution; {N_BATCHES} staged batches rotated",
                   "models": FLEET, "batch_per_gpu": B, "fleet_order": getattr(args, "fleet_order", "iid"),
                   "frames_per_model_this_rank": np.bincount(mid, minlength=len(FLEET)).tolist(),
                   "collective": "none" if coll is None else coll["collective"],
                   "rccl_world_size": None if coll is None else coll["rccl_world_size"]},
        "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBPS, "unit": "GB/s",
                     "frac": achieved / HBM_PEAK_GBPS, "traffic": traffic,
                     "traffic_over_algorithmic": None if not traffic else traffic / (B * bpf),
                     "valu_issue_frac": None if not n_valu else n_valu * 2.0 / (1024 * step_ms * 1e-3 * 2.4e9), "pmc_note": pmc_note,
                     "kernel_ms": step_ms,
                     "algorithmic_bytes_per_frame": bpf,
                     "kernel": "dexr_retarget_multi_dev: device-side bucketing (3 small kernels) + one solve launch per model "
                               "over its index list, in-place rows"},
    }
    out_json["dtype"] = "per model: f32, or f64 kinematics + f32 Hessian"  # (tip / register kernels; Shadow DexPilot on the sixteen-lane kernel)
    if world == 1:
        # the same frames in the reference's own arithmetic (float64 throughout).  dexr_retarget_multi_dev runs the
        # float32 / mixed-precision kernels only, so: the batch bucketed by model up front (untimed torch index ops), then one
        # float64 launch per model (dexr_retarget_kp_dev, precision = 1) over its contiguous rows -- the four launches are timed.
        try:
            from dex_retargeting_amd import _lib

            o64 = _lib.default_options(precision=1)
            b0 = last_b
            per, kernels_used = [], []
            for m, sq in enumerate(seqs):
                idx = torch.nonzero(b0["t_mid"] == m).flatten()
                n_m, n_opt = int(idx.numel()), sq.optimizer.opt_dof
                mdl, which = fleet.models[m], "dexr_kernel<.., double> (register / tip kernel)"
                if mdl.kernel()[0] == _lib.KERNEL_WIDE:
                    # (the sixteen-lane family's all-float64 counterpart is the general kernel on the generic tables of the same
                    # config -- the register kernel dexr_kernel<24, double> spills 16 000 registers: 57 ms for these 32 768 rows)
                    sg = RetargetingConfig.load_from_file(os.path.join(bench_data.CONFIG_DIR, FLEET[m])).build()
                    sg.optimizer.use_generic_tables = True
                    mdl, which = sg.optimizer.device_model(), "dexr_gen_kernel (generic tables, float64 throughout)"
                kernels_used.append(f"{FLEET[m]}: {which}")
                per.append(dict(model=mdl, n=n_m, kp=b0["t_kp"][idx].contiguous(), last=b0["t_last"][idx][:, :n_opt].contiguous(),
                                st0=b0["t_state0"][idx].contiguous(), st=torch.zeros(n_m, dtype=torch.int32, device=dev),
                                out=torch.empty((n_m, n_opt), dtype=torch.float32, device=dev), idx=idx, n_opt=n_opt,
                                dex=sq.optimizer.retargeting_type == "DEXPILOT"))

            def f64_step():
                for p_ in per:
                    if p_["dex"]:
                        p_["st"].copy_(p_["st0"])
                    p_["model"].retarget_dev(p_["n"], p_["kp"].data_ptr(), 0, p_["last"].data_ptr(), p_["st"].data_ptr() if p_["dex"] else 0,
                                             p_["out"].data_ptr(), opts=o64, stream=stream.cuda_stream, keypoints=True)

            f64_step()
            torch.cuda.synchronize()
            s64 = max(2, min(args.steps, 3))
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            t0 = time.perf_counter()
            e0.record(stream)
            for _ in range(s64):
                f64_step()
            e1.record(stream)
            torch.cuda.synchronize()
            el = time.perf_counter() - t0
            q64 = np.zeros_like(q)
            for p_ in per:
                q64[p_["idx"].cpu().numpy(), : p_["n_opt"]] = p_["out"].cpu().numpy()
            dq = np.abs(q64.astype(np.float64) - q).max(1)
            out_json["f64"] = {"dtype": "f64", "value": B * s64 / el, "unit": "frames/s", "ms_per_step": el / s64 * 1e3, "steps": s64,
                               "event_ms_per_step": float(e0.elapsed_time(e1)) / s64,
                               "max_abs_dq_vs_default_rad": float(dq.max()), "p999_abs_dq_vs_default_rad": float(np.percentile(dq, 99.9)),
                               "kernels": kernels_used,
                               "note": "one float64 launch per model (dexr_retarget_kp_dev, precision = 1) over its rows of the same batch, "
                                       "bucketed by model beforehand (untimed): the fleet entry point itself runs the float32 / "
                                       "mixed-precision kernels only"}
        except Exception as e:
            out_json["f64"] = {"error": repr(e)}
    if comm is not None:
        comm.close()
    return out_json, dict(q=q, last=last, st_in=st_in, mid=mid, kp=kp, world=world, standalone=standalone, coll=coll)


def fleet_check(out_json, ctx, args):
    """Checker part (oracle/): 4 096 frames of each model against the float64 oracle, fanned over the host cores; the CPU
    port timed on a small sample of each model (standalone N = 1 runs only)."""
    from oracle import cases

    q, last, st_in, mid, kp, world, standalone, coll = (ctx[k] for k in ("q", "last", "st_in", "mid", "kp", "world", "standalone", "coll"))
    from oracle import jobs

    probs = [cases.problem_from_config(r) for r in FLEET]
    parity, cpu_t, cpu_n = {}, 0.0, 0
    n_par = 4096  # frames of each model compared with the oracle (fanned over the host cores)
    with jobs.host_pool() as pool:
        for m, (rel, pr) in enumerate(zip(FLEET, probs)):
            idx = np.nonzero(mid == m)[0][:n_par]
            ref = np.ascontiguousarray(cases.ref_from_keypoints(pr, kp[idx]), dtype=np.float32)
            la = np.ascontiguousarray(last[idx][:, : pr.n_opt])
            got = q[idx][:, : pr.n_opt].astype(np.float64)
            o = jobs.pooled_oracle_solve(rel, ref, la, st_in[idx] if pr.kind == "dexpilot" else None, got, pool=pool)
            dq = np.abs(got - o["want"]).max(1)
            far = dq >= 1e-4
            parity[rel] = {"subset": len(idx), "max_abs_dq_rad": float(dq.max()), "frac_within_1e-4": float((~far).mean()),
                           "other_minimum": {"frames": int(far.sum()), "worse": int((o["F_gpu"][far] > o["F_want"][far] + 1e-9).sum())},
                           "max_abs_dq_rad_same_minimum": float(dq[~far].max()) if (~far).any() else None}
            if not args.no_cpu_baseline and world == 1 and standalone:
                from oracle import cport

                kw = jobs._kw(pr, ref[:128], st_in[idx][:128]) if pr.kind == "dexpilot" else {}
                cp = cport.CProblem(pr)
                t1 = time.perf_counter()
                cport.solve_ref_as_configured_c(cp, ref[:128], None, la[:128], **kw)
                cpu_t += time.perf_counter() - t1
                cpu_n += min(128, len(idx))
    out_json["parity"] = parity
    if cpu_n:
        out_json["cpu_baseline"] = {"value": cpu_n / cpu_t, "unit": "frames/s", "cores": 1, "kind": "port",
                                    "sample": "128 frames of each of the four models; the reference's per-frame procedure with the "
                                              "oracle's plain-C closure + scipy's compiled SLSQP (see bench.py cpu_baseline)"}
    if coll is not None:
        out_json["multi_gpu"] = coll
    return out_json
