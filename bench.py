#!/usr/bin/env python3
"""bench.py -- retargeted frames/s of the batched HIP solver on N MI355X GPUs (one process per GPU).

    python bench.py --gpus 1 --steps 20 --warmup 3
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W

A "step" is one pass of the hot path over one batch: B = 65 536 independent frames per GPU (weak scaling) of the
workload BASELINE.json quotes the metric on -- Allegro right hand, VectorOptimizer -- solved to the tight
tolerance, inputs resident in HBM when the timed region starts, result qpos resident (and, for N > 1, all-gathered
with one RCCL all-gather per step -- `dexr_allgather` of libdexr's C-ABI -- ) when it ends.  `--gpus N` without a
launcher around it starts the N ranks itself (re-executes under torch.distributed.run).  Consecutive steps solve DIFFERENT batches (four pre-staged
batches are rotated), so no step re-solves the previous step's inputs.

Headline workload (synthetic, seeded): keypoints = frame (b mod 621) of the human fixture
+ N(0, 2 mm); ref_value = kp[task] - kp[origin] formed inside the kernel (profile_online_retargeting.py:24-30);
last_qpos = the solver's own answer for the neighbouring frame (b-1), i.e. the warm start a running sequence has
(seq_retarget.py:124).  float32 arithmetic (dtype "f32"); the same line carries

* "f64":        the same config timed with float64 arithmetic throughout (the reference's arithmetic type);
* "cold_start": the reference's own test regime (tests/test_optimizer.py:27-81: reachable targets, start sigma = 0.5);
* "online_teleop": the reference's own benchmark shape (profile_online_retargeting.py:18-36): 621 fixture frames, ONE
                SeqRetargeting.retarget(ref) per frame, per robot: fps, mean / p99 ms per frame, the bare C-ABI call,
                and the compiled CPU port on the same loop (BASELINE configs[0]);
* "multi_gpu":  (launched by torch.distributed.run) the same steps with the gather on the solve stream and replayed
                from one captured HIP graph, RCCL version, an xGMI estimate;
* "also":       the other two single-GPU BASELINE configs (Shadow DexPilot, LEAP position), each with its own
                roofline / HBM traffic / parity block, and the mixed fleet;
* "general_kernel": a model beyond the fixed-size tables (arm + Shadow hand URDF, 37 variables in one component, 21 rows)
                on the general kernel, with its parity block;
* "parity":     max |dq| against the float64 oracle on a subset, and the distance to the reference-as-configured
                SLSQP answers (oracle/, checker only -- imported after every timed region).

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
FP64_VALU_PEAK_TFLOPS = 78.6
N_BATCHES = 4  # pre-staged input batches rotated over the steps
# what the sixteen-lane kernel computes in (csrc/dexr_wide.hpp:8-24): kinematics, residuals, objective value in float64; gradient,
# Hessian, factorisation in float32.  (The headline's tip kernel is float32 throughout: "f32".)
WIDE_DTYPE = "f64 kinematics + f32 Hessian"

FLEET = ["teleop/allegro_hand_right.yml", "teleop/shadow_hand_right_dexpilot.yml", "teleop/leap_hand_right.yml",
         "teleop/ability_hand_right.yml"]  # BASELINE.json configs[4]: 4 URDFs in one batch

WORKLOADS = {
    "allegro_vector": ("teleop/allegro_hand_right.yml", "Allegro right hand, VectorOptimizer"),
    "shadow_dexpilot": ("teleop/shadow_hand_right_dexpilot.yml", "Shadow right hand (24 DoF), DexPilotOptimizer"),
    "leap_position": ("offline/leap_hand_right.yml", "LEAP right hand + 6 free joints, PositionOptimizer"),
    "mixed_fleet": (None, "Mixed fleet: Allegro vector + Shadow DexPilot + LEAP vector + Ability vector, frames interleaved"),
}
KERNEL_NAMES = {0: "dexr_kernel (one lane per frame and component, Hessian in registers)",
                1: "dexr_quad_kernel (four lanes per frame)", 2: "dexr_big_kernel (Hessian in LDS)",
                3: "dexr_red_kernel (reduced variables: Hessian in registers, kinematics in LDS)",
                4: "dexr_wide_kernel (sixteen lanes per frame: chain-parallel FK, 4 x 4 lane grid Hessian / Cholesky)"}


def algorithmic_bytes_per_frame(n_opt: int, dexpilot: bool, n_rows: int, keypoints: bool) -> int:
    """Compulsory HBM traffic of one frame: the input rows (21 raw keypoints =
    252 B through dexr_retarget_kp_dev, or n_ref ready-made ref_value rows) + last_qpos in + qpos out (+ 4 B DexPilot
    state in and out)."""
    return (21 if keypoints else n_rows) * 12 + n_opt * 4 + n_opt * 4 + (8 if dexpilot else 0)


def pmc_counters(name: str, batch: int):
    """(traffic bytes per step, SQ_INSTS_VALU per step, note) from the committed rocprofv3 PMC summary profiles/pmc_<name>.json
    (tools/profile_round.sh <name>): 2 x FETCH_SIZE + WRITE_SIZE summed over EVERY kernel of a step (solve kernels and the
    ordering / bucketing kernels around them; gfx950 correction per MI355X_MICROARCH.md).  Attached only while the summary was
    taken on this batch size and on the sources that build the workload's kernels (dex_retargeting_amd/_build.source_hash)."""
    path = os.path.join(REPO, "profiles", f"pmc_{name}.json")
    if not os.path.exists(path):
        return None, None, f"no committed PMC summary (profiles/pmc_{name}.json)"
    from dex_retargeting_amd._build import source_hash

    pmc = json.load(open(path))
    if pmc.get("batch") != batch:
        return None, None, "committed PMC summary is for another batch size"
    if pmc.get("source_sha16") != source_hash(name):
        return None, None, "committed PMC summary was taken on other kernel sources (source_sha16 mismatch): counters withheld"
    return pmc.get("hbm_bytes_per_launch"), pmc.get("SQ_INSTS_VALU"), None


def algorithmic_flops_per_pass(compiled) -> float:
    """FP operations of ONE solver pass (FK + value/gradient/Hessian + factorisation + step) over all components of a
    frame, counted from the compiled tables: per component with n joints and T terms
    FK 100 n  +  T (20 + 18 n + 18 n(n+1)/2)  +  n^3/3 + 2 n^2  +  10 n."""
    total = 0.0
    for c in compiled.comps:
        n, t = int(c["n_joint"]), int(c["n_term"])
        total += 100 * n + t * (20 + 18 * n + 18 * n * (n + 1) / 2) + n ** 3 / 3 + 2 * n * n + 10 * n
    return total


class Workload:
    """One named single-model workload on one GPU: model handle + N_BATCHES staged input batches in HBM."""

    def __init__(self, name, rank, B, dev, torch):
        import bench_data
        from dex_retargeting_amd.constants import DEFAULT_URDF_DIR
        from dex_retargeting_amd.retargeting_config import RetargetingConfig

        RetargetingConfig.set_default_urdf_dir(str(DEFAULT_URDF_DIR))
        self.name, self.B, self.dev, self.torch = name, B, dev, torch
        self.rel, self.title = WORKLOADS[name]
        self.seq = RetargetingConfig.load_from_file(os.path.join(bench_data.CONFIG_DIR, self.rel)).build()
        self.opt = self.seq.optimizer
        self.model = self.opt.device_model()
        self.n_opt = self.opt.opt_dof
        self.n_ref = int(self.opt.compiled_model().header["n_ref"])
        self.dexpilot = self.opt.retargeting_type == "DEXPILOT"
        self.seed = bench_data.SEED + 1000 * rank
        self.stream = torch.cuda.current_stream()
        self.t_iters = torch.zeros(B, dtype=torch.int32, device=dev)
        self.t_status = torch.zeros(B, dtype=torch.int32, device=dev)
        self.t_state = torch.zeros(B, dtype=torch.int32, device=dev) if self.dexpilot else None
        self.t_q = torch.empty((B, self.n_opt), dtype=torch.float32, device=dev)
        self.tracking = self._stage_tracking(bench_data)
        self.cold = None

    def _stage_tracking(self, bench_data):
        """N_BATCHES batches of B frames: different fixture phase and noise per batch; the warm start of frame b is
        the solver's own (untimed) answer for frame b-1 of the same batch."""
        torch, B = self.torch, self.B
        mid = np.repeat(self.seq.joint_limits.mean(1)[None], B, 0).astype(np.float32)
        out = []
        for j in range(N_BATCHES):
            kp = bench_data.human_keypoints(B + 1, seed=self.seed + 17 * j, offset=155 * j)
            st = np.zeros(B, np.uint32) if self.dexpilot else None
            last = self.model.retarget(np.ascontiguousarray(kp[:-1]), None, mid, state=st, keypoints=True)
            out.append(dict(kind="kp", host_in=np.ascontiguousarray(kp[1:]), host_last=last, host_state=st,
                            t_in=torch.from_numpy(np.ascontiguousarray(kp[1:])).to(self.dev),
                            t_last=torch.from_numpy(last).to(self.dev),
                            t_state0=None if st is None else torch.from_numpy(st.astype(np.int32)).to(self.dev)))
        return out

    def stage_cold(self):
        """The reference's own test regime (tests/test_optimizer.py:27-81): reachable targets, start sigma = 0.5 rad."""
        import bench_data

        torch = self.torch
        out = []
        for j in range(N_BATCHES):
            ref, start = bench_data.reachable_batch(self.seq, self.B, 0.5, seed=self.seed + 31 * j + 5)
            out.append(dict(kind="ref", host_in=ref, host_last=start, host_state=None,
                            t_in=torch.from_numpy(ref).to(self.dev), t_last=torch.from_numpy(start).to(self.dev),
                            t_state0=torch.zeros(self.B, dtype=torch.int32, device=self.dev) if self.dexpilot else None))
        self.cold = out
        return out

    def launch(self, batch, out, opts=None, diagnostics=False):
        if self.dexpilot:
            self.t_state.copy_(batch["t_state0"])
        self.model.retarget_dev(self.B, batch["t_in"].data_ptr(), 0, batch["t_last"].data_ptr(),
                                self.t_state.data_ptr() if self.dexpilot else 0, out.data_ptr(),
                                status_ptr=self.t_status.data_ptr() if diagnostics else 0,
                                iters_ptr=self.t_iters.data_ptr() if diagnostics else 0, opts=opts,
                                stream=self.stream.cuda_stream, keypoints=batch["kind"] == "kp")

    def diagnostics(self, batches, opts=None):
        """Untimed: iteration statistics of every staged batch (frame level: the maximum over the frame's components)."""
        torch = self.torch
        its, conv, tilemax = [], 0, []
        for b in batches:
            self.launch(b, self.t_q, opts=opts, diagnostics=True)
            torch.cuda.synchronize()
            it = self.t_iters.cpu().numpy()
            its.append(it)
            conv += int((self.t_status == 0).sum())
            tilemax.append(it[: self.B // 64 * 64].reshape(-1, 64).max(1))
        it = np.concatenate(its)
        tm = np.concatenate(tilemax)
        return {"iters_mean": float(it.mean()), "iters_max": int(it.max()), "converged_frac": conv / (self.B * len(batches)),
                # passes a 64-frame tile executes (its slowest frame) vs. passes its frames need: the share of issued
                # lane-passes that do useful work when a wave owns a fixed tile (tile mode)
                "tile_max_mean": float(tm.mean()),
                "active_lane_fraction": float((it[: len(tm) * 64] + 1).sum() / (64.0 * (tm + 1).sum()))}

    def timed(self, batches, steps, warmup, opts=None, pipe=None, comm=None):
        """warmup untimed steps, then exactly `steps` steps bracketed by barrier + synchronize; returns
        (elapsed wall seconds = MAX over ranks, mean launch ms from HIP events on the launch stream).  `comm` is the
        native communicator (libdexr: dexr_comm_barrier / dexr_comm_max_f64 on RCCL); `pipe` a NativeGather."""
        torch = self.torch
        n = [0]

        def step():
            k = n[0]
            out = self.t_q if pipe is None else pipe.shard(k)
            self.launch(batches[k % len(batches)], out, opts=opts)
            if pipe is not None:
                pipe.gather(k)
            n[0] += 1

        for _ in range(warmup):
            step()
        if pipe is not None:
            pipe.finish()
        n[0] = 0  # the timed steps start a fresh group of a k-step gather (the batches keep rotating)
        ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        if comm is not None:
            comm.barrier(self.stream.cuda_stream)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        ev0.record(self.stream)
        for _ in range(steps):
            step()
        ev1.record(self.stream)
        if pipe is not None:
            pipe.finish()
        torch.cuda.synchronize()
        if comm is not None:
            comm.barrier(self.stream.cuda_stream)
        elapsed = time.perf_counter() - t0
        kernel_ms = float(ev0.elapsed_time(ev1)) / steps
        if comm is not None:
            elapsed = float(comm.max_f64([elapsed], self.stream.cuda_stream)[0])
        return elapsed, kernel_ms

    def sustained(self, batches, seconds=6.0, chunk=4000):
        """A rate the driver can see from outside: back-to-back launches of the workload for >= `seconds` (the
        staged batches rotating), HIP events on the launch stream every `chunk` steps.  ms/step from the events, from the wall
        clock, and of the first / last tenth of the run (clock drift under sustained load)."""
        torch = self.torch
        est = 0.05e-3
        for _ in range(3):
            self.launch(batches[0], self.t_q)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for k in range(200):
            self.launch(batches[k % len(batches)], self.t_q)
        torch.cuda.synchronize()
        est = max((time.perf_counter() - t0) / 200, 1e-6)
        n_chunks = max(10, int(np.ceil(max(seconds / est, 40000) / chunk)))
        evs = [torch.cuda.Event(enable_timing=True) for _ in range(n_chunks + 1)]
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        k = 0
        evs[0].record(self.stream)
        for c in range(n_chunks):
            for _ in range(chunk):
                self.launch(batches[k % len(batches)], self.t_q)
                k += 1
            evs[c + 1].record(self.stream)
        torch.cuda.synchronize()
        wall = time.perf_counter() - t0
        per = np.array([evs[c].elapsed_time(evs[c + 1]) / chunk for c in range(n_chunks)])
        tenth = max(1, n_chunks // 10)
        steps = n_chunks * chunk
        # the contract's own window -- 20 steps between two synchronizes -- on the chip as the long run leaves it, and again after
        # 100 ms of idling: the difference between `ms_per_step` of the line and the sustained rate is the power state a short
        # burst finds the chip in (tools probe, round 6: 39.2 us per step right after activity, 41.1 / 41.9 after 10 / 100 ms idle)
        def window(idle_s):
            torch.cuda.synchronize()
            time.sleep(idle_s)
            t1 = time.perf_counter()
            for k2 in range(20):
                self.launch(batches[k2 % len(batches)], self.t_q)
            torch.cuda.synchronize()
            return (time.perf_counter() - t1) / 20 * 1e3
        w_busy, w_idle = window(0.0), window(0.1)
        return {"steps": steps, "seconds": wall, "ms_per_step": float(evs[0].elapsed_time(evs[-1])) / steps, "wall_ms_per_step": wall / steps * 1e3,
                "first_decile_ms_per_step": float(per[:tenth].mean()), "last_decile_ms_per_step": float(per[-tenth:].mean()),
                "min_chunk_ms_per_step": float(per.min()), "max_chunk_ms_per_step": float(per.max()),
                "window20_right_after_ms_per_step": w_busy, "window20_after_100ms_idle_ms_per_step": w_idle,
                "value": self.B * steps / wall, "unit": "frames/s",
                "note": f"{steps} consecutive launches ({n_chunks} chunks of {chunk} between HIP events), {len(batches)} staged batches rotating, single stream"}

    def timed_two_streams(self, batches, steps, warmup):
        """Throughput when consecutive (independent) batches are issued alternately on TWO HIP streams, each with its
        own output / DexPilot-state buffers: the tail of one launch (its slowest frames, a few waves) overlaps the bulk
        of the next.  Same K steps, same barrier + synchronize bracket; reported beside -- never instead of -- the
        single-stream figure."""
        torch = self.torch
        streams = [self.stream, torch.cuda.Stream(device=self.dev)]
        outs = [self.t_q, torch.empty_like(self.t_q)]
        states = [self.t_state, None if self.t_state is None else torch.empty_like(self.t_state)]
        n = [0]

        def step():
            k = n[0]
            i = k & 1
            b = batches[k % len(batches)]
            with torch.cuda.stream(streams[i]):
                if self.dexpilot:
                    states[i].copy_(b["t_state0"])
                self.model.retarget_dev(self.B, b["t_in"].data_ptr(), 0, b["t_last"].data_ptr(),
                                        states[i].data_ptr() if self.dexpilot else 0, outs[i].data_ptr(),
                                        stream=streams[i].cuda_stream, keypoints=b["kind"] == "kp")
            n[0] += 1

        torch.cuda.synchronize()
        for _ in range(warmup):
            step()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(steps):
            step()
        torch.cuda.synchronize()
        elapsed = time.perf_counter() - t0
        return {"value": self.B * steps / elapsed, "unit": "frames/s", "ms_per_step": elapsed / steps * 1e3, "streams": 2,
                "note": "the same K steps issued alternately on two HIP streams (independent batches, separate output "
                        "and state buffers): launch tails overlap the next launch; throughput figure, the headline "
                        "value above is the single-stream one"}

    def timed_graph(self, batches, steps, comm):
        """[solve -> dexr_allgather] of len(batches) consecutive steps captured into ONE HIP graph (everything on the
        capture stream: no host call between the launches) and replayed steps / len(batches) times inside the usual
        barrier + synchronize bracket: the N > 1 step without the host's per-launch cost."""
        torch = self.torch
        from dex_retargeting_amd.distributed import NativeGather

        G = len(batches)
        if steps % G:
            return {"skipped": f"steps must be a multiple of {G}"}
        ng = NativeGather(comm, self.B, self.n_opt, self.dev, depth=G, overlap=False)
        s = torch.cuda.Stream(device=self.dev)
        states = [None if self.t_state is None else torch.empty_like(self.t_state) for _ in range(G)]

        def enqueue():
            for k, b in enumerate(batches):
                if self.dexpilot:
                    states[k].copy_(b["t_state0"])
                out = ng._shard[k]
                self.model.retarget_dev(self.B, b["t_in"].data_ptr(), 0, b["t_last"].data_ptr(),
                                        states[k].data_ptr() if self.dexpilot else 0, out.data_ptr(),
                                        stream=s.cuda_stream, keypoints=b["kind"] == "kp")
                comm.allgather(out.data_ptr(), ng._full[k].data_ptr(), ng.bytes_per_rank, s.cuda_stream)

        s.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(s):
            enqueue()  # warm-up outside the capture (RCCL sets its channels up on first use)
        torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, stream=s):
            enqueue()
        g.replay()
        torch.cuda.synchronize()
        comm.barrier(self.stream.cuda_stream)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(steps // G):
            g.replay()
        torch.cuda.synchronize()
        comm.barrier(self.stream.cuda_stream)
        elapsed = float(comm.max_f64([time.perf_counter() - t0], self.stream.cuda_stream)[0])
        return {"value": comm.world * self.B * steps / elapsed, "unit": "frames/s", "ms_per_step": elapsed / steps * 1e3,
                "note": f"{G} steps of [solve -> all-gather] captured into one HIP graph, replayed {steps // G} times"}

    def roofline(self, kernel_ms, iters_mean, batch_kind="kp", precision="f32", world=1):
        bpf = algorithmic_bytes_per_frame(self.n_opt, self.dexpilot, self.n_ref, batch_kind == "kp")
        achieved = self.B * bpf / (kernel_ms * 1e-3) / 1e9
        flops_frame = algorithmic_flops_per_pass(self.opt.compiled_model()) * (iters_mean + 1.0)  # +1: start point's model
        tf = self.B * flops_frame / (kernel_ms * 1e-3) / 1e12
        peak_tf = FP32_VALU_PEAK_TFLOPS if precision == "f32" else FP64_VALU_PEAK_TFLOPS
        # HBM bytes per step as counted by rocprofv3 PMC passes of this same command (tools/profile_round.sh writes the
        # summary, committed under profiles/): see pmc_counters.  One summary per record: <workload>, <workload>_f64,
        # <workload>_cold.
        pmc_name = self.name + ("_f64" if precision == "f64" else "") + ("_cold" if batch_kind == "ref" else "")
        traffic, valu_frac, pmc_note = None, None, None
        if world == 1:
            traffic, n_valu, pmc_note = pmc_counters(pmc_name, self.B)
            if n_valu:
                # a wave64 VALU instruction issues over 2 cycles on a SIMD-32 (MI355X_MICROARCH.md, "Wave scheduling":
                # 32 lanes/cycle x 2; 157.3 TF = 64 FLOP/clk/SIMD); 1024 SIMDs at 2.4 GHz.  A lower bound on the busy
                # share: packed-f32 and f64 instructions take 4 (round 3 charged 4 to every instruction: 2x too high)
                valu_frac = n_valu * 2.0 / (1024 * kernel_ms * 1e-3 * 2.4e9)
        fam, bucket, chain = self.model.kernel()
        kname = KERNEL_NAMES[fam] + f", bucket {bucket}" + (", serial-chain specialisation" + (" with the tip pass" if chain == 2 else "") if chain else "")
        if precision == "f64":
            kname = f"dexr_kernel<{bucket}, double" + (", CHAIN, TIP> (the tip pass of the serial-chain kernel in float64 arithmetic, "
                                                        "dexr_tip.hpp)" if chain == 2 else "> (register kernel, float64 arithmetic)")
        return {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBPS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBPS,
                "traffic": traffic, "traffic_over_algorithmic": None if not traffic else traffic / (self.B * bpf),
                "valu_issue_frac": valu_frac, "pmc_note": pmc_note,
                "valu": {"achieved": tf, "peak": peak_tf, "unit": "TFLOP/s", "frac": tf / peak_tf,
                         "algorithmic_flops_per_frame": flops_frame},
                "kernel": kname, "kernel_ms": kernel_ms, "algorithmic_bytes_per_frame": bpf}


def sequence_record(wl, torch, T=100, reps=3):
    """Sequence mode: B lock-step sequences of T frames each.  Fused = two kernel launches
    per T x B frames (dexr_retarget_seq_dev: every lane loops over its sequence's frames carrying last_qpos;
    dexr_seq_compose_dev: robot qpos, mimic fill, low-pass filter).  Frame-by-frame = one solve launch + ~10 torch
    element-wise launches per frame, replayed from a HIP graph.  Keypoints: sequence b plays the human fixture from
    phase 37 b (+ 2 mm noise), so consecutive frames of a sequence are consecutive fixture frames."""
    import bench_data
    from dex_retargeting_amd.retargeting_config import RetargetingConfig

    B, dev = wl.B, wl.dev
    cfg = RetargetingConfig.load_from_file(os.path.join(bench_data.CONFIG_DIR, wl.rel))
    fixture = torch.from_numpy(np.load(bench_data.HUMAN_FIXTURE)).to(dev)
    g = torch.Generator(device=dev)
    g.manual_seed(bench_data.SEED)
    idx = (torch.arange(B, device=dev)[None, :] * 37 + torch.arange(T + 1, device=dev)[:, None]) % fixture.shape[0]
    kp = fixture[idx] + 2e-3 * torch.randn((T + 1, B, 21, 3), generator=g, device=dev)
    kp[:, :, 0] = 0.0
    kp = kp.contiguous()
    out = {}
    fused = cfg.build_device(B)
    fused.retarget_keypoints(kp[0])  # frame 0: from the limit midpoint (untimed), initialises the filter
    raw = torch.empty((T, B, fused.n_opt), dtype=torch.float32, device=dev)
    res = torch.empty((T, B, fused.robot_qpos.shape[1]), dtype=torch.float64, device=dev)
    stream = torch.cuda.current_stream()
    state0 = (fused.last_qpos.clone(), fused.filtered.clone(), fused.state.clone())

    def rewind(obj):
        obj.last_qpos.copy_(state0[0])
        obj.filtered.copy_(state0[1])
        obj.state.copy_(state0[2])

    fused.retarget_sequence(kp[1:], out=res, raw_out=raw)  # warm-up
    ms = []
    for _ in range(reps):
        rewind(fused)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(stream)
        fused.retarget_sequence(kp[1:], out=res, raw_out=raw)
        e1.record(stream)
        torch.cuda.synchronize()
        ms.append(e0.elapsed_time(e1))
    out["fused"] = {"ms_per_T_frames": float(np.median(ms)), "frames_per_s": B * T / (float(np.median(ms)) * 1e-3),
                    "launches_per_T_frames": 2}
    res_fused = res.clone()
    # frame-by-frame, captured into a HIP graph in chunks of Tg frames (capturing 100 frames x ~12 launches is slow)
    step = cfg.build_device(B)
    step.retarget_keypoints(kp[0])
    Tg = 20
    buf = torch.empty((Tg, B, 21, 3), dtype=torch.float32, device=dev)
    graph, gout = step.capture(buf)
    ms = []
    for r in range(reps):
        rewind(step)
        t_ms = 0.0
        for c in range(T // Tg):
            buf.copy_(kp[1 + c * Tg: 1 + (c + 1) * Tg])
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(stream)
            graph.replay()
            e1.record(stream)
            torch.cuda.synchronize()
            t_ms += e0.elapsed_time(e1)
            if r == 0 and c == T // Tg - 1:
                diff = (gout - res_fused[c * Tg:(c + 1) * Tg]).abs().amax(dim=2)
                out["max_abs_diff_fused_vs_stepwise_rad_p99"] = float(torch.quantile(diff.flatten()[:1000000].float(), 0.99))
        ms.append(t_ms)
    out["frame_by_frame_hip_graph"] = {"ms_per_T_frames": float(np.median(ms)), "frames_per_s": B * T / (float(np.median(ms)) * 1e-3),
                                       "launches_per_T_frames": "T x (1 solve + ~10 element-wise), replayed from HIP graphs of 20 frames"}
    out.update({"T": T, "sequences": B, "unit": "frames/s", "config_file": wl.rel})
    return out


ONLINE_ROBOTS = [("teleop/allegro_hand_right.yml", "Allegro vector (BASELINE configs[0])"),
                 ("teleop/shadow_hand_right.yml", "Shadow vector"),
                 ("teleop/leap_hand_right.yml", "LEAP vector"),
                 ("teleop/allegro_hand_right_dexpilot.yml", "Allegro DexPilot"),
                 ("teleop/shadow_hand_right_dexpilot.yml", "Shadow DexPilot")]


def online_run(rel):
    """The reference's own benchmark shape (/root/reference/example/profiling/profile_online_retargeting.py:18-36,50-73):
    the 621-frame human fixture, ONE SeqRetargeting.retarget(ref_value) call per frame (B = 1, host arrays in and out),
    only the call itself inside the timer.  Returns per-frame seconds, the answers, and the per-frame seconds of the bare
    C-ABI call (dexr_retarget through ctypes, same inputs) for the same frames."""
    import bench_data
    from dex_retargeting_amd.retargeting_config import RetargetingConfig

    data = np.load(bench_data.HUMAN_FIXTURE)
    seq = RetargetingConfig.load_from_file(os.path.join(bench_data.CONFIG_DIR, rel)).build()
    opt = seq.optimizer
    indices = opt.target_link_human_indices
    position = opt.retargeting_type == "POSITION"
    # untimed: compile the tables, create the device model and run one call through the handle's staging buffers
    # (through the C-ABI directly, so that neither last_qpos nor the DexPilot state of `seq` has seen a frame)
    r0 = data[0][indices, :] if position else data[0][indices[1, :], :] - data[0][indices[0, :], :]
    opt.device_model().retarget(r0[None].astype(np.float32), None, seq.last_qpos[None].astype(np.float32),
                                state=np.zeros(1, np.uint32) if opt.retargeting_type == "DEXPILOT" else None)
    dt, qs, refs, lasts, raws = [], [], [], [], []
    for joint_pos in data:
        ref_value = joint_pos[indices, :] if position else joint_pos[indices[1, :], :] - joint_pos[indices[0, :], :]
        lasts.append(np.clip(seq.last_qpos, seq.joint_limits[:, 0], seq.joint_limits[:, 1]).astype(np.float32))
        tic = time.perf_counter()
        q = seq.retarget(ref_value)
        dt.append(time.perf_counter() - tic)
        qs.append(q)
        raws.append(np.asarray(seq.last_qpos, dtype=np.float32).copy())  # the optimizer's own answer (unfiltered, target-joint order)
        refs.append(ref_value.astype(np.float32))
    # the bare C-ABI call on the same (ref, last) pairs: what a C / C++ caller of libdexr.so pays per frame
    model = opt.device_model()
    dexpilot = opt.retargeting_type == "DEXPILOT"
    st = np.zeros(1, np.uint32) if dexpilot else None
    dt_abi = []
    for r, l in zip(refs, lasts):
        r1, l1 = r[None], l[None]
        tic = time.perf_counter()
        model.retarget(r1, None, l1, state=st)
        dt_abi.append(time.perf_counter() - tic)
    return np.array(dt), np.array(qs), np.array(dt_abi), np.array(refs), np.array(lasts), seq, np.array(raws)


def online_record():
    """Sub-record `online_teleop` (BASELINE configs[0]; SURVEY.md row 13 MEASUREMENT TEMPLATE).  Returns (record, context
    for online_cpu_port); touches nothing under oracle/."""
    out = {"frames": 621, "protocol": "profile_online_retargeting.py:18-36: one SeqRetargeting.retarget(ref_value) per fixture "
                                    "frame, perf_counter around the call only, B = 1, host arrays",
           "robots": {}}
    ctx = None
    for rel, title in ONLINE_ROBOTS:
        try:
            dt, qs, dt_abi, refs, lasts, seq, raws = online_run(rel)
        except Exception as e:  # never lose the headline line to a sub-record
            out["robots"][rel] = {"error": repr(e)}
            continue
        rec = {"title": title, "fps": len(dt) / dt.sum(), "mean_ms": float(dt.mean() * 1e3),
               "p50_ms": float(np.percentile(dt, 50) * 1e3), "p99_ms": float(np.percentile(dt, 99) * 1e3),
               "max_ms": float(dt.max() * 1e3),
               "c_abi_call": {"mean_ms": float(dt_abi.mean() * 1e3), "p50_ms": float(np.percentile(dt_abi, 50) * 1e3),
                              "p99_ms": float(np.percentile(dt_abi, 99) * 1e3),
                              "note": "dexr_retarget (host pointers) alone, through ctypes: pack -> one H2D -> solve "
                                      "kernel -> one D2H on the handle's private stream -> hipStreamSynchronize"}}
        ctx = ctx or []
        ctx.append(dict(rel=rel, refs=refs, lasts=lasts, raws=raws, lo=seq.joint_limits[:, 0].copy(), hi=seq.joint_limits[:, 1].copy()))
        out["robots"][rel] = rec
    return out, ctx


def online_cpu_port(rec, ctxs):
    """Checker-side leg of `online_teleop`: the CPU port on the SAME loop, for every robot of the record (oracle/: imported
    here, in the checker section of the run, after every GPU timing).  DexPilot robots carry their projection state from
    frame to frame like the reference's optimizer object does (optimizer.py:466-476)."""
    from oracle import cases, cport

    for ctx in ctxs:
        rel, refs, lasts, lo, hi = ctx["rel"], ctx["refs"], ctx["lasts"], ctx["lo"], ctx["hi"]
        prob = cases.problem_from_config(rel)
        cp = cport.CProblem(prob)
        n_cpu = len(refs)
        last = lasts[0].astype(np.float64)
        proj = np.zeros((1, prob.n_pair), bool) if prob.kind == "dexpilot" else None
        t_cpu = []
        for i in range(n_cpu):
            tic = time.perf_counter()
            kw = {}
            if proj is not None:
                w, rv, proj = prob.dexpilot_preamble(refs[i][None], proj)
                kw = dict(weights=w, dexpilot_ref=rv)
            q_ref, _ = cport.solve_ref_as_configured_c(cp, refs[i][None], None, np.clip(last, lo, hi)[None].astype(np.float32), **kw)
            t_cpu.append(time.perf_counter() - tic)
            last = q_ref[0].astype(np.float64)
        t_cpu = np.array(t_cpu)
        rec["robots"][rel]["cpu_port_same_loop"] = {
            "frames": n_cpu, "fps": n_cpu / t_cpu.sum(), "mean_ms": float(t_cpu.mean() * 1e3),
            "p99_ms": float(np.percentile(t_cpu, 99) * 1e3), "kind": "port", "cores": 1,
            "note": "the oracle's plain-C closure + scipy's compiled SLSQP at the reference's ftol, its own warm-start chain over "
                    "the 621 fixture frames (the reference's loop, profile_online_retargeting.py:18-36, with compiled stand-ins "
                    "for pinocchio / nlopt and no torch overhead)"}


def frame_parity(rel, refs, lasts, q_gpu, state_in=None, pool=None):
    """Checker: `q_gpu` (B, n_opt) against the float64 oracle minimiser of F from the same (ref, last[, DexPilot state]) frame by
    frame -- the block every sub-record carries: fraction within 1e-4 rad, max |dq| of those, and the frames that sit in ANOTHER
    local minimum (counted, with how many of them have the higher objective)."""
    from oracle import jobs

    q64 = np.asarray(q_gpu, dtype=np.float64)
    o = jobs.pooled_oracle_solve(rel, np.ascontiguousarray(refs, dtype=np.float32), np.ascontiguousarray(lasts, dtype=np.float32),
                                 state_in, q64, chunk=64, pool=pool)
    dq = np.abs(q64 - o["want"]).max(1)
    far = dq >= 1e-4
    return {"subset": int(len(dq)), "frac_within_1e-4": float((~far).mean()), "max_abs_dq_rad": float(dq[~far].max()) if (~far).any() else None,
            "median_abs_dq_rad": float(np.median(dq)),
            "other_minimum": {"frames": int(far.sum()), "worse": int((far & (o["F_gpu"] > o["F_want"] + 1e-10)).sum())}}


def online_parity(rec, ctxs):
    """Checker-side leg of `online_teleop` (VERDICT r5 #1c): every one of the 621 one-frame calls against the oracle from the SAME
    last_qpos (the loop's own warm-start chain) -- the launch shape every SeqRetargeting.retarget() of a 9-32-joint model takes
    (one frame per wave + ladder).  DexPilot robots: the projection bits each frame started from are rebuilt from the targets alone
    (optimizer.py:466-476: the state never depends on the answers)."""
    from oracle import cases, jobs

    with jobs.host_pool() as pool:
        for ctx in ctxs:
            rel, refs, lasts, raws = ctx["rel"], ctx["refs"], ctx["lasts"], ctx["raws"]
            prob = cases.problem_from_config(rel)
            st_in = None
            if prob.kind == "dexpilot":
                proj = np.zeros((1, prob.n_pair), bool)
                st_in = np.zeros(len(refs), np.uint32)
                for i in range(len(refs)):
                    st_in[i] = int((proj[0].astype(np.uint64) << np.arange(prob.n_pair, dtype=np.uint64)).sum())
                    _, _, proj = prob.dexpilot_preamble(refs[i][None], proj)
            try:
                rec["robots"][rel]["parity"] = frame_parity(rel, refs, lasts, raws, st_in, pool=pool)
            except Exception as e:
                rec["robots"][rel]["parity"] = {"error": repr(e)}


def reference_profile_script_record():
    """Sub-record `reference_profile_script`: /root/reference/example/profiling/profile_online_retargeting.py:39-77
    run UNMODIFIED -- its main(), its loop (:18-36), its prints -- with `dex_retargeting` aliased to the drop-in, in a process of
    its own (tests/reference_suite/run_profile_script.py; the script and its pickle are staged byte for byte by
    tests/reference_suite/stage.py, git-ignored).  7 robots x {vector, DexPilot} = 14 rows.  main() is executed twice in that
    process: the script creates the process's first HIP context inside its first row's timer (pass 1, as printed), pass 2 is the
    same script on a warm process."""
    import subprocess

    runner = os.path.join(REPO, "tests", "reference_suite", "run_profile_script.py")
    r = subprocess.run([sys.executable, runner, "--json", "--passes", "2"], capture_output=True, text=True, timeout=900, cwd=REPO)
    if r.returncode != 0:
        return {"error": (r.stderr or r.stdout)[-400:]}
    recs = [json.loads(l.split(" ", 1)[1]) for l in r.stdout.splitlines() if l.startswith("REFERENCE_PROFILE_SCRIPT ")]
    rows = recs[-1]["rows"]
    out = {"script": "example/profiling/profile_online_retargeting.py (unmodified; sha256 in tests/reference_suite/_ref/MANIFEST.json)",
           "frames_per_row": 621, "rows": rows, "unit": "fps (the script's own: 621 / summed perf_counter time around retarget())"}
    if len(recs) > 1:
        out["first_pass_rows"] = recs[0]["rows"]
    out["stdout"] = [l for l in r.stdout.splitlines() if not l.startswith("REFERENCE_PROFILE_SCRIPT ")][-15:]
    return out


def reference_profile_cpu_port(rec):
    """Checker-side leg of `reference_profile_script`: the CPU port (the oracle's plain-C closure + scipy's SLSQP at the reference's
    ftol, its own warm-start chain) on the SAME 14 rows, same fixture, one call per frame."""
    import bench_data
    from dex_retargeting_amd.constants import ROBOT_NAME_MAP, HandType, RetargetingType, get_default_config_path
    from oracle import cases, cport

    data = np.load(bench_data.HUMAN_FIXTURE)
    for row in rec["rows"]:
        rn = [k for k, v in ROBOT_NAME_MAP.items() if v == row["robot"]][0]
        rt = RetargetingType.vector if row["kind"] == "vector" else RetargetingType.dexpilot
        rel = os.path.relpath(str(get_default_config_path(rn, rt, HandType.right)), cases.CONFIG_DIR)
        prob = cases.problem_from_config(rel)
        cp = cport.CProblem(prob)
        refs = np.ascontiguousarray(cases.ref_from_keypoints(prob, data), dtype=np.float32)
        lo, hi = prob.joint_limits[:, 0], prob.joint_limits[:, 1]
        last = prob.joint_limits.mean(1).astype(np.float64)
        proj = np.zeros((1, prob.n_pair), bool) if prob.kind == "dexpilot" else None
        t = 0.0
        for i in range(len(refs)):
            tic = time.perf_counter()
            kw = {}
            if proj is not None:
                w, rv, proj = prob.dexpilot_preamble(refs[i][None], proj)
                kw = dict(weights=w, dexpilot_ref=rv)
            q_ref, _ = cport.solve_ref_as_configured_c(cp, refs[i][None], None, np.clip(last, lo, hi)[None].astype(np.float32), **kw)
            t += time.perf_counter() - tic
            last = q_ref[0].astype(np.float64)
        row["cpu_port_fps"] = len(refs) / t
    rec["cpu_port"] = {"kind": "port", "cores": 1, "note": "oracle/csrc closure + scipy SLSQP at the reference's ftol on the same 621-frame "
                                                            "loop per row (compiled stand-ins for pinocchio / nlopt, no torch overhead)"}


OFFLINE_ROBOTS = ["offline/allegro_hand_right.yml", "offline/shadow_hand_right.yml", "offline/leap_hand_right.yml",
                  "offline/ability_hand_right.yml"]


def offline_multi_robot_measure(torch, dev, tracks=8192, T=20):
    """Sub-record `offline_multi_robot`: the reference's offline viewer loop
    (/root/reference/example/position_retargeting/hand_robot_viewer.py:134-181) for K = 4 position configs with dummy free
    joints and `tracks` human hand tracks in lock-step: warm_start once, then per frame ONE fleet batch of K x tracks rows
    (MultiRobotSeqRetargeting).  Timed: T frames after the first (HIP events on the launch stream).  Returns (record,
    checker context); touches nothing under oracle/."""
    import bench_data
    from dex_retargeting_amd.constants import HandType
    from dex_retargeting_amd.multi_robot import MultiRobotSeqRetargeting
    from dex_retargeting_amd.retargeting_config import RetargetingConfig

    rets = [RetargetingConfig.load_from_file(os.path.join(bench_data.CONFIG_DIR, rel)).build() for rel in OFFLINE_ROBOTS]
    K = len(rets)
    multi = MultiRobotSeqRetargeting(rets, tracks, device=str(dev))
    kp, wrist_pos, wrist_quat = bench_data.world_tracks(tracks, T + 2)
    t_kp = torch.from_numpy(kp).to(dev)
    multi.warm_start(wrist_pos, wrist_quat, hand_type=HandType.right, is_mano_convention=True)
    multi.retarget(t_kp[0])  # frame 0: from the warm start (untimed: the solver's cold frame)
    stream = torch.cuda.current_stream()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    e0.record(stream)
    for t in range(1, T + 1):
        multi.retarget(t_kp[t])
    e1.record(stream)
    torch.cuda.synchronize()
    wall = time.perf_counter() - t0
    ms = float(e0.elapsed_time(e1))
    start = multi.last_qpos.clone()
    multi.retarget(t_kp[T + 1])  # one more frame, kept for the checker: start point, keypoints, answers
    q = multi.last_qpos.cpu().numpy()
    rec = {"robots": OFFLINE_ROBOTS, "tracks": tracks, "frames_timed": T, "rows_per_frame": K * tracks,
           "ms_per_frame": ms / T, "robot_frames_per_s": K * tracks * T / (ms * 1e-3), "wall_ms_per_frame": wall / T * 1e3,
           "unit": "robot-frames/s",
           "protocol": "hand_robot_viewer.py:134-181: warm_start(wrist_pos, wrist_quat, right, is_mano_convention=True) once per "
                       "robot, then per frame ref_value = joint[target_link_human_indices] -> retarget() for every robot; here "
                       "all K robots x all tracks of a frame are one dexr_retarget_multi_dev batch, last_qpos carried on the device"}
    return rec, dict(start=start.cpu().numpy(), q=q, kp=kp[T + 1], tracks=tracks)


def offline_multi_robot_check(rec, ctx, n_par=1024):
    """Checker: the last frame of the first n_par tracks of every robot against the float64 oracle from the same start."""
    from oracle import cases, jobs

    B = ctx["tracks"]
    n_par = min(n_par, B)
    rec["parity"] = {}
    with jobs.host_pool() as pool:
        for k, rel in enumerate(OFFLINE_ROBOTS):
            prob = cases.problem_from_config(rel)
            n = prob.n_opt
            lo, hi = prob.joint_limits[:, 0], prob.joint_limits[:, 1]
            rows = slice(k * B, k * B + n_par)
            last = np.clip(ctx["start"][rows, :n].astype(np.float64), lo, hi).astype(np.float32)
            ref = np.ascontiguousarray(cases.ref_from_keypoints(prob, ctx["kp"][:n_par]), dtype=np.float32)
            got = ctx["q"][rows, :n].astype(np.float64)
            o = jobs.pooled_oracle_solve(rel, ref, last, None, got, pool=pool)
            dq = np.abs(got - o["want"]).max(1)
            far = dq >= 1e-4
            rec["parity"][rel] = {"subset": n_par, "max_abs_dq_rad": float(dq.max()), "frac_within_1e-4": float((~far).mean()),
                                  "other_minimum": {"frames": int(far.sum()), "worse": int((o["F_gpu"][far] > o["F_want"][far] + 1e-9).sum())}}
    return rec


def general_kernel_measure(torch, dev, steps, warmup, B=65536):
    """Sub-record `general_kernel`: a model beyond the fixed-size tables -- an arm + Shadow hand URDF with 6 dummy free
    joints, position objective, 37 variables in one component, 21 reference rows (the reference accepts any URDF and any
    number of links, optimizer.py:18-52) -- on the general kernel (csrc/dexr_gen.hpp: one wavefront per frame, float64).
    Tracking workload like the headline's (human keypoints, warm start = the previous frame's solution).  Returns (record,
    checker context); touches nothing under oracle/."""
    import bench_data
    from dex_retargeting_amd import _lib
    from dex_retargeting_amd.retargeting_config import RetargetingConfig

    seq = RetargetingConfig.from_dict(bench_data.arm_hand_position_config()).build()
    model = seq.optimizer.device_model()
    assert model.kernel()[0] == _lib.KERNEL_GENERAL
    kp = bench_data.human_keypoints(B + 1)
    mid = np.repeat(seq.joint_limits.mean(1)[None], B, 0).astype(np.float32)
    last = model.retarget(np.ascontiguousarray(kp[:-1]), None, mid, keypoints=True)  # (untimed: the previous frame's answers)
    t_kp, t_last = torch.from_numpy(np.ascontiguousarray(kp[1:])).to(dev), torch.from_numpy(last).to(dev)
    out = torch.empty_like(t_last)
    it = torch.zeros(B, dtype=torch.int32, device=dev)
    stream = torch.cuda.current_stream()

    def go(diag=False):
        model.retarget_dev(B, t_kp.data_ptr(), 0, t_last.data_ptr(), 0, out.data_ptr(), iters_ptr=it.data_ptr() if diag else 0,
                           stream=stream.cuda_stream, keypoints=True)

    for _ in range(max(1, warmup)):
        go()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    e0.record(stream)
    for _ in range(steps):
        go()
    e1.record(stream)
    torch.cuda.synchronize()
    ms = float(e0.elapsed_time(e1)) / steps
    go(diag=True)
    torch.cuda.synchronize()
    iters = it.cpu().numpy()
    rec = {"workload": "arm + Shadow hand URDF (tests/urdf/arm_shadow_hand_right.urdf) + 6 dummy free joints, PositionOptimizer, "
                       "21 reference rows: 37 variables in one component (generic tables)", "batch": B, "dtype": "f64",
           "kernel": "dexr_gen_kernel (one wavefront per frame)", "value": B / (ms * 1e-3), "unit": "frames/s", "n_gpus": 1,
           "ms_per_step": ms, "steps": steps, "solver": {"iters_mean": float(iters.mean()), "iters_max": int(iters.max())}}
    n_opt = int(t_last.shape[1])
    bpf = algorithmic_bytes_per_frame(n_opt, False, 0, True)
    traffic, n_valu, note = pmc_counters("general_kernel", B)
    achieved = B * bpf / (ms * 1e-3) / 1e9
    rec["roofline"] = {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBPS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBPS,
                       "traffic": traffic, "traffic_over_algorithmic": None if not traffic else traffic / (B * bpf),
                       "valu_issue_frac": None if not n_valu else n_valu * 2.0 / (1024 * ms * 1e-3 * 2.4e9), "pmc_note": note,
                       "kernel_ms": ms, "algorithmic_bytes_per_frame": bpf}
    n = 128
    return rec, dict(kp=kp[1:n + 1], last=last[:n], q=out[:n].cpu().numpy())


def general_kernel_check(rec, ctx):
    """Checker: the first frames against the float64 oracle's minimiser from the same start (a frame that ends in another
    local minimum is counted, with the objective comparison, like everywhere else)."""
    import bench_data
    from oracle import cases, solvers
    from oracle.kin import OracleRobot
    from oracle.objectives import OracleProblem

    cfg = bench_data.arm_hand_position_config()
    prob = OracleProblem(OracleRobot(bench_data.ARM_HAND_URDF, add_dummy_free_joints=True), "position", None,
                         target_link_names=cfg["target_link_names"])
    prob.target_link_human_indices = np.arange(21)
    ref = np.ascontiguousarray(cases.ref_from_keypoints(prob, ctx["kp"]), dtype=np.float32)
    want = solvers.solve_lm_batched(prob, ref, None, ctx["last"], newton=True, max_iter=100)
    got, l64 = ctx["q"].astype(np.float64), ctx["last"].astype(np.float64)
    dq = np.abs(got - want).max(1)
    far = dq >= 1e-4
    Fg, Fw = prob.total(got, ref, None, l64), prob.total(want, ref, None, l64)
    rec["parity"] = {"subset": int(len(dq)), "max_abs_dq_rad_same_minimum": float(dq[~far].max()) if (~far).any() else None,
                     "frac_within_1e-4": float((~far).mean()),
                     "other_minimum": {"frames": int(far.sum()), "worse": int((Fg[far] > Fw[far] + 1e-9).sum())},
                     "oracle": "float64 projected LM/Newton on F (oracle/solvers.py), same start"}
    return rec


def parity_block(wl, batch, q_gpu, n_par, n_slsqp):
    """Checker (oracle) section: max |dq| against the float64 oracle minimiser of F on the first n_par frames of
    `batch`, and the distance to the reference-as-configured SLSQP answers on the first n_slsqp."""
    from oracle import cases, solvers

    prob = cases.problem_from_config(wl.rel)
    assert (prob.n_opt, prob.n_ref) == (wl.n_opt, wl.n_ref)
    ref = batch["host_in"][:n_par]
    if batch["kind"] == "kp":
        ref = cases.ref_from_keypoints(prob, ref).astype(np.float32)
    last = batch["host_last"][:n_par]

    def kw_for(sl):
        if not wl.dexpilot:
            return {}
        st = batch["host_state"][sl] if batch["host_state"] is not None else np.zeros(len(ref[sl]), np.uint32)
        proj = ((st[:, None] >> np.arange(prob.n_pair, dtype=np.uint32)) & 1).astype(bool)
        w, rv, _ = prob.dexpilot_preamble(ref[sl], proj)
        return dict(weights=w, dexpilot_ref=rv)

    kw = kw_for(slice(0, n_par))
    got = q_gpu[:n_par].astype(np.float64)
    last64 = last.astype(np.float64)
    if n_par >= 1024:  # thousands of frames: the oracle phase is fanned over the host cores (spawned processes)
        from oracle import jobs

        st_in = None if not wl.dexpilot else (batch["host_state"][:n_par] if batch["host_state"] is not None else np.zeros(n_par, np.uint32))
        o = jobs.pooled_oracle_solve(wl.rel, np.ascontiguousarray(ref), np.ascontiguousarray(last), st_in, got)
        want, F_got, F_want = o["want"], o["F_gpu"], o["F_want"]
    else:
        want = solvers.solve_lm_batched(prob, ref, None, last, newton=True, max_iter=100, **kw)
        F_got = prob.total(got, ref, None, last64, **kw)
        F_want = prob.total(want, ref, None, last64, **kw)
    dq = np.abs(got - want).max(1)
    other = dq >= 1e-4
    out = {"subset": n_par, "max_abs_dq_rad": float(dq.max()), "p99_abs_dq_rad": float(np.percentile(dq, 99)),
           "frac_within_1e-4": float((dq < 1e-4).mean()),
           "other_minimum": {"frames": int(other.sum()),
                             "gpu_objective_lower_or_equal": int((F_got[other] <= F_want[other] + 1e-9).sum()),
                             "worse": int((F_got[other] > F_want[other] + 1e-9).sum())},
           "max_abs_dq_rad_same_minimum": float(dq[~other].max()) if (~other).any() else None,
           "oracle": "float64 projected LM/Newton on F, positive-definite damped models only (oracle/solvers.py)"}
    if n_slsqp:
        sl = slice(0, n_slsqp)
        q_ref, _ = solvers.solve_ref_as_configured(prob, ref[sl], None, last[sl], **kw_for(sl))
        q_ref = q_ref.astype(np.float64)
        kws = {k: v[sl] for k, v in kw.items()}
        F_ref = prob.total(q_ref, ref[sl], None, last64[sl], **kws)
        d = np.abs(got[sl] - q_ref).max(1)
        out["vs_reference_as_configured"] = {
            "subset": n_slsqp, "median_abs_dq_rad": float(np.median(d)), "p99_abs_dq_rad": float(np.percentile(d, 99)),
            "max_abs_dq_rad": float(d.max()), "frac_F_gpu_le_F_ref": float((F_got[sl] <= F_ref + 1e-12).mean()),
            "median_F_ref_minus_F_gpu": float(np.median(F_ref - F_got[sl])),
            "reference": f"oracle restatement of the reference objective (value without, gradient with the regulariser) + "
                         f"scipy SLSQP ftol {prob.ftol:g} standing in for nlopt LD_SLSQP ftol_abs (optimizer.py:96-99)"}
    return out, prob, ref, last, kw_for


def reference_stack_record(rel, n_frames=621):
    """The TRUE reference stack (pinocchio + nlopt + the dex_retargeting package itself), if this box happens to have it:
    the reference's own profiling loop (example/profiling/profile_online_retargeting.py:18-36 -- one
    SeqRetargeting.retarget(ref_value) per fixture frame, perf_counter around the call) on its own classes, with this repo's
    URDFs and YAMLs as inputs.  Labelled apart from cpu_baseline.  On the
    images seen so far none of the three imports: the record then says which are missing and nothing is timed."""
    import importlib

    missing = []
    for mod in ("pinocchio", "nlopt", "dex_retargeting"):
        try:
            importlib.import_module(mod)
        except Exception:
            missing.append(mod)
    if missing:
        return {"available": False, "missing": missing,
                "note": "the reference's own stack is not installed here: cpu_baseline (kind 'port') stands in for it"}
    import bench_data
    from dex_retargeting.retargeting_config import RetargetingConfig as RefConfig  # the reference package itself
    from dex_retargeting_amd.constants import DEFAULT_URDF_DIR

    RefConfig.set_default_urdf_dir(str(DEFAULT_URDF_DIR))
    retargeting = RefConfig.load_from_file(os.path.join(bench_data.CONFIG_DIR, rel)).build()
    data = np.load(bench_data.HUMAN_FIXTURE)[:n_frames].astype(np.float64)
    indices = retargeting.optimizer.target_link_human_indices
    position = retargeting.optimizer.retargeting_type == "POSITION"
    dt = []
    for joint_pos in data:
        ref_value = joint_pos[indices, :] if position else joint_pos[indices[1, :], :] - joint_pos[indices[0, :], :]
        tic = time.perf_counter()
        retargeting.retarget(ref_value)
        dt.append(time.perf_counter() - tic)
    dt = np.array(dt)
    return {"available": True, "kind": "reference", "cores": 1, "frames": len(dt), "value": len(dt) / dt.sum(), "unit": "frames/s",
            "mean_ms": float(dt.mean() * 1e3), "p99_ms": float(np.percentile(dt, 99) * 1e3),
            "note": "dex_retargeting's own SeqRetargeting on pinocchio + nlopt, this repo's URDF / YAML files as inputs"}


def cgroup_cpu_max():
    """The container's CPU quota as the kernel reports it ("max 100000" = none; "<quota> <period>" = quota / period CPUs)."""
    for path in ("/sys/fs/cgroup/cpu.max", "/sys/fs/cgroup/cpu/cpu.cfs_quota_us"):
        try:
            return open(path).read().strip()
        except OSError:
            continue
    return None


def usable_cpus():
    """CPUs this process can actually use together: min(affinity mask, cgroup quota / period) -- BASELINE.md section 3
    asks for the core count usable, not the host's."""
    avail = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    cm = cgroup_cpu_max()
    try:
        parts = (cm or "").split()
        if len(parts) == 2 and parts[0] != "max":
            avail = max(1, min(avail, int(int(parts[0]) // int(parts[1]))))
        elif len(parts) == 1 and int(parts[0]) > 0:  # cgroup v1 cfs_quota_us, period 100 ms by default
            avail = max(1, min(avail, int(parts[0]) // 100000))
    except ValueError:
        pass
    return avail


def job_env(args):
    """(rank, local_rank, world, launched): read from the environment torch.distributed.run sets."""
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    return rank, local_rank, world, os.environ.get("RANK") is not None


class Watchdog:
    """Deadline for the optional N > 1 measurements.  A collective that misbehaves across ranks (a graph capture RCCL does
    not support, a rank that died) HANGS rather than raises; the headline has been measured by then.  Every rank arms the
    same deadlines at the same points; when one passes, rank 0 prints the line it already has (plus what timed out) and
    every rank leaves with os._exit, so the launcher sees a finished job instead of its own timeout."""

    def __init__(self, rank, seconds):
        import threading

        self.rank, self.seconds, self.line, self.stage, self.deadline = rank, seconds, None, None, None
        self.done = {}
        self._lock = threading.Lock()
        t = threading.Thread(target=self._run, daemon=True)
        t.start()

    def arm(self, stage):
        with self._lock:
            self.stage, self.deadline = stage, time.time() + self.seconds

    def disarm(self):
        with self._lock:
            self.stage, self.deadline = None, None

    def _run(self):
        while True:
            time.sleep(0.25)
            with self._lock:
                late = self.deadline is not None and time.time() > self.deadline
                stage = self.stage
            if late:
                if self.rank == 0 and self.line is not None:
                    line = dict(self.line)
                    line["multi_gpu"] = dict(self.done, watchdog=f"'{stage}' did not finish within {self.seconds:g} s; the "
                                                                 f"figures measured before it are reported, the job was ended")
                    import bench_line

                    bench_line.emit(line)
                os._exit(0)


def rccl_env():
    """What the RCCL tuner was told (nothing = its own choice per message size; SURVEY.md section 8e asks for it to be
    stated).  bench.py --nccl-algo / --nccl-proto set these before the communicator exists."""
    keys = ("NCCL_ALGO", "NCCL_PROTO", "NCCL_MIN_NCHANNELS", "NCCL_MAX_NCHANNELS", "NCCL_P2P_LEVEL", "RCCL_MSCCL_ENABLE",
            "RCCL_MSCCLPP_ENABLE", "HSA_ENABLE_IPC_MODE_LEGACY")
    return {k: os.environ.get(k, "unset (RCCL default)") for k in keys}


def gather_records(timed_fn, comm, B, n_cols, dev, steps, warmup, world, graph_fn=None, rank=0, on_headline=None,
                   strong_fn=None):
    """The N > 1 figures (also taken with ONE rank when launched by torch.distributed.run, so that the RCCL side runs on a
    1-GPU box): dexr_allgather of the (B, n_cols) f32 result rows.  Returns
    (elapsed, kernel_ms, record dict).

    no_gather      = the steps with the usual barrier bracket and NO collective ("with and without the
                     all-gather"); its step time picks k (distributed.steps_per_gather_for: the gathers must keep up with
                     the solves at <= 60 % of the xGMI ingest);
    headline       = ONE collective PER STEP on a second HIP stream, ordered after the solve by an event: the next step's solve
                     overlaps it; every gather has completed when the timed region ends (north_star's per-step reassembly);
    other mode     = one collective per k steps, k from the wire-time policy (4 when that is 1): results reach the other ranks up
                     to k - 1 steps late -- reported beside the headline, never as it;
    on solve stream= one collective per step on the solve stream itself (strictly serial);
    strong scaling = the metric's 65 536 frames over the whole node (B / N per GPU) with the per-step gather;
    graph replay   = [solve -> all-gather] x 4 captured into one HIP graph.
    Everything after the headline runs under a watchdog."""
    from dex_retargeting_amd.distributed import NativeGather, steps_per_gather_for

    wd = Watchdog(rank, float(os.environ.get("DEXR_BENCH_WATCHDOG_S", "120")))
    depth = min(8, steps + warmup + 1)
    shard_mb = B * n_cols * 4 / 1e6

    def fig(e, note, frames=None):
        return {"value": (world * B if frames is None else frames) * steps / e, "unit": "frames/s", "ms_per_step": e / steps * 1e3, "note": note}

    # (1) the shards alone: no collective.  Its step time (maximum over the ranks: every rank derives the same k) decides how
    # many steps share one all-gather in the headline
    e0, kernel_ms0 = timed_fn(None)
    k_policy = steps_per_gather_for(e0 / steps * 1e3, B * n_cols * 4, world)
    while steps % k_policy:
        k_policy -= 1  # whole groups inside the timed region
    # Round 6: the HEADLINE reassembles the qpos tensor EVERY step -- north_star's contract ("an RCCL
    # all-gather ... to reassemble the qpos tensor"); one collective per k steps delivers the other ranks' rows up to k - 1 steps
    # late, which is another contract: it is measured and reported beside the headline (`gather_every_k_steps`), never as it.
    k = 1
    # (2) headline: one all-gather per k steps on the second stream (k = 1 whenever the solve hides a per-step gather: every
    # workload but the 47 us Allegro step at N >= 4; with one rank always 1).  Under the watchdog as well: should the
    # collective never complete across the ranks, the job ends with a line that carries the no-gather figure and says so
    # -- labelled as what it is -- instead of the launcher's timeout and no line at all.
    rec = wd.done
    if on_headline is not None:
        fallback = on_headline(e0, kernel_ms0)
        fallback["config"] = dict(fallback.get("config", {}), collective="NONE -- the all-gather did not complete (see multi_gpu.watchdog): "
                                  "this line is the no-gather figure, not the metric's reassembled result")
        wd.line = fallback
    wd.arm("headline_all_gather")
    elapsed, kernel_ms = timed_fn(NativeGather(comm, B, n_cols, dev, depth=depth if k == 1 else 2, overlap=True, steps_per_gather=k))
    wd.disarm()
    rec.update({
        "collective": f"dexr_allgather (C-ABI of libdexr.so -> RCCL ncclAllGather, bound with dlopen): ONE per {k} step(s) "
                      f"({k} x {shard_mb:.2f} MB per rank), enqueued on a second HIP stream behind an event recorded after the "
                      f"solve; every gather completes inside the timed region",
        "steps_per_gather": k,
        "steps_per_gather_of_the_grouped_mode": k_policy if k_policy > 1 else 4,
        "steps_per_gather_policy": "(grouped mode only; the headline gathers every step) smallest k for which a gather at 60 % of 537 GB/s xGMI ingest + 30 us of collective overhead "
                                   "fits into k solve steps (distributed.steps_per_gather_for); the largest divisor of the step "
                                   "count <= 16 when one step's wire time already exceeds the step",
        "rccl_world_size": world, "rccl_version": comm.rccl_version(), "rccl_env": rccl_env(),
        "no_gather": fig(e0, "same steps and barrier bracket with NO collective: what the shards alone sustain"),
        "xgmi": {"shard_MB_per_rank_per_step": shard_mb, "received_MB_per_gpu_per_step": shard_mb * (world - 1),
                 "note": "every GPU receives (N-1) shards per step over its 7 xGMI links (~76.8 GB/s per link and "
                         "direction, 537 GB/s aggregate ingest at best): the all-gather lower bound per step is "
                         f"{shard_mb * (world - 1) / 537.0 * 1e3:.1f} us at N={world} if all links carry it, "
                         f"{shard_mb * (world - 1) / 76.8 * 1e3:.1f} us on a single ring direction"}})
    if on_headline is not None:
        wd.line = on_headline(elapsed, kernel_ms)
    k2 = k_policy if k_policy > 1 else 4  # the grouped mode, for comparison (the wire-time policy's k; 4 when that is 1)
    while steps % k2:
        k2 -= 1
    wd.arm("gather_other_mode")
    ek, _ = timed_fn(NativeGather(comm, B, n_cols, dev, depth=depth if k2 == 1 else 2, overlap=True, steps_per_gather=k2))
    rec["gather_every_%d_step%s" % (k2, "" if k2 == 1 else "s")] = fig(
        ek, f"ONE all-gather per {k2} steps on the second stream: same bytes, 1 / {k2} of the collectives, the other ranks' rows arrive up "
            f"to {k2 - 1} steps late (not the metric's contract)")
    wd.arm("gather_on_solve_stream")
    e2, _ = timed_fn(NativeGather(comm, B, n_cols, dev, depth=depth, overlap=False))
    rec["gather_on_solve_stream"] = fig(e2, "one all-gather per step enqueued on the solve stream itself (serial)")
    if strong_fn is not None and world > 1:
        wd.arm("strong_scaling")
        rec["strong_scaling"] = strong_fn()
    if graph_fn is not None:
        wd.arm("graph_replay")
        try:
            rec["graph_replay"] = graph_fn()
        except Exception as e:  # capture support varies with the RCCL build: never lose the line to it
            rec["graph_replay"] = {"error": repr(e)}
    wd.disarm()
    return elapsed, kernel_ms, dict(rec)


def run_single(args):
    import torch

    rank, local_rank, world, launched = job_env(args)
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    comm = None
    # launched by torch.distributed.run (RANK set): take the N > 1 path even with one rank, so that the RCCL side of
    # this script can be exercised on a 1-GPU box too
    if world > 1 or (launched and os.environ.get("DEXR_BENCH_DIST", "1") != "0"):
        from dex_retargeting_amd.distributed import native_comm

        comm = native_comm(rank, world)

    from dex_retargeting_amd import _lib

    B = args.batch
    import bench_line

    if args.probe == "general_kernel":  # (tools/profile_round.sh general_kernel: the sub-record's loop alone, for rocprofv3)
        rec, _ = general_kernel_measure(torch, dev, args.steps, args.warmup)
        probe = {"probe": "general_kernel", "config": {"batch_per_gpu": rec["batch"]}, "roofline": {"kernel_ms": rec["ms_per_step"]},
                 "value": rec["value"], "unit": "frames/s"}
        print(bench_line.dumps(probe))  # (a few bytes for tools/prof_summary.py, not a contract line)
        return
    wl = Workload(args.workload, rank, B, dev, torch)
    if args.probe in ("f64", "cold_start"):
        batches = wl.tracking if args.probe == "f64" else wl.stage_cold()
        opts = _lib.default_options(precision=1) if args.probe == "f64" else None
        e, k = wl.timed(batches, args.steps, args.warmup, opts=opts)
        probe = {"probe": args.probe, "config": {"batch_per_gpu": B}, "roofline": {"kernel_ms": k}, "value": B * args.steps / e,
                 "unit": "frames/s"}
        print(bench_line.dumps(probe))  # (a few bytes for tools/prof_summary.py, not a contract line)
        return
    diag = wl.diagnostics(wl.tracking)

    # ---- headline: float32 tracking -----------------------------------------------------------------------------
    def contract_line(elapsed, kernel_ms, coll=None):
        frames = world * B * args.steps
        return {
            "metric": json.load(open(os.path.join(REPO, "BASELINE.json")))["metric"],
            "value": frames / elapsed, "unit": "frames/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": elapsed / args.steps * 1e3, "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": f"{wl.title}, {B} frames/GPU, human-keypoint refs (fixture frame b mod 621 + 2 mm noise), "
                                   f"warm start = previous frame's solution; {N_BATCHES} staged batches rotated over the steps",
                       "config_file": wl.rel, "batch_per_gpu": B, "n_opt": wl.n_opt, "n_ref": wl.n_ref,
                       "collective": "none" if comm is None else "dexr_allgather (RCCL ncclAllGather) on a second stream, one per step",
                       "rccl_world_size": None if comm is None else world},
            "solver": dict(diag, tol_rad=2e-6, newton=1),
            "roofline": dict(wl.roofline(kernel_ms, diag["iters_mean"], world=world),
                             traffic_note="bytes per launch from the rocprofv3 --pmc passes in profiles/ (same command, same "
                                          "batch, same kernel sources); null when no matching profile is committed",
                             note="path is FP32 VALU/latency bound (n_dof <= 24 per lane, no dense contraction); the HBM "
                                  "fraction is reported as north_star asks, see DESIGN.md section 4"),
        }

    def strong_scaling():
        """The metric read as STRONG scaling: its 65 536 frames over the whole node, B / N frames per GPU, per-step gather
        on the second stream (each GPU then holds a fraction of a wave per SIMD: the step sits on the launch-latency floor
        of profiles/r03_small_latency_tip.txt, and the all-gather moves 1 / N as much)."""
        from dex_retargeting_amd.distributed import NativeGather

        Bs = max(64, B // world)
        ws = Workload(args.workload, rank, Bs, dev, torch)
        es, ks = ws.timed(ws.tracking, args.steps, args.warmup, comm=comm,
                          pipe=NativeGather(comm, Bs, ws.n_opt, dev, depth=min(8, args.steps + args.warmup + 1), overlap=True))
        e0s, _ = ws.timed(ws.tracking, args.steps, args.warmup, comm=comm)
        return {"total_frames_per_step": world * Bs, "frames_per_gpu": Bs, "value": world * Bs * args.steps / es,
                "unit": "frames/s", "ms_per_step": es / args.steps * 1e3, "solve_kernel_ms": ks,
                "no_gather": {"value": world * Bs * args.steps / e0s, "ms_per_step": e0s / args.steps * 1e3},
                "scaling": "strong", "note": f"{world * Bs} frames per step over {world} GPUs"}

    coll = None
    if comm is not None:
        elapsed, kernel_ms, coll = gather_records(
            lambda pipe: wl.timed(wl.tracking, args.steps, args.warmup, pipe=pipe, comm=comm),
            comm, B, wl.n_opt, dev, args.steps, args.warmup, world, rank=rank,
            graph_fn=lambda: wl.timed_graph(wl.tracking, args.steps, comm),
            on_headline=(lambda e, k: contract_line(e, k)) if rank == 0 else None,
            strong_fn=strong_scaling)
    else:
        elapsed, kernel_ms = wl.timed(wl.tracking, args.steps, args.warmup)
    # per-step answers of the last timed step's batch for the parity check
    last_batch = wl.tracking[(args.steps - 1) % N_BATCHES]
    wl.launch(last_batch, wl.t_q)
    torch.cuda.synchronize()
    q_head = wl.t_q.cpu().numpy()

    # ---- sub-records (rank 0's GPU only; untouched by the collective) --------------------------------------------
    sub = {}
    online_ctx = None
    if world > 1:
        args.headline_only = True  # scaling runs: the headline (+ the pipelined-gather figure) only
    if rank == 0 and not args.headline_only:
        o64 = _lib.default_options(precision=1)
        d64 = wl.diagnostics(wl.tracking, opts=o64)
        e64, k64 = wl.timed(wl.tracking, args.steps, args.warmup, opts=o64)
        wl.launch(last_batch, wl.t_q, opts=o64)
        torch.cuda.synchronize()
        q64 = wl.t_q.cpu().numpy()
        sub["f64"] = {"dtype": "f64", "value": B * args.steps / e64, "unit": "frames/s", "n_gpus": 1,
                      "ms_per_step": e64 / args.steps * 1e3, "solver": d64,
                      "roofline": wl.roofline(k64, d64["iters_mean"], precision="f64"),
                      "max_abs_dq_vs_f32_rad": float(np.abs(q64.astype(np.float64) - q_head).max()),
                      "note": "same workload and config, dexr_solve_options.precision = 1: float64 arithmetic throughout"}
        cold = wl.stage_cold()
        dc = wl.diagnostics(cold)
        ec, kc = wl.timed(cold, args.steps, args.warmup)
        wl.launch(cold[0], wl.t_q)
        torch.cuda.synchronize()
        q_cold = wl.t_q.cpu().numpy()
        sub["cold_start"] = {"dtype": "f32", "value": B * args.steps / ec, "unit": "frames/s", "n_gpus": 1,
                             "ms_per_step": ec / args.steps * 1e3, "solver": dc,
                             "roofline": wl.roofline(kc, dc["iters_mean"], batch_kind="ref"),
                             "workload": "reachable targets (the robot's own FK at q* ~ U(limits)), start = q* + 0.5 rad "
                                         "N(0,1) clipped to the limits: tests/test_optimizer.py:27-81 of the reference"}
    if rank == 0 and not args.headline_only:
        try:
            sub["sustained"] = wl.sustained(wl.tracking, seconds=args.sustained_seconds)
            sub["sustained"]["vs_ms_per_step"] = sub["sustained"]["ms_per_step"] / (elapsed / args.steps * 1e3)
        except Exception as e:
            sub["sustained"] = {"error": repr(e)}
    if rank == 0 and not args.headline_only:
        try:
            sub["two_streams"] = wl.timed_two_streams(wl.tracking, args.steps, args.warmup)
        except Exception as e:
            sub["two_streams"] = {"error": repr(e)}
    if rank == 0 and not args.headline_only:
        try:
            sub["sequence_mode"] = sequence_record(wl, torch)
        except Exception as e:  # never lose the headline line to a sub-record
            sub["sequence_mode"] = {"error": repr(e)}
    if rank == 0 and not args.headline_only and args.workload == "allegro_vector":
        torch.cuda.synchronize()
        try:
            sub["online_teleop"], online_ctx = online_record()
        except Exception as e:
            sub["online_teleop"] = {"error": repr(e)}
    ref_script = None
    if rank == 0 and not args.headline_only and args.workload == "allegro_vector":
        torch.cuda.synchronize()
        try:
            ref_script = reference_profile_script_record()
        except Exception as e:
            ref_script = {"error": repr(e)}
        sub["reference_profile_script"] = ref_script
    also = {}
    if rank == 0 and not args.headline_only and args.workload == "allegro_vector":
        for name in ("shadow_dexpilot", "leap_position"):
            w2 = Workload(name, rank, B, dev, torch)
            d2 = w2.diagnostics(w2.tracking)
            e2, k2 = w2.timed(w2.tracking, args.steps, args.warmup)
            try:
                ts2 = w2.timed_two_streams(w2.tracking, args.steps, args.warmup)
            except Exception as e:
                ts2 = {"error": repr(e)}
            b2 = w2.tracking[(args.steps - 1) % N_BATCHES]
            w2.launch(b2, w2.t_q)
            torch.cuda.synchronize()
            q2 = w2.t_q.cpu().numpy()
            # the same config in the reference's own arithmetic (optimizer.py:263-300, 524-573: float64 throughout):
            # dexr_solve_options.precision = 1 -> the register kernel dexr_kernel<24, double> (one lane per frame; the only
            # all-float64 implementation of these models) -- a few steps, whatever it costs
            try:
                o64 = _lib.default_options(precision=1)
                s64, w64 = max(2, min(args.steps, 4)), 1
                d64b = w2.diagnostics(w2.tracking[:1], opts=o64)
                e64b, k64b = w2.timed(w2.tracking, s64, w64, opts=o64)
                w2.launch(b2, w2.t_q, opts=o64)
                torch.cuda.synchronize()
                f64rec = {"dtype": "f64", "value": B * s64 / e64b, "unit": "frames/s", "ms_per_step": e64b / s64 * 1e3, "steps": s64,
                          "kernel": f"dexr_kernel<{w2.model.kernel()[1]}, double> (register kernel, float64 arithmetic throughout)",
                          "solver": d64b, "max_abs_dq_vs_default_rad": float(np.abs(w2.t_q.cpu().numpy().astype(np.float64) - q2).max()),
                          "p999_abs_dq_vs_default_rad": float(np.percentile(np.abs(w2.t_q.cpu().numpy().astype(np.float64) - q2).max(1), 99.9))}
            except Exception as e:
                f64rec = {"error": repr(e)}
            # ... and on the GENERAL kernel (csrc/dexr_gen.hpp: one wavefront per frame, float64 kinematics, Hessian, factorisation) through
            # the generic table format of the SAME config (Optimizer.use_generic_tables): the float64 implementation that does not
            # spill -- this is the like-for-like figure of the record; the register kernel's stays beside it
            try:
                from dex_retargeting_amd.retargeting_config import RetargetingConfig as _RC
                import bench_data as _bd

                sg = _RC.load_from_file(os.path.join(_bd.CONFIG_DIR, w2.rel)).build()
                sg.optimizer.use_generic_tables = True
                mg = sg.optimizer.device_model()
                assert mg.kernel()[0] == _lib.KERNEL_GENERAL
                t_qg, t_itg = torch.empty_like(w2.t_q), torch.zeros(B, dtype=torch.int32, device=dev)

                def go_g(b, diag=False):
                    if w2.dexpilot:
                        w2.t_state.copy_(b["t_state0"])
                    mg.retarget_dev(B, b["t_in"].data_ptr(), 0, b["t_last"].data_ptr(), w2.t_state.data_ptr() if w2.dexpilot else 0,
                                    t_qg.data_ptr(), iters_ptr=t_itg.data_ptr() if diag else 0, stream=w2.stream.cuda_stream, keypoints=True)

                go_g(w2.tracking[0])
                torch.cuda.synchronize()
                sgn = max(2, min(args.steps, 5))
                eg0, eg1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                tg0 = time.perf_counter()
                eg0.record(w2.stream)
                for k_ in range(sgn):
                    go_g(w2.tracking[k_ % N_BATCHES])
                eg1.record(w2.stream)
                torch.cuda.synchronize()
                elg = time.perf_counter() - tg0
                go_g(b2, diag=True)
                torch.cuda.synchronize()
                qg = t_qg.cpu().numpy().astype(np.float64)
                itg = t_itg.cpu().numpy()
                dqg = np.abs(qg - q2).max(1)
                f64rec = {"dtype": "f64", "value": B * sgn / elg, "unit": "frames/s", "ms_per_step": elg / sgn * 1e3, "steps": sgn,
                          "kernel_ms": float(eg0.elapsed_time(eg1)) / sgn,
                          "kernel": "dexr_gen_kernel (one wavefront per frame; float64 kinematics, Hessian and factorisation) on the generic "
                                    "tables of the same config",
                          "solver": {"iters_mean": float(itg.mean()), "iters_max": int(itg.max())},
                          "frac_within_1e-4_of_default": float((dqg < 1e-4).mean()), "p99_abs_dq_vs_default_rad": float(np.percentile(dqg, 99)),
                          "register_kernel": f64rec}
            except Exception as e:
                f64rec = dict(f64rec, general_kernel_error=repr(e))
            also[name] = (w2, b2, q2,
                          {"config_file": w2.rel, "workload": w2.title, "dtype": WIDE_DTYPE if w2.model.kernel()[0] == _lib.KERNEL_WIDE else "f32",
                           "value": B * args.steps / e2,
                           "unit": "frames/s", "n_gpus": 1, "ms_per_step": e2 / args.steps * 1e3, "solver": d2,
                           "roofline": w2.roofline(k2, d2["iters_mean"]), "two_streams": ts2, "f64": f64rec})

    # small batches (round 5): up to 2 048 frames a model on the sixteen-lane kernel runs one frame per wave with a ladder of
    # damping values (dexr_tuning.sprint_max_batch / sprint_ladder); the same batches four frames per wave beside it
    small_ctx = {}
    if rank == 0 and not args.headline_only and args.workload == "allegro_vector":
        try:
            sb = {"frames": 700, "note": "700 tracking frames per launch: default policy (one frame per wave + ladder) vs four frames per wave"}
            for name in ("shadow_dexpilot", "leap_position"):
                w3 = Workload(name, rank, 700, dev, torch)
                d3 = w3.diagnostics(w3.tracking)
                e3, k3 = w3.timed(w3.tracking, args.steps, args.warmup)
                w3.model.tune(sprint_max_batch=0)
                d4 = w3.diagnostics(w3.tracking)
                e4, k4 = w3.timed(w3.tracking, args.steps, args.warmup)
                w3.model.tune(sprint_max_batch=-1)
                b3 = w3.tracking[0]
                w3.launch(b3, w3.t_q)
                torch.cuda.synchronize()
                small_ctx[name] = (w3, b3, w3.t_q.cpu().numpy(), None if b3["host_state"] is None else b3["host_state"].astype(np.uint32))
                sb[name] = {"ms_per_step": e3 / args.steps * 1e3, "value": 700 * args.steps / e3, "iters_mean": d3["iters_mean"], "iters_max": d3["iters_max"],
                            "four_per_wave": {"ms_per_step": e4 / args.steps * 1e3, "iters_mean": d4["iters_mean"], "iters_max": d4["iters_max"]}}
            sub["small_batch"] = sb
        except Exception as e:
            sub["small_batch"] = {"error": repr(e)}

    fleet_m, offline_m, gen_m = None, None, None
    if rank == 0 and not args.headline_only and args.workload == "allegro_vector":
        try:  # BASELINE configs[4], the per-GPU slice: 1 048 576 / 8 frames of four robots in one batch
            import bench_fleet

            fleet_m = bench_fleet.fleet_measure(args, batch=131072, standalone=False)
        except Exception as e:
            fleet_m = repr(e)
        try:
            offline_m = offline_multi_robot_measure(torch, dev)
        except Exception as e:
            offline_m = repr(e)
        try:
            gen_m = general_kernel_measure(torch, dev, min(args.steps, 5), min(args.warmup, 2))
        except Exception as e:
            gen_m = repr(e)

    if rank != 0:
        comm.close()
        return

    out = contract_line(elapsed, kernel_ms)
    if coll is not None:
        out["multi_gpu"] = coll
    out.update(sub)

    # ---- checker sections (oracle = checker only; nothing above this line touches oracle/) -------------------------
    n_par = 4096  # 4 096-item subset (the oracle phase runs in a host process pool)
    par, prob, ref_now, last_now, kw_for = parity_block(wl, last_batch, q_head, min(n_par, B), 0 if args.no_cpu_baseline else min(128, B))
    out["parity"] = par
    if "cold_start" in sub:
        from oracle import solvers

        n_c = min(512, B)
        want_c = solvers.solve_lm_batched(prob, wl.cold[0]["host_in"][:n_c], None, wl.cold[0]["host_last"][:n_c], newton=True, max_iter=100)
        dqc = np.abs(q_cold[:n_c].astype(np.float64) - want_c).max(1)
        l64 = wl.cold[0]["host_last"][:n_c].astype(np.float64)
        Fg = prob.total(q_cold[:n_c].astype(np.float64), wl.cold[0]["host_in"][:n_c], None, l64)
        Fw = prob.total(want_c, wl.cold[0]["host_in"][:n_c], None, l64)
        far = dqc >= 1e-4
        sub["cold_start"]["parity"] = {"subset": n_c, "frac_within_1e-4": float((~far).mean()),
                                       "other_minimum": {"frames": int(far.sum()),
                                                         "gpu_objective_lower_or_equal": int((Fg[far] <= Fw[far] + 1e-9).sum())},
                                       "note": "far starts are multi-modal: frames that end in another minimum than the "
                                               "oracle's are counted, with the objective comparison"}
    for name, (w2, b2, q2, rec) in also.items():
        rec["parity"] = parity_block(w2, b2, q2, min(4096, B), 0 if args.no_cpu_baseline else min(64, B))[0]
        out.setdefault("also", {})[name] = rec

    if online_ctx is not None:
        try:
            online_parity(out["online_teleop"], online_ctx)
        except Exception as e:
            out["online_teleop"]["parity_error"] = repr(e)
    for name, (w3, b3, q3, st3) in small_ctx.items():
        try:
            from oracle import cases as _cases

            prob3 = _cases.problem_from_config(w3.rel)
            out["small_batch"][name]["parity"] = frame_parity(
                w3.rel, _cases.ref_from_keypoints(prob3, b3["host_in"]), b3["host_last"], q3, st3)
        except Exception as e:
            out["small_batch"][name]["parity"] = {"error": repr(e)}
    if online_ctx is not None and not args.no_cpu_baseline:
        try:
            online_cpu_port(out["online_teleop"], online_ctx)
        except Exception as e:
            out["online_teleop"]["cpu_port_error"] = repr(e)
    if ref_script is not None and "rows" in ref_script and not args.no_cpu_baseline:
        try:
            reference_profile_cpu_port(ref_script)
        except Exception as e:
            ref_script["cpu_port_error"] = repr(e)
    if fleet_m is not None:
        import bench_fleet

        try:
            out.setdefault("also", {})["mixed_fleet"] = fleet_m if isinstance(fleet_m, str) else bench_fleet.fleet_check(*fleet_m, args)
        except Exception as e:
            out.setdefault("also", {})["mixed_fleet"] = {"error": repr(e)}
    if offline_m is not None:
        try:
            out["offline_multi_robot"] = {"error": offline_m} if isinstance(offline_m, str) else offline_multi_robot_check(*offline_m)
        except Exception as e:
            out["offline_multi_robot"] = {"error": repr(e)}
    if gen_m is not None:
        try:
            out["general_kernel"] = {"error": gen_m} if isinstance(gen_m, str) else general_kernel_check(*gen_m)
        except Exception as e:
            out["general_kernel"] = {"error": repr(e)}

    # ---- CPU baseline: the reference path as configured, on the host cores (oracle = checker code only) -------------
    if world == 1 and not args.no_cpu_baseline:
        from oracle import cport, solvers

        cp = cport.CProblem(prob)

        def timed_cpu(solve, budget_s, chunk):
            done, t_cpu, evals = 0, 0.0, 0
            while done < min(args.cpu_sample, B) and t_cpu < budget_s:  # bounded sample of the same workload
                sl = slice(done, min(done + chunk, len(ref_now)))
                if sl.start >= sl.stop:
                    break
                t1 = time.perf_counter()
                _, ev = solve(ref_now[sl], None, last_now[sl], **kw_for(sl))
                t_cpu += time.perf_counter() - t1
                evals += int(ev.sum())
                done = sl.stop
            return done, t_cpu, evals

        done, t_cpu, evals = timed_cpu(lambda *a, **k: cport.solve_ref_as_configured_c(cp, *a, **k), 12.0, 200)
        # cost of ONE closure evaluation in the compiled port (FK + Jacobians + loss + chain rule), timed alone
        tgt = cp.target(ref_now[0])
        x0 = last_now[0].astype(np.float64)
        t1 = time.perf_counter()
        for _ in range(2000):
            cp.evaluate(x0, tgt, None, x0)
        t_eval = (time.perf_counter() - t1) / 2000
        out["cpu_baseline"] = {"value": done / t_cpu, "unit": "frames/s", "cores": 1, "kind": "port",
                               "sample": f"first {done} frames of the same workload; per frame the reference's own procedure "
                                         f"(optimizer.py:77-102): closure (value without / gradient with the regulariser) + "
                                         f"SLSQP at ftol {prob.ftol:g} from last_qpos.  Closure = the oracle's plain-C "
                                         f"restatement (oracle/csrc/dexr_oracle.c: FK, point Jacobians, SmoothL1, mimic fold); "
                                         f"SLSQP = scipy's compiled Kraft routine standing in for nlopt's; one process",
                               "evaluations_per_frame": evals / max(done, 1),
                               "closure_us_per_evaluation_incl_ctypes": t_eval * 1e6,
                               "note": "optimistic stand-in for pinocchio + nlopt + torch: the reference additionally pays "
                                       "torch autograd overhead in every evaluation (optimizer.py:266-291)",
                               "host_cpus": os.cpu_count()}
        try:
            out["reference_stack"] = reference_stack_record(wl.rel)
        except Exception as e:  # an installed but unusable stack must not cost the line
            out["reference_stack"] = {"available": False, "error": repr(e)}
        d2, t2, _ = timed_cpu(lambda *a, **k: solvers.solve_ref_as_configured(prob, *a, **k), 4.0, 25)
        out["cpu_baseline_numpy_port"] = {"value": d2 / t2, "unit": "frames/s", "cores": 1, "kind": "port",
                                          "sample": f"first {d2} frames, the same solve with the numpy closure "
                                                    f"(oracle/objectives.py): the figure rounds 1-2 reported"}
        # the same solve fanned over host processes (frames are independent): what the reference could do on this box
        avail = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
        procs = int(os.environ.get("DEXR_CPU_PROCS", min(usable_cpus(), 64)))
        if procs > 1:
            from oracle import cases, cpu_worker

            per_proc = max(8, int(out["cpu_baseline"]["value"] * 4.0))  # ~4 s of work per process
            full_ref = cases.ref_from_keypoints(prob, last_batch["host_in"]).astype(np.float32)
            res = cpu_worker.run_all_cores(wl.rel, full_ref, last_batch["host_last"], procs, per_proc)
            if res is not None:
                out["cpu_baseline_all_cores"] = {
                    "value": res[0] / res[1], "unit": "frames/s", "cores": min(res[2], usable_cpus()), "processes": res[2],
                    "kind": "port",
                    "sample": f"{res[0]} frames = {res[2]} processes x {per_proc} frames of the same workload (frame i mod "
                              f"{B} once the batch is exhausted), started together, same solver as cpu_baseline; "
                              f"{avail} CPUs in the process's affinity mask, cgroup cpu.max = {cgroup_cpu_max()!r} (a quota "
                              f"below the process count caps what the workers can use together)"}
    import bench_line

    bench_line.emit(out)  # DETAIL lines + bench_detail.json, then the compact (<= 4 KB) contract line LAST
    if comm is not None:
        comm.close()


def relaunch(args):
    """`python bench.py --gpus N` with no launcher around it: start N ranks (one process per GPU) by re-executing this
    command under torch.distributed.run -- exactly how the driver launches the N > 1 runs."""
    import socket

    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}",
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    sys.stdout.flush()
    os.execv(sys.executable, cmd)


def dry_run_launch(args):
    """--dry-run-launch: the launch / rendezvous plumbing of an N-rank run with NO GPU work (what the CPU test suite
    runs with N = 2): every rank joins the job's store, the 128-byte id rank 0 publishes reaches every rank (the same
    exchange native_comm() performs for the RCCL unique id), a gloo all-gather collects one row per rank, and rank 0
    prints the line."""
    import torch
    import torch.distributed as dist

    from dex_retargeting_amd.distributed import exchange_bytes, rendezvous_store

    rank, local_rank, world, launched = job_env(args)
    store = rendezvous_store(rank, world)
    uid = exchange_bytes(store, rank, "dexr/dry_run_id", lambda: bytes(range(128)))
    dist.init_process_group("gloo", rank=rank, world_size=world, store=dist.PrefixStore("dexr_dry", store))
    mine = torch.tensor([[float(rank), float(local_rank), float(len(uid)), float(sum(uid))]], dtype=torch.float64)
    full = torch.empty((world, 4), dtype=torch.float64)
    dist.all_gather_into_tensor(full, mine)
    dist.barrier()
    if rank == 0:
        print(json.dumps({"dry_run": True, "n_gpus": world, "world_size": world, "launched_by_torchrun": launched,
                          "ranks": full[:, 0].tolist(), "local_ranks": full[:, 1].tolist(),
                          "id_bytes": full[:, 2].tolist(), "id_checksum": full[:, 3].tolist()}))
    dist.destroy_process_group()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--batch", type=int, default=65536, help="frames per GPU")
    ap.add_argument("--workload", default="allegro_vector", choices=sorted(WORKLOADS))
    ap.add_argument("--no-cpu-baseline", action="store_true", help="skip every host-CPU solve (baseline and SLSQP distance)")
    ap.add_argument("--headline-only", action="store_true", help="skip the f64 / cold-start / other-config sub-records")
    ap.add_argument("--cpu-sample", type=int, default=20000, help="frames of the workload timed on the host CPU")
    ap.add_argument("--sustained-seconds", type=float, default=6.0, help="length of the back-to-back `sustained` record")
    ap.add_argument("--nccl-algo", default=None, help="sets NCCL_ALGO before the RCCL communicator exists (e.g. Ring, Tree, Direct)")
    ap.add_argument("--nccl-proto", default=None, help="sets NCCL_PROTO (e.g. Simple, LL, LL128)")
    ap.add_argument("--fleet-order", default="iid", choices=("iid", "sorted"),
                    help="mixed_fleet: order of the global batch (sorted = by robot; ranks take 1/N of every robot either way)")
    ap.add_argument("--probe", default=None, choices=("f64", "cold_start", "general_kernel"),
                    help="time ONLY this sub-record's loop and print a small JSON line (what tools/profile_round.sh wraps in "
                         "rocprofv3 for the sub-records' PMC summaries)")
    ap.add_argument("--dry-run-launch", action="store_true",
                    help="launch + rendezvous plumbing only (no GPU work): used by the CPU tests of the N > 1 launcher")
    args = ap.parse_args()
    if args.nccl_algo:
        os.environ["NCCL_ALGO"] = args.nccl_algo
    if args.nccl_proto:
        os.environ["NCCL_PROTO"] = args.nccl_proto
    if args.gpus > 1 and os.environ.get("RANK") is None:
        return relaunch(args)  # does not return
    if args.dry_run_launch:
        return dry_run_launch(args)
    if args.workload == "mixed_fleet":
        import bench_fleet

        if args.batch == 65536:
            args.batch = 131072  # 1 048 576 frames / 8 GPUs
        return bench_fleet.run(args)
    return run_single(args)


if __name__ == "__main__":
    main()
