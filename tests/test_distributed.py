"""world_size-2 gloo tests (CPU) of the N > 1 path: contiguous sharding + one all-gather reassembles exactly what a
single process would have produced.  The HIP solve needs a GPU, so the per-shard solver injected here is a CPU
interpreter of the SAME compiled kinematic tables (tests/table_interp.solve_vector: table-driven FK, Jacobians and
damped Gauss-Newton steps -- real per-item work whose result depends on every input of the item), plus a trivial
stand-in for the DexPilot state plumbing.  What is tested is the partition, padding, ordering and state handling that
bench.py / ShardedRetargeter rely on."""
import json
import os
import socket
import subprocess
import sys

import numpy as np
import pytest

from dex_retargeting_amd.distributed import shard_bounds

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_shard_bounds_partition():
    for B in (0, 1, 7, 64, 65536, 524288 + 3):
        for world in (1, 2, 3, 8):
            edges = [shard_bounds(B, r, world) for r in range(world)]
            assert edges[0][0] == 0 and edges[-1][1] == B
            assert all(edges[i][1] == edges[i + 1][0] for i in range(world - 1))
            sizes = [b - a for a, b in edges]
            assert max(sizes) - min(sizes) <= 1
    with pytest.raises(ValueError):
        shard_bounds(10, 2, 2)


_compiled = {}


def _allegro_tables():
    if "m" not in _compiled:
        from dex_retargeting_amd.constants import DEFAULT_URDF_DIR
        from dex_retargeting_amd.retargeting_config import RetargetingConfig
        from oracle import cases

        RetargetingConfig.set_default_urdf_dir(str(DEFAULT_URDF_DIR))
        seq = RetargetingConfig.load_from_file(os.path.join(cases.CONFIG_DIR, "teleop/allegro_hand_right.yml")).build()
        _compiled["m"] = (seq.optimizer.compiled_model(), seq.joint_limits)
    return _compiled["m"]


def _inputs(B):
    """Same seeded inputs on every rank: human-keypoint vectors and a start inside the joint limits."""
    from oracle import cases

    _, lim = _allegro_tables()
    rng = np.random.default_rng(0)
    kp = cases.human_keypoints(B, seed=1)
    ref = (kp[:, [4, 8, 12, 16]] - kp[:, [0, 0, 0, 0]]).astype(np.float32)
    last = rng.uniform(lim[:, 0], lim[:, 1], (B, lim.shape[0])).astype(np.float32)
    return ref, last


def _table_solve(ref, fixed, last, state):
    """Per-shard solver: CPU interpreter of the compiled Allegro tables (real, input-dependent per-item work)."""
    import table_interp

    if state is not None:
        state[:] = (state + 1) * 3
    if last.shape[0] == 0:
        return np.zeros_like(last, dtype=np.float32)
    return table_interp.solve_vector(_allegro_tables()[0], ref, last, iters=3)


def _worker(rank, world, port, B, out_dir):
    import torch.distributed as dist

    from dex_retargeting_amd.distributed import ShardedRetargeter

    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    ref, last = _inputs(B)
    state = np.arange(B, dtype=np.uint32)
    sr = ShardedRetargeter(solve=_table_solve, device="cpu")
    q = sr.retarget(ref, None, last, state)
    np.savez(os.path.join(out_dir, f"r{rank}.npz"), q=q, state=state)
    dist.destroy_process_group()


@pytest.mark.parametrize("B", [10, 129])
def test_two_rank_gloo_allgather_equals_single_process(tmp_path, B):
    import torch.multiprocessing as mp

    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    mp.spawn(_worker, args=(2, port, B, str(tmp_path)), nprocs=2, join=True)
    ref, last = _inputs(B)
    st = np.arange(B, dtype=np.uint32)
    want = _table_solve(ref, None, last, st)
    assert np.abs(want - last).max() > 1e-2  # the solver moved the joints: real work was sharded
    for r in range(2):
        got = np.load(os.path.join(str(tmp_path), f"r{r}.npz"))
        assert np.array_equal(got["q"], want)
        assert np.array_equal(got["state"], st)


def _pipe_worker(rank, world, port, out_dir, G):
    import torch
    import torch.distributed as dist

    from dex_retargeting_amd.distributed import PipelinedAllGather

    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    per, n, steps = 5, 3, 7
    seen = {}

    def on_full(k0, t, cnt):
        for g in range(cnt):
            seen[k0 + g] = t[:, g].clone()  # (world, per, n) of step k0 + g

    pg = PipelinedAllGather(per, n, torch.float32, "cpu", depth=2, on_full=on_full, steps_per_gather=G)
    for k in range(steps):
        out = pg.shard(k)
        out.copy_(torch.full((per, n), float(100 * k + rank)))  # the "solve" of step k on this rank
        pg.gather(k)
        with pytest.raises(RuntimeError):
            pg.gather(k)  # each step once
    last = pg.finish()
    assert sorted(seen) == list(range(steps))
    assert tuple(last.shape) == (world, G, per, n)
    torch.save({"seen": seen}, os.path.join(out_dir, f"p{rank}.pt"))
    dist.destroy_process_group()


@pytest.mark.parametrize("G", [1, 3])
def test_pipelined_all_gather_two_ranks_gloo(tmp_path, G):
    """Every step's gathered tensor holds rank r's shard in slot r, in step order, although buffers rotate and (G = 3)
    three steps share one collective, the last group being only partly filled."""
    import torch
    import torch.multiprocessing as mp

    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    mp.spawn(_pipe_worker, args=(2, port, str(tmp_path), G), nprocs=2, join=True)
    for rank in range(2):
        d = torch.load(os.path.join(str(tmp_path), f"p{rank}.pt"))
        for k, t in d["seen"].items():
            want = torch.stack([torch.full((5, 3), float(100 * k + r)) for r in range(2)])
            assert torch.equal(t, want), (rank, k)


class _GlooComm:
    """Stand-in for _lib.Comm on host memory: the same pointer-level allgather(send, recv, bytes_per_rank, stream)
    contract as dexr_allgather, carried by a gloo all_gather."""

    def __init__(self, rank, world):
        self.rank, self.world = rank, world

    def allgather(self, send_ptr, recv_ptr, bytes_per_rank, stream=0):
        import ctypes

        import torch
        import torch.distributed as dist

        n = bytes_per_rank // 4
        send = torch.from_numpy(np.ctypeslib.as_array((ctypes.c_float * n).from_address(send_ptr)))
        recv = torch.from_numpy(np.ctypeslib.as_array((ctypes.c_float * (n * self.world)).from_address(recv_ptr)))
        dist.all_gather_into_tensor(recv, send)


def _native_gather_worker(rank, world, port, out_dir, G):
    import torch
    import torch.distributed as dist

    from dex_retargeting_amd.distributed import NativeGather

    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    per, n, steps, depth = 5, 3, 11, 2
    ng = NativeGather(_GlooComm(rank, world), per, n, "cpu", depth=depth, overlap=True, steps_per_gather=G)
    seen = {}
    for k in range(steps):
        out = ng.shard(k)
        out.copy_(torch.full((per, n), float(100 * k + rank)))  # the "solve" of step k on this rank
        before = ng.collectives
        ng.gather(k)
        if ng.collectives > before:  # a group went out: its G steps are complete on every rank
            full = ng._full[ng._last]
            for g in range(G):
                seen[k - (G - 1) + g] = (full[:, g] if G > 1 else full).clone()
    full = ng.finish()
    tail = steps % G
    for g in range(tail):
        seen[steps - tail + g] = full[:, g].clone()
    assert ng.collectives == -(-steps // G)
    torch.save({"seen": seen}, os.path.join(out_dir, f"n{rank}.pt"))
    dist.destroy_process_group()


@pytest.mark.parametrize("G", [1, 4])
def test_native_gather_k_step_mode_two_ranks_gloo(tmp_path, G):
    """NativeGather's group / buffer-rotation logic with world size 2 (the communicator replaced by a gloo stand-in with
    the C-ABI's pointer contract): with steps_per_gather = 4, four steps share one collective, eleven steps need three
    (the last one partly filled), and every rank sees every rank's rows of every step in order."""
    import torch
    import torch.multiprocessing as mp

    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    mp.spawn(_native_gather_worker, args=(2, port, str(tmp_path), G), nprocs=2, join=True)
    for rank in range(2):
        d = torch.load(os.path.join(str(tmp_path), f"n{rank}.pt"))
        assert sorted(d["seen"]) == list(range(11))
        for k, t in d["seen"].items():
            want = torch.stack([torch.full((5, 3), float(100 * k + r)) for r in range(2)])
            assert torch.equal(t, want), (rank, k)


def test_every_communicator_gets_its_own_store_key(monkeypatch):
    """a fixed key hands a stale unique id to the non-zero ranks of a second communicator / a restarted
    attempt.  Keys are distinct per communicator and per elastic restart, and identical across ranks (no exchange)."""
    from dex_retargeting_amd import distributed

    monkeypatch.setattr(distributed, "_comm_generation", [0])
    monkeypatch.delenv("TORCHELASTIC_RESTART_COUNT", raising=False)
    a, b = distributed.comm_key(), distributed.comm_key()
    assert a != b and a == "dexr/unique_id/r0/c0" and b == "dexr/unique_id/r0/c1"
    monkeypatch.setenv("TORCHELASTIC_RESTART_COUNT", "2")
    monkeypatch.setattr(distributed, "_comm_generation", [0])
    assert distributed.comm_key() == "dexr/unique_id/r2/c0"


def test_steps_per_gather_policy():
    from dex_retargeting_amd.distributed import steps_per_gather_for

    shard = 65536 * 16 * 4
    assert steps_per_gather_for(0.047, shard, 1) == 1                      # one GPU: nothing to gather
    assert steps_per_gather_for(0.047, shard, 8) == 16                     # Allegro at 8 GPUs: wire time > step -> max
    k = steps_per_gather_for(1.34, 65536 * 24 * 4, 8)                      # Shadow DexPilot: the solve hides the gather
    assert k == 1
    k2 = steps_per_gather_for(0.047, shard, 2)                             # 2 GPUs: 13 us of wire per 47 us step
    assert 1 <= k2 <= 2


def test_bench_watchdog_prints_the_line_it_has_and_ends_the_job():
    """bench.py's N > 1 path arms a deadline before every optional measurement: a collective that misbehaves across ranks
    hangs instead of raising.  When the deadline passes, rank 0 prints the headline line it already holds (with what timed
    out) and the process leaves with exit code 0; a non-zero rank just leaves."""
    code = ("import sys, time; sys.argv = ['bench.py']; sys.path.insert(0, %r); import bench;"
            "wd = bench.Watchdog(RANK, 0.6);"
            "wd.line = {'metric': 'm', 'value': 1.0}; wd.done['no_gather'] = {'ms_per_step': 1.0};"
            "wd.arm('graph_replay'); time.sleep(30); print('NOT REACHED')") % REPO
    for rank in (0, 1):
        r = subprocess.run([sys.executable, "-c", code.replace("RANK", str(rank))], capture_output=True, text=True, timeout=60)
        assert r.returncode == 0 and "NOT REACHED" not in r.stdout, (r.returncode, r.stdout, r.stderr[-500:])
        if rank == 0:
            line = json.loads(r.stdout.strip().splitlines()[-1])
            assert line["value"] == 1.0 and "graph_replay" in line["multi_gpu"]["watchdog"]
            assert line["multi_gpu"]["no_gather"] == {"ms_per_step": 1.0}
        else:
            assert r.stdout.strip() == ""


def test_all_cores_cpu_baseline_starts_the_workers_it_reports():
    """the all-cores record said 64 processes and had started 7.  plan_workers starts exactly `procs`
    workers of `frames_per_proc` frames each (wrapping around the batch), and run_all_cores reports the number started."""
    from oracle import cases, cpu_worker

    assert cpu_worker.plan_workers(65536, 64, 9658) == [(w * 9658, 9658) for w in range(64)]
    assert cpu_worker.plan_workers(0, 4, 10) == []
    rel = "teleop/allegro_hand_right.yml"
    prob = cases.problem_from_config(rel)
    d = cases.human_set(prob, 6)
    res = cpu_worker.run_all_cores(rel, d["ref"], d["last"], 3, 4, deadline_s=100.0)
    assert res is not None
    frames, seconds, workers = res
    assert (frames, workers) == (12, 3) and seconds > 0  # 3 workers x 4 frames out of a 6-frame batch: indices wrap


def test_bench_gpus_n_starts_n_ranks_by_itself():
    """`python bench.py --gpus 2` with no launcher and no WORLD_SIZE in the environment must start two ranks (the way
    the driver invokes the N = 1 run).  --dry-run-launch keeps the GPU out of it: every rank joins the job's store,
    receives the 128-byte id rank 0 published (the exchange native_comm performs for the RCCL unique id) and takes part
    in a gloo all-gather; rank 0's line carries n_gpus = world_size = 2."""
    import json
    import subprocess

    repo = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT",
                                                            "TORCHELASTIC_USE_AGENT_STORE")}
    r = subprocess.run([sys.executable, os.path.join(repo, "bench.py"), "--gpus", "2", "--dry-run-launch"], env=env,
                       capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout
    rec = json.loads(lines[0])
    assert rec["n_gpus"] == 2 and rec["world_size"] == 2 and rec["launched_by_torchrun"] is True
    assert rec["ranks"] == [0.0, 1.0] and rec["local_ranks"] == [0.0, 1.0]
    assert rec["id_bytes"] == [128.0, 128.0] and rec["id_checksum"] == [float(sum(range(128)))] * 2


def test_unique_id_is_created_without_a_gpu_and_has_the_abi_size():
    from dex_retargeting_amd import _lib

    a, b = _lib.comm_unique_id(), _lib.comm_unique_id()
    assert len(a) == _lib.UNIQUE_ID_BYTES == 128 and a != b
    with pytest.raises(ValueError):
        _lib.Comm(b"short", 0, 1)


def test_committed_bench_lines_follow_the_driver_contract():
    """The JSON lines committed under profiles/ (copies of what `python bench.py` printed on the GPU box) carry every
    field the driver parses, the roofline and cpu_baseline objects, and self-consistent numbers."""
    import json

    REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    prof = os.path.join(REPO, "profiles")
    lines = {}
    # (this used to validate the round-3 lines; now the lines of the last two rounds, as the driver's boxes printed them)
    names = ("r05_bench_default.json", "r05_bench_mixed_fleet.json", "r06_bench_default.json", "r06_bench_mixed_fleet.json",
             "r06_bench_1rank_native_rccl_allegro_vector.json", "r06_bench_1rank_native_rccl_leap_position.json",
             "r06_bench_1rank_native_rccl_mixed_fleet.json")
    for name in names:
        with open(os.path.join(prof, name)) as f:
            lines[name] = json.loads([ln for ln in f if ln.startswith("{")][-1])
        assert len(json.dumps(lines[name], separators=(",", ":"))) <= 4096, name  # the compact line the driver parses
    base = json.load(open(os.path.join(REPO, "BASELINE.json")))
    for name, d in lines.items():
        for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
                  "vs_baseline", "dtype", "data", "config"):
            assert k in d, (name, k)
        assert d["higher_is_better"] is True and d["scaling"] == "weak" and d["vs_baseline"] is None
        assert "synthetic" in d["data"] and "workload" in d["config"]
        fleet = "mixed_fleet" in name
        # (the arithmetic type the path computes in: the tip kernel is float32 throughout; a fleet mixes kernels, said per model)
        assert d["dtype"] == "f32" or (fleet and "f32" in d["dtype"]), (name, d["dtype"])
        assert d["unit"] == "frames/s"
        if not fleet and "leap_position" not in name:
            assert d["metric"] == base["metric"], (d["metric"], base["metric"])
        r = d["roofline"]
        for k in ("bound", "achieved", "peak", "unit", "frac", "traffic"):
            assert k in r, (name, k)
        assert r["bound"] == "hbm" and r["unit"] == "GB/s" and abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-5 * r["frac"] + 1e-9  # (compact lines: 6 significant digits each)
        assert 0 < r["frac"] < 1 and r["peak"] == 8000.0
        # throughput and step time describe the same run
        frames = d["config"].get("batch_per_gpu", d["config"].get("frames_per_gpu", None))
        if frames:
            assert abs(d["value"] - d["n_gpus"] * frames / (d["ms_per_step"] * 1e-3)) / d["value"] < 0.02
    for name in ("r05_bench_default.json", "r06_bench_default.json"):
        c = lines[name]["cpu_baseline"]
        for k in ("value", "unit", "cores", "kind", "sample"):
            assert k in c
        assert c["kind"] in ("port", "reference") and c["cores"] >= 1 and c["value"] > 0
    assert lines["r06_bench_1rank_native_rccl_allegro_vector.json"]["config"].get("rccl_world_size") == 1
    # round 6: the blocks VERDICT r5 asked for travel in the compact line
    d6 = lines["r06_bench_default.json"]
    for k in ("sustained", "reference_profile_script", "online_ms_per_retarget", "small_batch", "also", "f64"):
        assert k in d6, k
    assert d6["reference_profile_script"]["rows"] == 14 and d6["sustained"]["steps"] >= 40000
    assert all("parity" in v for v in d6["online_ms_per_retarget"].values())
    assert all("f64" in v and "f32" in v["dtype"] for v in d6["also"].values())


# ---- skew-proof fleet sharding ----------------------------------------------------
FLEET_RELS = ("teleop/allegro_hand_right.yml", "teleop/leap_hand_right.yml")
FLEET_COST = np.array([1.0, 20.0])  # relative per-frame solve cost used as "work" (a per-finger vector model vs a heavy one)


def _fleet_tables():
    if "fleet" not in _compiled:
        from dex_retargeting_amd.constants import DEFAULT_URDF_DIR
        from dex_retargeting_amd.retargeting_config import RetargetingConfig
        from oracle import cases

        RetargetingConfig.set_default_urdf_dir(str(DEFAULT_URDF_DIR))
        out = []
        for rel in FLEET_RELS:
            seq = RetargetingConfig.load_from_file(os.path.join(cases.CONFIG_DIR, rel)).build()
            out.append((seq.optimizer.compiled_model(), seq.joint_limits, np.asarray(seq.optimizer.target_link_human_indices)))
        _compiled["fleet"] = out
    return _compiled["fleet"]


def _fleet_inputs(B, order):
    from oracle import cases

    tabs = _fleet_tables()
    rng = np.random.default_rng(5)
    mid = rng.integers(0, len(tabs), B).astype(np.int32)
    if order == "sorted":
        mid = np.sort(mid)  # a batch sorted by robot: contiguous slices would give rank 0 only model 0
    kp = cases.human_keypoints(B, seed=3).astype(np.float32)
    n_max = max(t[1].shape[0] for t in tabs)
    last = np.zeros((B, n_max), np.float32)
    for m, (_, lim, _) in enumerate(tabs):
        sel = mid == m
        last[sel, : lim.shape[0]] = rng.uniform(lim[:, 0], lim[:, 1], (int(sel.sum()), lim.shape[0]))
    return mid, kp, last


def _fleet_solve(mid, kp, last, state):
    """Per-shard fleet solver on the CPU: every frame goes through the table interpreter of ITS model."""
    import table_interp

    tabs = _fleet_tables()
    out = np.zeros_like(last, dtype=np.float32)
    for m, (compiled, lim, hidx) in enumerate(tabs):
        sel = np.nonzero(mid == m)[0]
        if sel.size == 0:
            continue
        ref = (kp[sel][:, hidx[1]] - kp[sel][:, hidx[0]]).astype(np.float32)
        out[sel, : lim.shape[0]] = table_interp.solve_vector(compiled, ref, last[sel, : lim.shape[0]], iters=2)
    if state is not None:
        state[:] = state * 2 + mid.astype(np.uint32)
    return out


def _fleet_worker(rank, world, port, B, order, out_dir):
    import torch.distributed as dist

    from dex_retargeting_amd.distributed import ShardedFleet

    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    mid, kp, last = _fleet_inputs(B, order)
    state = np.arange(B, dtype=np.uint32)
    sf = ShardedFleet(_fleet_solve, last.shape[1], device="cpu", n_models=len(FLEET_RELS))
    q = sf.retarget(mid, kp, last, state)
    work = float(FLEET_COST[mid[sf.last_shards[rank]]].sum())
    np.savez(os.path.join(out_dir, f"f{rank}.npz"), q=q, state=state, work=work, n=sf.last_shards[rank].size)
    dist.destroy_process_group()


def test_shard_by_model_balances_every_model_whatever_the_batch_order():
    from dex_retargeting_amd.distributed import shard_by_model, unshard_by_model

    rng = np.random.default_rng(0)
    for B in (0, 1, 7, 1000, 131072):
        for world in (1, 2, 3, 8):
            for order in ("iid", "sorted", "one_model"):
                mid = rng.integers(0, 4, B)
                if order == "sorted":
                    mid = np.sort(mid)
                if order == "one_model":
                    mid[:] = 2
                sh = shard_by_model(mid, world, 4)
                assert len(sh) == world
                cat = np.concatenate(sh) if B else np.zeros(0, np.int64)
                assert np.array_equal(np.sort(cat), np.arange(B))            # disjoint cover
                assert all(np.all(np.diff(x) > 0) for x in sh if x.size > 1)  # ascending batch positions
                sizes = [x.size for x in sh]
                assert max(sizes) - min(sizes) <= 1                           # equal slots for the all-gather
                for m in range(4):
                    c = [int((mid[x] == m).sum()) for x in sh]
                    assert max(c) - min(c) <= 1, (B, world, order, m, c)      # 1/N of EVERY model's frames
                # reassembly is exact
                per = max(1, -(-B // world))
                full = np.full((world, per, 3), -1.0)
                for r, x in enumerate(sh):
                    full[r, : x.size] = np.stack([x, x * 2, mid[x]], 1) if x.size else np.zeros((0, 3))
                back = unshard_by_model(full, sh, B)
                assert np.array_equal(back[:, 0], np.arange(B)) and np.array_equal(back[:, 2], mid)
    with pytest.raises(ValueError):
        shard_by_model(np.array([0, 5]), 2, 4)
    with pytest.raises(ValueError):
        shard_by_model(np.array([0, -1]), 2)
    # the contiguous split this replaces, on the same sorted batch: one rank gets all of the heavy model
    mid = np.sort(rng.integers(0, 2, 4000))
    lo, hi = shard_bounds(4000, 0, 2)
    heavy_contig = [int((mid[a:b] == 1).sum()) for a, b in (shard_bounds(4000, r, 2) for r in range(2))]
    heavy_by_model = [int((mid[x] == 1).sum()) for x in shard_by_model(mid, 2, 2)]
    assert min(heavy_contig) == 0 and abs(heavy_by_model[0] - heavy_by_model[1]) <= 1


@pytest.mark.parametrize("order", ["sorted", "iid"])
def test_two_rank_gloo_fleet_sorted_by_robot_is_balanced_and_reassembled_exactly(tmp_path, order):
    """A mixed-fleet batch SORTED BY ROBOT over two ranks: both ranks receive the same solve work (per-model frame counts
    within one frame of each other, hence cost-weighted work within one heavy frame) and the all-gather reassembles,
    in batch order, exactly the rows and state words a single process computes."""
    import torch.multiprocessing as mp

    B = 203
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    mp.spawn(_fleet_worker, args=(2, port, B, order, str(tmp_path)), nprocs=2, join=True)
    mid, kp, last = _fleet_inputs(B, order)
    st = np.arange(B, dtype=np.uint32)
    want = _fleet_solve(mid, kp, last, st)
    assert np.abs(want - last).max() > 1e-2  # real per-frame work was sharded
    got = [np.load(os.path.join(str(tmp_path), f"f{r}.npz")) for r in range(2)]
    for g in got:
        assert np.array_equal(g["q"], want) and np.array_equal(g["state"], st)
    assert abs(int(got[0]["n"]) - int(got[1]["n"])) <= 1
    assert abs(float(got[0]["work"]) - float(got[1]["work"])) <= FLEET_COST.max() + FLEET_COST.min()
    if order == "sorted":  # what contiguous B/N slices would have given the two ranks on this batch
        contig = [float(FLEET_COST[mid[a:b]].sum()) for a, b in (shard_bounds(B, r, 2) for r in range(2))]
        assert max(contig) / min(contig) > 3.0
