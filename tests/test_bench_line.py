"""The compact final bench line (bench_line.py): what the driver parses must stay small and strict.

Round 4's line was 20.4 KB and `BENCH_r04.parsed` came back null; these tests rebuild the final line from recorded detail
dicts (copies of what bench.py assembled on the GPU box) and pin its size, strictness and contract fields.
"""
import io
import json
import os
from contextlib import redirect_stdout

import pytest

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PROF = os.path.join(REPO, "profiles")

import sys  # noqa: E402

sys.path.insert(0, REPO)
import bench_line  # noqa: E402

RECORDED = ["r04_bench_default.json", "r04_bench_mixed_fleet.json", "r04_bench_1rank_native_rccl_allegro_vector.json",
            "r04_bench_1rank_native_rccl_leap_position.json", "r04_bench_1rank_native_rccl_mixed_fleet.json"]
RECORDED += [n for n in sorted(os.listdir(PROF)) if n.startswith("r05_bench_detail") or n.startswith("r06_bench_detail")]


def _detail(name):
    with open(os.path.join(PROF, name)) as f:
        txt = f.read().strip()
    if txt.startswith("{\n") or name.startswith("r05_bench_detail") or name.startswith("r06_bench_detail"):
        return json.loads(txt)
    return json.loads([ln for ln in txt.splitlines() if ln.startswith("{")][-1])


def _reject_constants(name):
    raise ValueError(f"non-strict JSON constant {name}")


@pytest.mark.parametrize("name", RECORDED)
def test_final_line_is_small_strict_and_complete(name):
    d = _detail(name)
    line = bench_line.final_line(bench_line.sanitize(d))
    s = bench_line.dumps(line)
    assert len(s.encode()) < bench_line.MAX_BYTES <= 4096, len(s)
    assert "\n" not in s
    back = json.loads(s, parse_constant=_reject_constants)
    assert back == line
    for k in bench_line.CONTRACT + ("config",):
        assert k in back, k
    assert back["higher_is_better"] is True and back["scaling"] == "weak" and back["vs_baseline"] is None
    assert back["unit"] == "frames/s" and back["dtype"] == "f32" and "workload" in back["config"]
    assert abs(back["value"] - d["value"]) / d["value"] < 1e-5 and abs(back["ms_per_step"] - d["ms_per_step"]) / d["ms_per_step"] < 1e-5
    r = back["roofline"]
    for k in ("bound", "achieved", "peak", "unit", "frac", "traffic"):
        assert k in r, k
    assert r["bound"] == "hbm" and r["unit"] == "GB/s" and r["peak"] == 8000.0
    assert abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-6 and 0 < r["frac"] < 1
    if "cpu_baseline" in d:
        c = back["cpu_baseline"]
        for k in ("value", "unit", "cores", "kind", "sample"):
            assert k in c, k
        assert c["kind"] in ("port", "reference") and c["cores"] >= 1 and c["value"] > 0
    if "parity" in d:
        assert back["parity"], "parity block lost"
    frames = back["config"].get("batch_per_gpu")
    if frames:  # throughput and step time describe the same run
        assert abs(back["value"] - back["n_gpus"] * frames / (back["ms_per_step"] * 1e-3)) / back["value"] < 0.02


def test_non_finite_values_become_null_and_oversize_blocks_are_dropped():
    d = _detail("r04_bench_default.json")
    d["value"] = float("nan")
    d["roofline"]["traffic"] = float("inf")
    d["also"] = {f"config_{i}": dict(d["also"]["leap_position"]) for i in range(40)}  # would not fit
    line = bench_line.final_line(bench_line.sanitize(d))
    s = bench_line.dumps(line)
    assert len(s) < bench_line.MAX_BYTES
    assert line["value"] is None and line["roofline"]["traffic"] is None
    assert "also" not in line and "roofline" in line and "cpu_baseline" in line and "parity" in line
    json.loads(s, parse_constant=_reject_constants)


def test_emit_prints_detail_first_and_the_compact_line_last(tmp_path):
    d = _detail("r04_bench_default.json")
    buf = io.StringIO()
    with redirect_stdout(buf):
        bench_line.emit(d, path=str(tmp_path / "bench_detail.json"))
    lines = buf.getvalue().splitlines()
    assert all(ln.startswith("DETAIL ") for ln in lines[:-1]) and len(lines) > 5
    last = json.loads(lines[-1], parse_constant=_reject_constants)
    assert len(lines[-1]) < 4096 and last["metric"] == d["metric"] and last["roofline"]["frac"] > 0
    assert [ln for ln in lines if ln.startswith("{")] == [lines[-1]]  # exactly one JSON line, and it is the last one
    full = json.load(open(tmp_path / "bench_detail.json"))
    assert full["also"]["shadow_dexpilot"]["solver"] == d["also"]["shadow_dexpilot"]["solver"]
    got = {}
    for ln in lines[:-1]:
        got.update(json.loads(ln[len("DETAIL "):]))
    assert got["roofline"] == d["roofline"] and got["cpu_baseline"] == d["cpu_baseline"]


def test_bench_scripts_print_through_bench_line():
    """No `print(json.dumps(<the whole record>))` left in the bench scripts: every record goes through bench_line.emit."""
    for name in ("bench.py", "bench_fleet.py"):
        src = open(os.path.join(REPO, name)).read()
        assert "bench_line.emit(" in src
        for ln in src.splitlines():
            if "print(json.dumps(" in ln:
                assert "dry_run" in ln, (name, ln)


def test_usable_cpus_respects_the_cgroup_quota(monkeypatch):
    import bench

    monkeypatch.setattr(bench, "cgroup_cpu_max", lambda: "1600000 100000")
    monkeypatch.setattr(os, "sched_getaffinity", lambda pid: set(range(64)), raising=False)
    assert bench.usable_cpus() == 16
    monkeypatch.setattr(bench, "cgroup_cpu_max", lambda: "max 100000")
    assert bench.usable_cpus() == 64
    monkeypatch.setattr(bench, "cgroup_cpu_max", lambda: None)
    assert bench.usable_cpus() == 64
