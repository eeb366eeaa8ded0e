"""The ONE line the driver parses, kept small.

bench.py measures many things (float64 record, cold start, the other BASELINE configs, online latency, the fleet, the
sequence kernel ...).  All of it goes to `DETAIL <json>` lines printed EARLIER and to `bench_detail.json`; the LAST stdout
line is `final_line(detail)`: the contract fields + `roofline` + `cpu_baseline` + `parity` + one stub per BASELINE config,
strict JSON (no NaN / Infinity), at most MAX_BYTES.  Round 4's 20 KB line was not parsed by the driver: this module and
tests/test_bench_line.py exist so that cannot happen again.

Measurement template in the reference: example/profiling/profile_online_retargeting.py:18-36,60-62 (one number printed
per robot); ours is one JSON line.
"""
import json
import math
import os

MAX_BYTES = 4096

CONTRACT = ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
            "vs_baseline", "dtype", "data")
ROOFLINE_KEYS = ("bound", "achieved", "peak", "unit", "frac", "traffic", "traffic_over_algorithmic", "valu_issue_frac",
                 "kernel_ms", "algorithmic_bytes_per_frame")
CPU_KEYS = ("value", "unit", "cores", "kind", "sample")


def _num(x, digits=6):
    """Finite numbers rounded to `digits` significant digits; everything non-finite -> None (strict JSON)."""
    if isinstance(x, bool) or x is None or isinstance(x, str):
        return x
    if isinstance(x, int):
        return x
    try:
        x = float(x)
    except (TypeError, ValueError):
        return None
    if not math.isfinite(x):
        return None
    if x == 0.0:
        return 0.0
    return float(f"{x:.{digits}g}")


def _pick(d, keys):
    return {k: _num(d.get(k)) for k in keys if isinstance(d, dict) and k in d}


def _roofline(r):
    if not isinstance(r, dict):
        return None
    out = _pick(r, ROOFLINE_KEYS)
    out.setdefault("traffic", None)
    return out


def _parity(p):
    if not isinstance(p, dict):
        return None
    if "frac_within_1e-4" not in p:  # per-model blocks (fleet, offline multi-robot): the worst figures over the models
        ms = [v for v in p.values() if isinstance(v, dict) and "frac_within_1e-4" in v]
        if not ms:
            return {}
        return {"models": len(ms), "subset_per_model": _num(ms[0].get("subset")),
                "min_frac_within_1e-4": _num(min(v["frac_within_1e-4"] for v in ms)),
                "max_abs_dq_rad": _num(max((v.get("max_abs_dq_rad") or 0.0) for v in ms)),
                "other_minimum": {"frames": sum(int((v.get("other_minimum") or {}).get("frames", 0)) for v in ms),
                                  "worse": sum(int((v.get("other_minimum") or {}).get("worse", 0)) for v in ms)}}
    out = _pick(p, ("subset", "max_abs_dq_rad", "frac_within_1e-4"))
    om = p.get("other_minimum")
    if isinstance(om, dict):
        out["other_minimum"] = _pick(om, ("frames", "worse"))
    return out


def _stub(rec):
    """value / ms_per_step / roofline.frac / parity fraction of a sub-record."""
    if not isinstance(rec, dict):
        return {"error": str(rec)[:120]}
    if "error" in rec and "value" not in rec:
        return {"error": str(rec["error"])[:120]}
    out = _pick(rec, ("value", "ms_per_step", "dtype"))
    r = rec.get("roofline")
    if isinstance(r, dict):
        out["roofline"] = _pick(r, ("frac", "traffic", "traffic_over_algorithmic", "kernel_ms"))
    p = rec.get("parity")
    if isinstance(p, dict):
        out["parity"] = _parity(p)
    return out


def final_line(d):
    """detail dict (what run_single / bench_fleet assemble) -> the compact dict printed as the last stdout line."""
    out = {k: _num(d.get(k)) for k in CONTRACT}
    cfg = d.get("config", {}) or {}
    out["config"] = {"workload": str(cfg.get("workload", ""))[:200]}
    for k in ("config_file", "batch_per_gpu", "frames_per_gpu", "n_opt", "n_ref", "rccl_world_size", "models"):
        if k in cfg:
            out["config"][k] = cfg[k] if isinstance(cfg[k], (str, list)) else _num(cfg[k])
    if "collective" in cfg:
        out["config"]["collective"] = str(cfg["collective"])[:160]
    if isinstance(d.get("solver"), dict):
        out["solver"] = _pick(d["solver"], ("iters_mean", "iters_max", "converged_frac", "active_lane_fraction"))
    out["roofline"] = _roofline(d.get("roofline"))
    if isinstance(d.get("cpu_baseline"), dict):
        c = d["cpu_baseline"]
        out["cpu_baseline"] = _pick(c, ("value", "unit", "cores", "kind"))
        out["cpu_baseline"]["sample"] = str(c.get("sample", ""))[:160]
        ac = d.get("cpu_baseline_all_cores")
        if isinstance(ac, dict):
            out["cpu_baseline"]["all_cores"] = _pick(ac, ("value", "cores", "processes"))
    if isinstance(d.get("parity"), dict):
        out["parity"] = _parity(d["parity"])
        v = d["parity"].get("vs_reference_as_configured")
        if isinstance(v, dict):
            out["parity"]["vs_slsqp_as_configured"] = _pick(v, ("subset", "median_abs_dq_rad", "frac_F_gpu_le_F_ref"))
    if isinstance(d.get("f64"), dict):
        out["f64"] = _stub(d["f64"])
        out["f64"]["max_abs_dq_vs_f32_rad"] = _num(d["f64"].get("max_abs_dq_vs_f32_rad"))
    for k in ("cold_start", "two_streams", "general_kernel"):
        if k in d:
            out[k] = _stub(d[k])
    if isinstance(d.get("small_batch"), dict) and "error" not in d["small_batch"]:
        sbd = d["small_batch"]
        out["small_batch"] = {"frames": _num(sbd.get("frames"))}
        for k, v in sbd.items():
            if isinstance(v, dict) and "ms_per_step" in v:
                out["small_batch"][k] = {"ms_per_step": _num(v["ms_per_step"]), "iters_max": _num(v.get("iters_max")),
                                         "four_per_wave_ms": _num((v.get("four_per_wave") or {}).get("ms_per_step")),
                                         "four_per_wave_iters_max": _num((v.get("four_per_wave") or {}).get("iters_max"))}
    if isinstance(d.get("also"), dict):
        out["also"] = {k: _stub(v) for k, v in d["also"].items()}
    if isinstance(d.get("online_teleop"), dict) and isinstance(d["online_teleop"].get("robots"), dict):
        out["online_ms_per_retarget"] = {os.path.basename(k).replace(".yml", ""): _num(v.get("mean_ms"), 4)
                                         for k, v in d["online_teleop"]["robots"].items() if isinstance(v, dict)}
    mg = d.get("multi_gpu")
    if isinstance(mg, dict):
        m = {k: mg[k] for k in ("steps_per_gather", "rccl_world_size", "rccl_version") if k in mg}
        for k, v in mg.items():
            if isinstance(v, dict) and ("value" in v or "ms_per_step" in v):
                m[k] = _pick(v, ("value", "ms_per_step"))
        if "watchdog" in mg:
            m["watchdog"] = str(mg["watchdog"])[:160]
        out["multi_gpu"] = m
    out["detail"] = "DETAIL lines above + bench_detail.json"
    return shrink(out)


def dumps(line):
    return json.dumps(line, allow_nan=False, separators=(",", ":"))


def shrink(line):
    """Drop optional blocks, least important first, until the line fits."""
    for k in ("small_batch", "online_ms_per_retarget", "two_streams", "cold_start", "general_kernel", "multi_gpu", "also", "f64", "solver"):
        if len(dumps(line)) <= MAX_BYTES:
            break
        line.pop(k, None)
    if len(dumps(line)) > MAX_BYTES:
        raise ValueError("compact bench line exceeds %d bytes" % MAX_BYTES)
    return line


def sanitize(x):
    """Plain JSON types only; non-finite floats -> None."""
    if isinstance(x, dict):
        return {str(k): sanitize(v) for k, v in x.items()}
    if isinstance(x, (list, tuple)):
        return [sanitize(v) for v in x]
    if isinstance(x, (str, bool)) or x is None or isinstance(x, int):
        return x
    try:
        f = float(x)
    except (TypeError, ValueError):
        return str(x)
    return f if math.isfinite(f) else None


def emit(detail, path=None):
    """Print the detail (one `DETAIL <json>` line per top-level block, so no single line is large), write
    bench_detail.json, then print the compact line LAST."""
    import sys

    safe = sanitize(detail)
    for k, v in safe.items():
        if isinstance(v, (dict, list)):
            print("DETAIL " + json.dumps({k: v}))
    for p in ([path] if path else [os.path.join(os.path.dirname(os.path.abspath(__file__)), "gpurun_out", "bench_detail.json"),
                                   "bench_detail.json"]):
        try:
            os.makedirs(os.path.dirname(os.path.abspath(p)), exist_ok=True)
            with open(p, "w") as f:
                json.dump(safe, f, indent=1)
        except OSError:
            pass
    sys.stdout.flush()
    print(dumps(final_line(safe)), flush=True)
