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
    f = rec.get("f64")
    if isinstance(f, dict):  # the same config in the reference's arithmetic (float64 throughout)
        out["f64"] = {"error": str(f["error"])[:80]} if "error" in f and "value" not in f else _pick(f, ("value", "ms_per_step"))
    return out


def _mini_parity(p):
    """[fraction within 1e-4 rad, frames in another minimum, of which worse] -- three numbers for blocks that come in bulk."""
    if not isinstance(p, dict) or "frac_within_1e-4" not in p:
        return None
    om = p.get("other_minimum") or {}
    return [_num(p["frac_within_1e-4"]), _num(om.get("frames")), _num(om.get("worse"))]


def final_line(d):
    """detail dict (what run_single / bench_fleet assemble) -> the compact dict printed as the last stdout line."""
    out = {k: _num(d.get(k)) for k in CONTRACT}
    cfg = d.get("config", {}) or {}
    out["config"] = {"workload": str(cfg.get("workload", ""))[:110]}
    for k in ("config_file", "batch_per_gpu", "frames_per_gpu", "n_opt", "n_ref", "rccl_world_size", "models"):
        if k in cfg:
            out["config"][k] = cfg[k] if isinstance(cfg[k], (str, list)) else _num(cfg[k])
    if "collective" in cfg:
        out["config"]["collective"] = str(cfg["collective"])[:70]
    if isinstance(d.get("solver"), dict):
        out["solver"] = _pick(d["solver"], ("iters_mean", "iters_max", "converged_frac", "active_lane_fraction"))
    out["roofline"] = _roofline(d.get("roofline"))
    if isinstance(d.get("cpu_baseline"), dict):
        c = d["cpu_baseline"]
        out["cpu_baseline"] = _pick(c, ("value", "unit", "cores", "kind"))
        out["cpu_baseline"]["sample"] = str(c.get("sample", ""))[:70]
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
                                         "parity": _mini_parity(v.get("parity"))}
        out["small_batch"]["parity_is"] = "[frac within 1e-4 rad of oracle, frames in other minimum, worse ones]"
    if isinstance(d.get("also"), dict):
        out["also"] = {k: _stub(v) for k, v in d["also"].items()}
    if isinstance(d.get("online_teleop"), dict) and isinstance(d["online_teleop"].get("robots"), dict):
        # per robot: mean ms per SeqRetargeting.retarget() call + parity of the 621 calls vs the oracle (as small_batch.parity_is)
        out["online_ms_per_retarget"] = {os.path.basename(k).replace(".yml", "").replace("_hand_right", ""):
                                         {"ms": _num(v.get("mean_ms"), 4), "parity": _mini_parity(v.get("parity"))}
                                         for k, v in d["online_teleop"]["robots"].items() if isinstance(v, dict)}
    su = d.get("sustained")
    if isinstance(su, dict):
        out["sustained"] = {"error": str(su["error"])[:100]} if "error" in su else _pick(
            su, ("steps", "seconds", "ms_per_step", "first_decile_ms_per_step", "last_decile_ms_per_step", "vs_ms_per_step",
                 "window20_right_after_ms_per_step", "window20_after_100ms_idle_ms_per_step"))
    rs = d.get("reference_profile_script")
    if isinstance(rs, dict):
        rows = rs.get("rows") if isinstance(rs.get("rows"), list) else None
        if rows:  # the reference's own benchmark script, unmodified: 14 "fps" rows -> five numbers (all rows: DETAIL line)
            fps = sorted(float(r["fps"]) for r in rows)
            pick = {(r["kind"], r["robot"]): r["fps"] for r in rows}
            out["reference_profile_script"] = {"rows": len(rows), "fps_min": _num(fps[0], 4), "fps_median": _num(fps[len(fps) // 2], 4),
                                               "fps_max": _num(fps[-1], 4), "allegro_vector": _num(pick.get(("vector", "allegro_hand")), 4),
                                               "shadow_dexpilot": _num(pick.get(("dexpilot", "shadow_hand")), 4)}
        else:
            out["reference_profile_script"] = {"error": str(rs.get("error"))[:100]}
    mg = d.get("multi_gpu")
    if isinstance(mg, dict):
        m = {k: mg[k] for k in ("steps_per_gather", "rccl_world_size", "rccl_version") if k in mg}
        for k, v in mg.items():
            if isinstance(v, dict) and ("value" in v or "ms_per_step" in v):
                m[k] = _pick(v, ("value", "ms_per_step"))
        if "watchdog" in mg:
            m["watchdog"] = str(mg["watchdog"])[:160]
        out["multi_gpu"] = m
    out["detail"] = "DETAIL lines above"
    return shrink(out)


def dumps(line):
    return json.dumps(line, allow_nan=False, separators=(",", ":"))


def shrink(line):
    """Drop optional blocks, least important first, until the line fits.  Never raises (a ValueError here cost round 4
    its contract line): if the contract fields + roofline + cpu_baseline alone are still too long, their strings are cut."""
    for k in ("two_streams", "cold_start", "solver", "general_kernel", "small_batch", "online_ms_per_retarget", "reference_profile_script",
              "sustained", "multi_gpu", "also", "f64", "parity", "detail"):
        if len(dumps(line)) <= MAX_BYTES:
            break
        line.pop(k, None)
    if len(dumps(line)) > MAX_BYTES:
        cfg = line.get("config") if isinstance(line.get("config"), dict) else {}
        line["config"] = {"workload": str(cfg.get("workload", ""))[:80]}
        if isinstance(line.get("cpu_baseline"), dict):
            line["cpu_baseline"]["sample"] = str(line["cpu_baseline"].get("sample", ""))[:60]
            line["cpu_baseline"].pop("all_cores", None)
        if isinstance(line.get("roofline"), dict):
            line["roofline"] = {k: line["roofline"].get(k) for k in ("bound", "achieved", "peak", "unit", "frac", "traffic")}
        line["metric"] = str(line.get("metric", ""))[:200]
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
    # (the detail file lives under gpurun_out/ -- scratch, merged back by gpurun -- never in the repo root: a 220-byte stub a test
    # once wrote there ended up in git, VERDICT r5 hygiene)
    for p in ([path] if path else [os.path.join(os.path.dirname(os.path.abspath(__file__)), "gpurun_out", "bench_detail.json")]):
        try:
            os.makedirs(os.path.dirname(os.path.abspath(p)), exist_ok=True)
            with open(p, "w") as f:
                json.dump(safe, f, indent=1)
        except OSError:
            pass
    sys.stdout.flush()
    try:
        line = dumps(final_line(safe))
    except Exception as e:  # the contract line is printed whatever a sub-record's shape does to the compaction above
        line = dumps({k: _num(safe.get(k)) for k in CONTRACT} | {"config": {"workload": str((safe.get("config") or {}).get("workload", ""))[:200]},
                                                                  "roofline": _roofline(safe.get("roofline")),
                                                                  "cpu_baseline": _pick(safe.get("cpu_baseline") or {}, ("value", "unit", "cores", "kind")),
                                                                  "compaction_error": repr(e)[:200]})
    print(line, flush=True)
