"""Shared by the profiling tools: apply a dict of developer knobs to a model handle / solve options.

The knobs used to be process-wide environment variables read inside libdexr on every launch; they are fields of
``dexr_tuning`` (per model handle, ``Model.tune``) and ``dexr_solve_options`` now.  The tools keep their short knob
names."""
from dex_retargeting_amd import _lib

_TUNE = {"persist_from": int, "persist_occ": int, "qchunk": int, "resident_waves": int, "max_blind": int,
         "stall_from": int, "stall_ratio": float, "stall_cap": float, "lam_jump": float, "lam_fastdec": float, "lam_recover": float,
         "floor_scale": float, "step_cap": float, "blind_tol_scale": float, "chain": int, "pivot_rule": int, "longest_first": int}
_KERNEL = {"auto": _lib.KERNEL_AUTO, "register": _lib.KERNEL_REGISTER, "quad": _lib.KERNEL_QUAD, "lds": _lib.KERNEL_LDS,
           "reduced": _lib.KERNEL_REDUCED, "wide": _lib.KERNEL_WIDE}
_OPTS = {"max_iter": int, "tol": float, "lambda0": float, "newton": int, "polish": int, "strict": int}


def apply(model, knobs: dict):
    """knobs: {'kernel': 'quad', 'persist_from': 0, 'newton': 0, ...} -> (Tuning, SolveOptions or None)."""
    tk = {k: _TUNE[k](v) for k, v in knobs.items() if k in _TUNE}
    if "kernel" in knobs:
        tk["kernel"] = _KERNEL[knobs["kernel"]]
    t = model.tune(**tk)
    ok = {k: _OPTS[k](v) for k, v in knobs.items() if k in _OPTS}
    unknown = set(knobs) - set(tk) - set(ok) - {"kernel"}
    if unknown:
        raise KeyError(f"unknown knobs {sorted(unknown)}")
    return t, (_lib.default_options(**ok) if ok else None)
