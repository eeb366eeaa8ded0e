"""ORACLE (test infrastructure only -- never imported by the product path).

Worker functions for the oracle phase of the GPU parity tests and of bench.py's checker section (run in spawned
processes: numpy/scipy only, never HIP).

The float64 oracle is CPU-slow (seconds per thousand frames), so comparisons of thousands of frames -- for each of
the 39 shipped configs in tests/test_gpu_all_configs.py, 4 096 frames of each BASELINE config in bench.py -- fan the
oracle solves out over the GPU box's host cores (:func:`pooled_oracle_solve`)."""
import os
import sys

import numpy as np

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if REPO not in sys.path:
    sys.path.insert(0, REPO)


def host_pool(max_workers: int = 48):
    """A spawn-context process pool over the CPUs this process may use (never fork: the parent holds a HIP context)."""
    import multiprocessing as mp
    from concurrent.futures import ProcessPoolExecutor

    n = min(len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1), max_workers)
    return ProcessPoolExecutor(max_workers=max(1, n), mp_context=mp.get_context("spawn"))


def pooled_oracle_solve(rel, ref, last, state_in, q_gpu, chunk: int = 128, pool=None):
    """:func:`oracle_solve` over chunks of `chunk` frames in a host pool; returns the concatenated dict."""
    B = ref.shape[0]
    parts = [slice(i, min(i + chunk, B)) for i in range(0, B, chunk)]
    own = pool is None
    ex = host_pool() if own else pool
    try:
        res = list(ex.map(oracle_solve, [(rel, ref[c], last[c], None if state_in is None else state_in[c], q_gpu[c]) for c in parts]))
    finally:
        if own:
            ex.shutdown()
    return {k: np.concatenate([r[k] for r in res]) for k in res[0]}


def _kw(prob, ref, state_bits):
    if prob.kind != "dexpilot":
        return {}
    proj = ((state_bits[:, None] >> np.arange(prob.n_pair, dtype=np.uint32)) & 1).astype(bool)
    w, rv, _ = prob.dexpilot_preamble(ref, proj)
    return dict(weights=w, dexpilot_ref=rv)


def oracle_solve(args):
    """(rel, ref f32 (B,n_ref,3), last f32 (B,n_opt), state_in u32 (B,) or None, q_gpu f64 (B,n_opt)[, also_r3]) ->
    dict(want, F_want, F_gpu, pg_gpu): float64 minimiser of F from the same start, objective values of both answers and
    the projected-gradient inf-norm of F at the GPU answer."""
    import warnings

    warnings.filterwarnings("ignore")
    from oracle import cases, solvers

    rel, ref, last, state_in, q_gpu = args[:5]
    prob = cases.problem_from_config(rel)
    kw = _kw(prob, ref, state_in)
    want = solvers.solve_lm_batched(prob, ref, None, last, newton=True, max_iter=100, **kw)
    last64 = last.astype(np.float64)
    extra = {}
    if len(args) > 5 and args[5]:  # also the rounds 1-3 oracle (steps from indefinite models allowed), for the table
        extra["want_r3"] = solvers.solve_lm_batched(prob, ref, None, last, newton=True, max_iter=100, require_pd=False, **kw)
    F_want = prob.total(want, ref, None, last64, **kw)
    F_gpu = prob.total(q_gpu, ref, None, last64, **kw)
    _, g, _ = prob.evaluate(q_gpu, ref, None, last64, **kw)  # gradient of F (includes the regulariser, quirk Q1)
    lo, hi = prob.bounds
    held = ((q_gpu <= lo + 1e-9) & (g > 0)) | ((q_gpu >= hi - 1e-9) & (g < 0))
    pg = np.abs(np.where(held, 0.0, g)).max(1)
    return dict(want=want, F_want=F_want, F_gpu=F_gpu, pg_gpu=pg, **extra)


def certify_local_minimum(args):
    """(rel, ref, last, state_in, q_gpu) for a FEW frames -> (moved, dF): how far a tight scipy minimisation started AT
    the GPU answer moves it and how much it lowers F."""
    import warnings

    warnings.filterwarnings("ignore")
    from oracle import cases, solvers

    rel, ref, last, state_in, q_gpu = args
    prob = cases.problem_from_config(rel)
    kw = _kw(prob, ref, state_in)
    pol = solvers.solve_tight(prob, ref, None, last, x0=q_gpu, **kw)
    last64 = last.astype(np.float64)
    dF = prob.total(q_gpu, ref, None, last64, **kw) - prob.total(pol, ref, None, last64, **kw)
    return np.abs(pol - q_gpu).max(1), dF


def slsqp_as_configured(args):
    """(rel, ref, last, state_in) -> q (B,n_opt) f32 of the reference-as-configured solve (oracle/solvers.py)."""
    import warnings

    warnings.filterwarnings("ignore")
    from oracle import cases, solvers

    rel, ref, last, state_in = args
    prob = cases.problem_from_config(rel)
    kw = _kw(prob, ref, state_in)
    q, _ = solvers.solve_ref_as_configured(prob, ref, None, last, **kw)
    return q
