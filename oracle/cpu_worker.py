"""ORACLE (test infrastructure only -- never imported by the product path).

One host process of bench.py's ``cpu_baseline_all_cores`` leg: solves a slice of the bench workload with the
reference-as-configured CPU solve (oracle.cport.solve_ref_as_configured_c == the reference's per-frame
``Optimizer.retarget``, optimizer.py:77-102: the closure evaluated by the oracle's plain-C restatement, scipy's
compiled SLSQP standing in for nlopt).

    python -m oracle.cpu_worker <config.yml (relative)> <inputs.npz> <start> <count> <sync dir>

Protocol: import, solve one untimed frame, touch ``<sync dir>/ready.<start>``, wait for ``<sync dir>/go`` (at most
60 s), solve ``count`` frames, print ``<count> <seconds>``.  :func:`run_all_cores` is the parent side; every wait
in it has a deadline, and any failure makes it return None (the bench then simply omits the leg).
"""
from __future__ import annotations

import os
import subprocess
import sys
import tempfile
import time

import numpy as np


_cp = {}


def _solve(prob, ref, last):
    """The compiled-closure port (oracle/cport.py) when its library is there, the numpy port otherwise."""
    from . import cport, solvers

    kw = {}
    if prob.kind == "dexpilot":
        w, rv, _ = prob.dexpilot_preamble(ref, np.zeros((ref.shape[0], prob.n_pair), bool))
        kw = dict(weights=w, dexpilot_ref=rv)
    try:
        if id(prob) not in _cp:
            _cp[id(prob)] = cport.CProblem(prob)
    except Exception:
        _cp[id(prob)] = None
    if _cp[id(prob)] is not None:
        cport.solve_ref_as_configured_c(_cp[id(prob)], ref, None, last, **kw)
    else:
        solvers.solve_ref_as_configured(prob, ref, None, last, **kw)


def main(argv):
    from . import cases

    rel, npz, start, count, sync = argv[0], argv[1], int(argv[2]), int(argv[3]), argv[4]
    prob = cases.problem_from_config(rel)
    d = np.load(npz)
    ref, last = d["ref"][start:start + count], d["last"][start:start + count]
    _solve(prob, ref[:1], last[:1])
    open(os.path.join(sync, f"ready.{start}"), "w").close()
    deadline = time.time() + 60
    while not os.path.exists(os.path.join(sync, "go")) and time.time() < deadline:
        time.sleep(0.005)
    t0 = time.perf_counter()
    _solve(prob, ref, last)
    print(ref.shape[0], time.perf_counter() - t0, flush=True)


def plan_workers(n_available: int, procs: int, frames_per_proc: int):
    """Frame index ranges of the all-cores leg: exactly `procs` workers, `frames_per_proc` frames each, taken from the
    workload's frames in order and wrapping around when procs x frames_per_proc exceeds what one batch holds (round 3
    capped the total at the batch size but kept the per-worker count, which started 7 workers where 64 were reported).
    Returns a list of (start, count) into the wrapped index sequence  i -> i mod n_available."""
    procs = max(1, int(procs))
    per = max(1, int(frames_per_proc))
    return [(w * per, per) for w in range(procs)] if n_available > 0 else []


def run_all_cores(rel: str, ref: np.ndarray, last: np.ndarray, procs: int, frames_per_proc: int, deadline_s: float = 120.0):
    """Returns (frames, wall seconds between 'go' and the last worker finishing, workers started) or None on any
    failure/timeout."""
    plan = plan_workers(ref.shape[0], procs, frames_per_proc)
    if not plan:
        return None
    n = plan[-1][0] + plan[-1][1]
    idx = np.arange(n) % ref.shape[0]
    env = dict(os.environ, OMP_NUM_THREADS="1", OPENBLAS_NUM_THREADS="1", MKL_NUM_THREADS="1")
    repo = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    t_end = time.time() + deadline_s
    workers = []
    with tempfile.TemporaryDirectory() as sync:
        npz = os.path.join(sync, "inputs.npz")
        np.savez(npz, ref=ref[idx], last=last[idx])
        try:
            for s, c in plan:
                workers.append(subprocess.Popen([sys.executable, "-m", "oracle.cpu_worker", rel, npz, str(s), str(c), sync],
                                                cwd=repo, env=env, stdout=subprocess.PIPE, stderr=subprocess.DEVNULL, text=True))
            while sum(os.path.exists(os.path.join(sync, f"ready.{s}")) for s, _ in plan) < len(plan):
                if time.time() > t_end or any(w.poll() not in (None, 0) for w in workers):
                    raise TimeoutError("workers did not come up")
                time.sleep(0.01)
            t0 = time.perf_counter()
            open(os.path.join(sync, "go"), "w").close()
            frames = 0
            for w in workers:
                out, _ = w.communicate(timeout=max(1.0, t_end - time.time()))
                frames += int(out.split()[0])
            return frames, time.perf_counter() - t0, len(workers)
        except Exception:
            return None
        finally:
            for w in workers:
                if w.poll() is None:
                    w.kill()


if __name__ == "__main__":
    main(sys.argv[1:])
