#!/usr/bin/env python3
"""Generates tests/golden/arbiter_counts.json: for EVERY config of tests/golden/multimodal_frames.npz (27 configs, 234 frames
of the round-3 MI355X run that were >= 1e-4 rad from the rounds 1-3 oracle) where does the arbiter -- scipy SLSQP, the
reference's own algorithm (optimizer.py:41,96-99), driven to convergence from the same start (oracle/solvers.solve_tight) --
land: at the library's recorded answer, at the old oracle's, elsewhere; and with which of them do the two LM oracles
(positive-definite steps only / the rounds 1-3 rule) agree.  CPU only (~2 min on 8 cores); the counts are deterministic and
tests/test_oracle.py recomputes and compares them config by config, then checks the aggregate.

    python tests/golden/gen_arbiter_counts.py
"""
import json
import os
import sys
import warnings
from concurrent.futures import ProcessPoolExecutor

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))


def counts(key):
    from oracle import cases, solvers

    d = np.load(os.path.join(HERE, "multimodal_frames.npz"))
    prob = cases.problem_from_config(key.replace("__", "/") + ".yml")
    ref, last, q_lib, q_old = (d[f"{key}__{f}"] for f in ("ref", "last", "q_gpu", "q_oracle"))
    kw = {}
    if prob.kind == "dexpilot":
        st = d[f"{key}__state_in"]
        proj = ((st[:, None] >> np.arange(prob.n_pair, dtype=np.uint32)) & 1).astype(bool)
        w, rv, _ = prob.dexpilot_preamble(ref, proj)
        kw = dict(weights=w, dexpilot_ref=rv)
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        tight = solvers.solve_tight(prob, ref, None, last, **kw)
        new = solvers.solve_lm_batched(prob, ref, None, last, newton=True, max_iter=100, **kw)
        old = solvers.solve_lm_batched(prob, ref, None, last, newton=True, max_iter=100, require_pd=False, **kw)

    def same(a, b):
        return np.abs(a - b).max(1) < 1e-4

    return key, dict(frames=int(len(ref)), tight_at_library=int(same(tight, q_lib).sum()), tight_at_old_oracle=int(same(tight, q_old).sum()),
                     new_oracle_at_tight=int(same(new, tight).sum()), new_oracle_at_library=int(same(new, q_lib).sum()),
                     old_rule_reproduces_old_oracle=int(same(old, q_old).sum()))


def main():
    d = np.load(os.path.join(HERE, "multimodal_frames.npz"))
    keys = sorted({k[: -len("__q_gpu")] for k in d.files if k.endswith("__q_gpu")})
    with ProcessPoolExecutor(max_workers=min(8, os.cpu_count() or 1)) as ex:
        out = dict(ex.map(counts, keys))
    tot = {f: sum(v[f] for v in out.values()) for f in next(iter(out.values()))}
    json.dump({"per_config": out, "total": tot}, open(os.path.join(HERE, "arbiter_counts.json"), "w"), indent=1)
    print(len(keys), "configs", tot)


if __name__ == "__main__":
    main()
