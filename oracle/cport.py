"""ORACLE (test infrastructure only -- never imported by the product path).

ctypes binding of oracle/csrc/dexr_oracle.c (plain-C restatement of one evaluation of the reference's objective closures,
optimizer.py:146-198, 249-304, 510-575) and the CPU baseline built on it:

* :class:`CProblem`  -- flattens an :class:`oracle.objectives.OracleProblem` (chains of the computed links from
  oracle/kin.py's raw URDF tree) into the arrays the C code walks; ``evaluate`` = one closure call.
* :func:`solve_ref_as_configured_c` -- what the reference does per frame (optimizer.py:77-102): the closure (value
  WITHOUT, gradient WITH the norm_delta term) handed to SLSQP with ftol 1e-6 / 1e-5 and the +-1e-3 box.  scipy's
  compiled SLSQP stands in for nlopt's (same Kraft routine); the per-evaluation work is compiled C instead of numpy
  -- the closest this container gets to the cost structure of pinocchio + nlopt (minus the reference's torch autograd
  overhead per evaluation, which makes this an OPTIMISTIC stand-in for the reference's CPU path).

The library is built by ``build()`` (gcc) into oracle/_build/ -- called from __graft_entry__.build(); built artefacts are
git-ignored and travel to the GPU box with the snapshot.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess
from typing import Optional

import numpy as np

from .objectives import OracleProblem

HERE = os.path.dirname(os.path.abspath(__file__))
SRC = os.path.join(HERE, "csrc", "dexr_oracle.c")
LIB = os.path.join(HERE, "_build", "liboracle.so")
_lib = None


def build(force: bool = False) -> str:
    if force or not os.path.exists(LIB) or os.path.getmtime(LIB) < os.path.getmtime(SRC):
        os.makedirs(os.path.dirname(LIB), exist_ok=True)
        r = subprocess.run(["gcc", "-O2", "-shared", "-fPIC", "-o", LIB, SRC, "-lm"], capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError("gcc failed for the oracle's C restatement:\n" + r.stderr[-2000:])
    return LIB


def load():
    global _lib
    if _lib is None:
        lib = C.CDLL(build())
        ip, dp = C.POINTER(C.c_int), C.POINTER(C.c_double)
        lib.oracle_create.restype = C.c_void_p
        lib.oracle_create.argtypes = [C.c_int] * 7 + [C.c_double, C.c_double] + [ip] * 5 + [dp, dp] + [ip] * 3 + [dp] * 3 + [ip, ip]
        lib.oracle_destroy.argtypes = [C.c_void_p]
        lib.oracle_destroy.restype = None
        lib.oracle_evaluate.restype = C.c_double
        lib.oracle_evaluate.argtypes = [C.c_void_p, dp, dp, dp, dp, dp, dp]
        lib.oracle_link_positions.argtypes = [C.c_void_p, dp, dp]
        lib.oracle_link_positions.restype = None
        _lib = lib
    return _lib


def _i(a):
    return np.ascontiguousarray(a, dtype=np.int32)


def _d(a):
    return np.ascontiguousarray(a, dtype=np.float64)


def _p(a, t):
    return a.ctypes.data_as(C.POINTER(t))


class CProblem:
    KIND = {"vector": 0, "position": 1, "dexpilot": 2}

    def __init__(self, prob: OracleProblem):
        self.prob = prob
        r = prob.robot
        off, jt, jq, R0, p0, ax = [0], [], [], [], [], []
        for name in prob.computed_links:
            for j in r._chain(name):
                jt.append({"fixed": 0, "revolute": 1, "prismatic": 2}[j.type])
                jq.append(r.qidx[j.name] if j.type != "fixed" else -1)
                R0.append(j.R0.reshape(9))
                p0.append(j.p0)
                ax.append(j.axis)
            off.append(len(jt))
        self._keep = [_i(prob.idx_pin2target), _i(prob.idx_pin2fixed), _i(prob.idx_pin2mimic), _i(prob.idx_pin2source),
                      _i(prob.idx_target2source), _d(prob.multipliers), _d(prob.offsets), _i(off), _i(jt), _i(jq),
                      _d(np.array(R0).reshape(-1)), _d(np.array(p0).reshape(-1)), _d(np.array(ax).reshape(-1)),
                      _i(getattr(prob, "origin_idx", [])), _i(getattr(prob, "task_idx", []))]
        k = self._keep
        lib = load()
        self._h = lib.oracle_create(self.KIND[prob.kind], r.dof, prob.n_opt, len(prob.idx_pin2fixed), len(prob.computed_links),
                                    prob.n_ref, len(prob.mimic), float(prob.huber_delta), float(prob.norm_delta),
                                    _p(k[0], C.c_int), _p(k[1], C.c_int), _p(k[2], C.c_int), _p(k[3], C.c_int),
                                    _p(k[4], C.c_int), _p(k[5], C.c_double), _p(k[6], C.c_double), _p(k[7], C.c_int),
                                    _p(k[8], C.c_int), _p(k[9], C.c_int), _p(k[10], C.c_double), _p(k[11], C.c_double),
                                    _p(k[12], C.c_double), _p(k[13], C.c_int), _p(k[14], C.c_int))
        if not self._h:
            raise RuntimeError("oracle_create failed (chain longer than 64 joints or more than 512 dofs)")
        self._grad = np.zeros(prob.n_opt)

    def __del__(self):
        if getattr(self, "_h", None) and _lib is not None:
            _lib.oracle_destroy(self._h)
            self._h = None

    def target(self, ref_value, dexpilot_ref=None) -> np.ndarray:
        """The float64 target rows one frame's closure compares against (formed ONCE per frame, like the reference does
        outside the closure: optimizer.py:246 scales the float32 ref_value in float32; :507 DexPilot's rounded rows)."""
        p = self.prob
        if p.kind == "vector":
            rv = np.asarray(ref_value)
            return _d((rv * rv.dtype.type(p.scaling)).astype(np.float64).reshape(-1, 3))
        if p.kind == "dexpilot":
            return _d(np.asarray(dexpilot_ref, dtype=np.float64).reshape(-1, 3))
        return _d(np.asarray(ref_value).astype(np.float64).reshape(-1, 3))

    def evaluate(self, x, tgt, fixed=None, last=None, weights=None, need_grad=True):
        """One closure call: (value without the norm_delta term, gradient with it)."""
        x = _d(x)
        fx = None if fixed is None or np.size(fixed) == 0 else _d(fixed)
        la = None if last is None else _d(last)
        w = None if weights is None else _d(weights)
        g = self._grad if need_grad else None
        f = load().oracle_evaluate(self._h, _p(x, C.c_double), _p(tgt, C.c_double),
                                   None if fx is None else _p(fx, C.c_double), None if la is None else _p(la, C.c_double),
                                   None if w is None else _p(w, C.c_double), None if g is None else _p(g, C.c_double))
        return f, (g.copy() if need_grad else None)

    def link_positions(self, q) -> np.ndarray:
        q = _d(q)
        out = np.zeros((len(self.prob.computed_links), 3))
        load().oracle_link_positions(self._h, _p(q, C.c_double), _p(out, C.c_double))
        return out


def solve_ref_as_configured_c(cp: CProblem, ref, fixed, last, weights=None, dexpilot_ref=None):
    """Per-item SLSQP exactly as configured by the reference (see oracle/solvers.solve_ref_as_configured), with the C
    closure.  Returns (x (B,n) float32, n_evals (B,))."""
    from scipy.optimize import minimize

    prob = cp.prob
    ref = np.asarray(ref)
    B = ref.shape[0]
    last = np.asarray(last).reshape(B, -1)
    lo, hi = prob.bounds
    bounds = list(zip(lo, hi))
    out = np.zeros((B, prob.n_opt), dtype=np.float32)
    evals = np.zeros(B, dtype=int)
    lib, h = load(), cp._h
    dp = C.POINTER(C.c_double)
    g = np.zeros(prob.n_opt)
    gp = g.ctypes.data_as(dp)
    for b in range(B):
        last64 = _d(last[b].astype(np.float32))  # optimizer.py:93
        tgt = cp.target(ref[b], None if dexpilot_ref is None else dexpilot_ref[b])
        fx = None if fixed is None or np.size(fixed) == 0 else _d(np.asarray(fixed).reshape(B, -1)[b])
        w = None if weights is None else _d(weights[b])
        tp, lp = tgt.ctypes.data_as(dp), last64.ctypes.data_as(dp)
        fp = None if fx is None else fx.ctypes.data_as(dp)
        wp = None if w is None else w.ctypes.data_as(dp)
        cnt = [0]

        def fun(x):
            cnt[0] += 1
            x = np.ascontiguousarray(x, dtype=np.float64)
            f = lib.oracle_evaluate(h, x.ctypes.data_as(dp), tp, fp, lp, wp, gp)
            return f, g.copy()

        x0 = np.clip(last[b].astype(np.float64), lo, hi)
        res = minimize(fun, x0, jac=True, method="SLSQP", bounds=bounds, options=dict(ftol=prob.ftol, maxiter=200))
        out[b] = res.x.astype(np.float32)  # optimizer.py:99
        evals[b] = cnt[0]
    return out, evals
