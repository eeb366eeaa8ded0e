"""Generic table format (include/dexr_tables.h, "generic tables"): the compiled form of models that outgrow the fixed-size
component records -- more than 32 joints in one component (a 7-DoF arm + a 24-DoF hand + 6 free joints), more than 16
reference rows (a 6-finger DexPilot problem has 21 vectors), more than 16 target links, deep tree forks.  The reference
accepts all of these (optimizer.py:18-52: any URDF, any number of links / vectors); they are served by the general
kernel (csrc/dexr_gen.hpp) instead of a ValueError.  Host, cold path."""
from __future__ import annotations

from typing import Dict, List, Optional, Sequence, Tuple

import numpy as np

from .urdf import KinematicModel

GEN_MAGIC = 0x47584544
GEN_MAX = 64
GEN_HEADER_DTYPE = np.dtype([("magic", "<u4"), ("n_joint", "<i4"), ("n_frame", "<i4"), ("n_term", "<i4"), ("n_var", "<i4"),
                             ("n_fam", "<i4"), ("max_depth", "<i4"), ("has_keypoint_map", "<i4")])
SRC_OPT, SRC_FIXED, SRC_MIMIC, SRC_DIRECT = 0, 1, 2, 3


def _pad2(a: np.ndarray) -> np.ndarray:
    a = np.ascontiguousarray(a, dtype=np.int32).reshape(-1)
    return a if a.size % 2 == 0 else np.concatenate([a, np.zeros(1, np.int32)])


def build_generic(model: KinematicModel, joint_set: Sequence[int], frames: List[Tuple[str, int, np.ndarray]],
                  terms: List[Tuple[int, int, int]], src: Dict[int, tuple], lo: np.ndarray, hi: np.ndarray,
                  human: Optional[Tuple[np.ndarray, np.ndarray]] = None) -> bytes:
    """joint_set: sorted pin (dof) indices, closed under ancestors; frames: (name, parent pin joint or -1, offset (3) in
    that joint's frame); terms: (task frame idx, origin frame idx or -1, ref row); src: pin joint -> value source tuple as
    built by model_compiler.compile_model; lo / hi: box per api index; human: (origin, task) keypoint index per ref row."""
    pins = list(joint_set)
    nj, nf, nt = len(pins), len(frames), len(terms)
    if nj > GEN_MAX or nf > GEN_MAX or nt > GEN_MAX:
        raise ValueError(f"model needs {nj} joints / {nf} target links / {nt} reference rows; the general kernel serves up "
                         f"to {GEN_MAX} of each (DEXR_GEN_MAXJ, include/dexr_tables.h)")
    local = {p: k for k, p in enumerate(pins)}
    parent = np.array([local.get(model.joints[p].parent, -1) if model.joints[p].parent >= 0 else -1 for p in pins], np.int32)
    depth = np.zeros(nj, np.int32)
    for k in range(nj):  # pins are in depth-first order: parents come first
        depth[k] = 0 if parent[k] < 0 else depth[parent[k]] + 1
    X = np.zeros((nj, 12))
    axis = np.zeros((nj, 3))
    jmul, joff = np.ones(nj), np.zeros(nj)
    jtype = np.zeros(nj, np.int32)
    src_idx = np.zeros(nj, np.int32)
    var = np.full(nj, -1, np.int32)
    var_api: List[int] = []
    for k, p in enumerate(pins):
        j = model.joints[p]
        X[k, :9] = j.placement[:3, :3].reshape(-1)
        X[k, 9:] = j.placement[:3, 3]
        axis[k] = j.axis
        jtype[k] = 0 if j.type == "revolute" else 1
        s = src[p]
        if s[0] == SRC_OPT:
            var[k] = len(var_api)
            var_api.append(int(s[1]))
        elif s[0] == SRC_FIXED:
            src_idx[k], jmul[k], joff[k] = s[1], s[2], s[3]
        elif s[0] == SRC_DIRECT:
            src_idx[k] = s[1]
    for k, p in enumerate(pins):
        s = src[p]
        if s[0] == SRC_MIMIC:
            var[k] = var[local[s[1]]]
            jmul[k], joff[k] = s[2], s[3]
    nv = len(var_api)
    fam_off, fam = [0], []
    for v in range(nv):
        fam += [k for k in range(nj) if var[k] == v]
        fam_off.append(len(fam))
    frame_joint = np.zeros(nf, np.int32)
    frame_off = np.zeros((nf, 3))
    frame_anc = np.zeros(nf, np.uint64)
    for i, (_, pj, offv) in enumerate(frames):
        frame_joint[i] = local[pj] if pj >= 0 else -1
        frame_off[i] = offv
        a, m = int(frame_joint[i]), 0
        while a >= 0:
            m |= 1 << a
            a = int(parent[a])
        frame_anc[i] = m
    joint_anc = np.zeros(nj, np.uint64)  # ancestors-or-self
    for k in range(nj):
        a, m = k, 0
        while a >= 0:
            m |= 1 << a
            a = int(parent[a])
        joint_anc[k] = m
    tt = np.array([[t[0], t[1], t[2]] for t in terms], np.int32).reshape(-1, 3)
    if sorted(tt[:, 2].tolist()) != list(range(nt)):
        raise ValueError("generic tables need exactly one term per reference row")
    h_o = np.full(nt, -1, np.int32)  # indexed by REFERENCE ROW
    h_t = np.zeros(nt, np.int32)
    if human is not None:
        for row in range(nt):
            h_o[row], h_t[row] = int(human[0][row]), int(human[1][row])
    hdr = np.zeros((), GEN_HEADER_DTYPE)
    hdr["magic"], hdr["n_joint"], hdr["n_frame"], hdr["n_term"], hdr["n_var"] = GEN_MAGIC, nj, nf, nt, nv
    hdr["n_fam"], hdr["max_depth"], hdr["has_keypoint_map"] = len(fam), int(depth.max(initial=0)), 0 if human is None else 1
    lo_v = np.array([lo[a] for a in var_api], np.float64) if nv else np.zeros(0)
    hi_v = np.array([hi[a] for a in var_api], np.float64) if nv else np.zeros(0)
    f64 = [X.reshape(-1), axis.reshape(-1), jmul, joff, lo_v, hi_v, frame_off.reshape(-1)]
    i32 = [jtype, parent, depth, src_idx, var, np.array(var_api, np.int32), np.array(fam_off, np.int32),
           np.array(fam, np.int32), frame_joint, tt[:, 0], tt[:, 1], tt[:, 2], h_o, h_t]
    return (hdr.tobytes() + b"".join(np.ascontiguousarray(a, np.float64).tobytes() for a in f64) + frame_anc.tobytes() + joint_anc.tobytes()
            + b"".join(_pad2(a).tobytes() for a in i32))
