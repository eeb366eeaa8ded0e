"""URDF + optimizer spec -> flat kinematic tables for the HIP kernels (host, cold path).

Produces the blob described in include/dexr_tables.h.  What is encoded, per reference behaviour:

* joint value sources: optimised variable / fixed_qpos entry / mimic of another joint
  (``qpos[idx_pin2target] = x``, ``qpos[idx_pin2fixed] = fixed_qpos``, ``adaptor.forward_qpos``:
  /root/reference/src/dex_retargeting/optimizer.py:141-151,244-254 and kinematics_adaptor.py:102-105);
* body-frame placement = parent joint pose * constant offset (robot_wrapper.py:85-87 over pinocchio's
  fixed-joint folding);
* residual terms: task/origin link pairs for Vector/DexPilot (optimizer.py:226-234, 357-392, 417-428), single
  links for Position (optimizer.py:134);
* components: optimised joints that no term couples are solved independently (exact: the objective is a sum
  over terms plus a diagonal regulariser, optimizer.py:272-274,300).

Frames are re-aligned so every joint axis is local +z (see dexr_tables.h); this does not change any link
position.  Nothing here runs per frame.
"""
from __future__ import annotations

from dataclasses import dataclass, field
from typing import Dict, List, Optional, Sequence, Tuple

import numpy as np

from . import generic_tables as gt
from .urdf import KinematicModel


class TableOverflow(ValueError):
    """The model does not fit the fixed-size component records (include/dexr_tables.h); compile_model / compile_fk
    then emit the generic table, which the general kernel serves."""


MAXJ, MAXF, MAXT, NSLOT = 32, 16, 16, 3
MAGIC, VERSION = 0x52584544, 5
KIND_VECTOR, KIND_POSITION, KIND_DEXPILOT, KIND_FKONLY = 0, 1, 2, 3
SRC_OPT, SRC_FIXED, SRC_MIMIC, SRC_DIRECT = 0, 1, 2, 3

COMP_DTYPE = np.dtype([
    ("n_joint", "<i4"), ("n_frame", "<i4"), ("n_term", "<i4"), ("n_base_frame", "<i4"),
    ("X", "<f4", (MAXJ, 12)),
    ("jtype", "<i4", (MAXJ,)), ("restore", "<i4", (MAXJ,)), ("save", "<i4", (MAXJ,)),
    ("src_kind", "<i4", (MAXJ,)), ("src_idx", "<i4", (MAXJ,)), ("api", "<i4", (MAXJ,)),
    ("fbeg", "<i4", (MAXJ,)), ("fend", "<i4", (MAXJ,)),
    ("mult", "<f4", (MAXJ,)), ("off", "<f4", (MAXJ,)), ("lo", "<f4", (MAXJ,)), ("hi", "<f4", (MAXJ,)),
    ("frame_joint", "<i4", (MAXF,)), ("frame_off", "<f4", (MAXF, 3)), ("frame_anc", "<u4", (MAXF,)),
    ("term_task", "<i4", (MAXT,)), ("term_origin", "<i4", (MAXT,)), ("term_ref", "<i4", (MAXT,)),
    ("n_var", "<i4"), ("var", "<i4", (MAXJ,)), ("vmul", "<f4", (MAXJ,)), ("var_joint", "<i4", (MAXJ,)),
])
HEADER_DTYPE = np.dtype([
    ("magic", "<u4"), ("version", "<u4"), ("kind", "<i4"), ("n_opt", "<i4"), ("n_fixed", "<i4"),
    ("n_ref", "<i4"), ("n_comp", "<i4"), ("num_fingers", "<i4"), ("n_q", "<i4"), ("comp_bytes", "<i4"),
    ("huber_delta", "<f4"), ("norm_delta", "<f4"), ("scaling", "<f4"), ("inv_norm", "<f4"),
    ("project_dist", "<f4"), ("escape_dist", "<f4"), ("eta1", "<f4"), ("eta2", "<f4"),
    ("n_keypoints", "<i4"), ("human_origin", "<i4", (MAXT,)), ("human_task", "<i4", (MAXT,)),
])


def _align_z_to(axis: np.ndarray) -> np.ndarray:
    """Rotation A with A @ e_z == axis (unit)."""
    a = np.asarray(axis, dtype=np.float64)
    a = a / np.linalg.norm(a)
    z = np.array([0.0, 0.0, 1.0])
    c = float(a @ z)
    if c > 1.0 - 1e-14:
        return np.eye(3)
    if c < -1.0 + 1e-14:
        return np.diag([1.0, -1.0, -1.0])  # Rx(pi)
    v = np.cross(z, a)
    K = np.array([[0, -v[2], v[1]], [v[2], 0, -v[0]], [-v[1], v[0], 0]])
    return np.eye(3) + K + K @ K / (1.0 + c)


@dataclass
class TermSpec:
    task_link: str
    origin_link: Optional[str]  # None for position terms
    ref_row: int


@dataclass
class CompiledModel:
    kind: int
    n_opt: int
    n_fixed: int
    n_ref: int
    n_q: int
    header: np.ndarray
    comps: np.ndarray  # structured array (n_comp,)
    comp_vars: List[List[int]] = field(default_factory=list)  # api indices solved by each component
    generic: Optional[bytes] = None  # generic table (dexr_gen_header + arrays) of a model that outgrows the records above

    @property
    def n_comp(self) -> int:
        return int(self.comps.shape[0])

    @property
    def max_joints(self) -> int:
        return int(self.comps["n_joint"].max()) if self.n_comp else 0

    def to_blob(self) -> bytes:
        if self.generic is not None:
            return self.header.tobytes() + self.generic
        return self.header.tobytes() + self.comps.tobytes()

    def save(self, path: str) -> None:
        """Write the table blob (the on-disk format of include/dexr_tables.h) so a deployment can create models
        without the URDF / YAML / this compiler: ``_lib.Model(open(path, 'rb').read())``."""
        with open(path, "wb") as f:
            f.write(self.to_blob())

    @staticmethod
    def from_blob(blob: bytes) -> "CompiledModel":
        if len(blob) < HEADER_DTYPE.itemsize:
            raise ValueError("blob shorter than the header")
        header = np.frombuffer(blob[:HEADER_DTYPE.itemsize], dtype=HEADER_DTYPE)[0].copy()
        if int(header["magic"]) != MAGIC or int(header["version"]) != VERSION:
            raise ValueError("not a dexr table blob of this version")
        n = int(header["n_comp"])
        body = blob[HEADER_DTYPE.itemsize:]
        if n == 0 and len(body) >= gt.GEN_HEADER_DTYPE.itemsize and \
                int(np.frombuffer(body[:4], dtype="<u4")[0]) == gt.GEN_MAGIC:
            return CompiledModel(int(header["kind"]), int(header["n_opt"]), int(header["n_fixed"]), int(header["n_ref"]),
                                 int(header["n_q"]), header, np.zeros(0, dtype=COMP_DTYPE), [], generic=bytes(body))
        if int(header["comp_bytes"]) != COMP_DTYPE.itemsize or len(body) != n * COMP_DTYPE.itemsize:
            raise ValueError("blob size does not match its header")
        comps = np.frombuffer(body, dtype=COMP_DTYPE).copy()
        return CompiledModel(int(header["kind"]), int(header["n_opt"]), int(header["n_fixed"]), int(header["n_ref"]),
                             int(header["n_q"]), header, comps, [])

    @staticmethod
    def load(path: str) -> "CompiledModel":
        with open(path, "rb") as f:
            return CompiledModel.from_blob(f.read())


def _build_component(model: KinematicModel, joint_set: Sequence[int], frames: List[Tuple[str, int, np.ndarray]],
                     terms: List[Tuple[int, int, int]], src: Dict[int, tuple], lo: np.ndarray, hi: np.ndarray,
                     direct: bool = False) -> np.ndarray:
    """joint_set: sorted pin indices; frames: (link name, parent pin joint or -1, offset(3) in that joint frame);
    terms: (task frame idx, origin frame idx or -1, ref row) with frame indices into `frames`."""
    rec = np.zeros((), dtype=COMP_DTYPE)
    pins = list(joint_set)
    nj = len(pins)
    if nj > MAXJ:
        raise TableOverflow(f"component needs {nj} joints, table format supports {MAXJ}")
    if len(frames) > MAXF:
        raise TableOverflow(f"component needs {len(frames)} frames, table format supports {MAXF}")
    if len(terms) > MAXT:
        raise TableOverflow(f"component needs {len(terms)} terms, table format supports {MAXT}")
    local = {p: k for k, p in enumerate(pins)}
    parent_local = []
    for p in pins:
        pp = model.joints[p].parent
        while pp >= 0 and pp not in local:  # cannot happen: ancestors are always included
            pp = model.joints[pp].parent
        parent_local.append(local[pp] if pp >= 0 else -1)

    # re-alignment rotations
    A = [_align_z_to(model.joints[p].axis) for p in pins]

    # fork bookkeeping -> restore/save slots
    children: Dict[int, List[int]] = {}
    for k, pk in enumerate(parent_local):
        children.setdefault(pk, []).append(k)
    slot_of: Dict[int, int] = {}
    live_until: Dict[int, int] = {}
    for k in range(nj):
        ch = children.get(k, [])
        needs = [c for c in ch if c != k + 1]
        if needs:
            live_until[k] = max(needs)
    free_slots = list(range(NSLOT))
    active: List[Tuple[int, int]] = []  # (joint, slot)
    restore = np.full(nj, -1, dtype=np.int32)
    save = np.full(nj, -1, dtype=np.int32)
    for k in range(nj):
        pk = parent_local[k]
        if pk == -1:
            restore[k] = -2
        elif pk == k - 1:
            restore[k] = -1
        else:
            restore[k] = slot_of[pk]
        # release slots whose last user has been processed
        for (j, s) in list(active):
            if live_until[j] <= k:
                active.remove((j, s))
                free_slots.append(s)
                free_slots.sort()
        if k in live_until and live_until[k] > k:
            if not free_slots:
                raise TableOverflow("kinematic tree forks deeper than the kernel's saved-transform slots")
            s = free_slots.pop(0)
            slot_of[k] = s
            save[k] = s
            active.append((k, s))

    rec["n_joint"] = nj
    for k, p in enumerate(pins):
        j = model.joints[p]
        Ap = A[parent_local[k]] if parent_local[k] >= 0 else np.eye(3)
        T = j.placement
        R = Ap.T @ T[:3, :3] @ A[k]
        t = Ap.T @ T[:3, 3]
        rec["X"][k, :9] = R.reshape(-1)
        rec["X"][k, 9:] = t
        rec["jtype"][k] = 0 if j.type == "revolute" else 1
        s = src[p]
        rec["src_kind"][k] = s[0]
        if s[0] == SRC_OPT:
            rec["src_idx"][k] = s[1]
            rec["api"][k] = s[1]
            rec["mult"][k], rec["off"][k] = 1.0, 0.0
            rec["lo"][k], rec["hi"][k] = lo[s[1]], hi[s[1]]
        elif s[0] == SRC_FIXED:
            rec["src_idx"][k], rec["api"][k] = s[1], -1
            rec["mult"][k], rec["off"][k] = s[2], s[3]
        elif s[0] == SRC_MIMIC:
            rec["src_idx"][k], rec["api"][k] = local[s[1]], -1
            rec["mult"][k], rec["off"][k] = s[2], s[3]
        else:
            rec["src_idx"][k], rec["api"][k] = s[1], -1
            rec["mult"][k], rec["off"][k] = 1.0, 0.0
    rec["restore"][:nj] = restore
    rec["save"][:nj] = save
    # reduced variables: optimised joints in joint order; mimic joints ride on their source's variable
    rec["var"][:] = -1
    rec["var_joint"][:] = -1
    nv = 0
    for k in range(nj):
        if int(rec["src_kind"][k]) == SRC_OPT:
            rec["var"][k], rec["vmul"][k] = nv, 1.0
            rec["var_joint"][nv] = k
            nv += 1
    for k in range(nj):
        if int(rec["src_kind"][k]) == SRC_MIMIC:
            rec["var"][k] = rec["var"][int(rec["src_idx"][k])]
            rec["vmul"][k] = rec["mult"][k]
    rec["n_var"] = nv

    # frames sorted by attached local joint (base first), remember permutation for terms
    order = sorted(range(len(frames)), key=lambda i: (local[frames[i][1]] if frames[i][1] >= 0 else -1, i))
    newidx = {old: new for new, old in enumerate(order)}
    rec["n_frame"] = len(frames)
    nbase = 0
    rec["fbeg"][:] = 0
    rec["fend"][:] = 0
    for new, old in enumerate(order):
        name, pj, offv = frames[old]
        lj = local[pj] if pj >= 0 else -1
        rec["frame_joint"][new] = lj
        if lj >= 0:
            rec["frame_off"][new] = A[lj].T @ offv
            anc = 0
            a = lj
            while a >= 0:
                anc |= (1 << a)
                a = parent_local[a]
            rec["frame_anc"][new] = anc
        else:
            rec["frame_off"][new] = offv
            rec["frame_anc"][new] = 0
            nbase += 1
    rec["n_base_frame"] = nbase
    for k in range(nj):
        idxs = [new for new in range(len(frames)) if rec["frame_joint"][new] == k]
        rec["fbeg"][k] = idxs[0] if idxs else 0
        rec["fend"][k] = idxs[-1] + 1 if idxs else 0
    rec["n_term"] = len(terms)
    for t, (ft, fo, row) in enumerate(terms):
        rec["term_task"][t] = newidx[ft]
        rec["term_origin"][t] = newidx[fo] if fo >= 0 else -1
        rec["term_ref"][t] = row
    return rec


def compile_model(model: KinematicModel, kind: int, idx_pin2target: Sequence[int], idx_pin2fixed: Sequence[int],
                  terms: Sequence[TermSpec], *, lower: np.ndarray, upper: np.ndarray,
                  mimic: Sequence[Tuple[int, int, float, float]] = (), huber_delta: float = 0.02,
                  norm_delta: float = 4e-3, scaling: float = 1.0, num_fingers: int = 0,
                  project_dist: float = 0.03, escape_dist: float = 0.05, eta1: float = 1e-4,
                  eta2: float = 3e-2, human_indices: Optional[np.ndarray] = None,
                  n_keypoints: int = 21, force_generic: bool = False) -> CompiledModel:
    """mimic: (mimic pin idx, source pin idx, multiplier, offset).  lower/upper: optimiser box per target joint
    (already widened; +-inf allowed)."""
    n_opt, n_fixed = len(idx_pin2target), len(idx_pin2fixed)
    # what the fixed-size records cannot hold goes to the generic table (one component, general kernel)
    overflow = force_generic or len(terms) > MAXT or (kind == KIND_DEXPILOT and num_fingers > 5)
    opt_of_pin = {int(p): i for i, p in enumerate(idx_pin2target)}
    fixed_of_pin = {int(p): i for i, p in enumerate(idx_pin2fixed)}
    mimic_of_pin = {int(m): (int(s), float(a), float(b)) for (m, s, a, b) in mimic}

    src: Dict[int, tuple] = {}
    root_var: Dict[int, Optional[int]] = {}  # pin joint -> api variable it depends on (None if constant)
    for p in range(model.dof):
        if p in mimic_of_pin:
            s, a, b = mimic_of_pin[p]
            if s in mimic_of_pin:
                raise ValueError("mimic joints that mimic other mimic joints are not supported")
            if s in opt_of_pin:
                src[p] = (SRC_MIMIC, s, a, b)
                root_var[p] = opt_of_pin[s]
            elif s in fixed_of_pin:
                src[p] = (SRC_FIXED, fixed_of_pin[s], a, b)
                root_var[p] = None
            else:
                raise ValueError(f"mimic source joint {model.joints[s].name} is neither optimised nor fixed")
        elif p in opt_of_pin:
            src[p] = (SRC_OPT, opt_of_pin[p])
            root_var[p] = opt_of_pin[p]
        elif p in fixed_of_pin:
            src[p] = (SRC_FIXED, fixed_of_pin[p], 1.0, 0.0)
            root_var[p] = None
        else:
            raise ValueError(f"joint {model.joints[p].name} is neither target, fixed nor mimic")

    lo = np.clip(np.asarray(lower, dtype=np.float64), -3.0e38, 3.0e38)
    hi = np.clip(np.asarray(upper, dtype=np.float64), -3.0e38, 3.0e38)

    # frames used by the terms
    def frame_of(link: str):
        f = model.frames[model.body_frame_index(link)]
        return f.parent, f.placement[:3, 3].copy()

    # union-find over optimised variables
    uf = list(range(n_opt))

    def find(a):
        while uf[a] != a:
            uf[a] = uf[uf[a]]
            a = uf[a]
        return a

    term_vars: List[List[int]] = []
    for t in terms:
        vs = set()
        for link in (t.task_link, t.origin_link):
            if link is None:
                continue
            pj, _ = frame_of(link)
            for a in (model.ancestors(pj) if pj >= 0 else []):
                if root_var[a] is not None:
                    vs.add(root_var[a])
        term_vars.append(sorted(vs))
    if kind == KIND_DEXPILOT:
        for i in range(1, n_opt):
            uf[find(i)] = find(0)  # the projection pre-amble needs all vectors of an item together
    for vs in term_vars:
        for v in vs[1:]:
            uf[find(v)] = find(vs[0])
    touched = set(v for vs in term_vars for v in vs)
    roots: List[int] = []
    for v in range(n_opt):
        if v in touched and find(v) not in roots:
            roots.append(find(v))
    if not roots:
        roots = [find(0)] if n_opt else []
    # orphans (variables no term depends on) ride along with the first component: they only see the regulariser
    orphan_root = roots[0] if roots else None

    comps = []
    comp_vars: List[List[int]] = []
    pin_of_opt = {i: int(p) for i, p in enumerate(idx_pin2target)}
    for r in roots:
        vars_c = [v for v in range(n_opt) if (find(v) == r if v in touched else r == orphan_root)]
        my_terms = [i for i, vs in enumerate(term_vars) if (vs and find(vs[0]) == r) or (not vs and r == roots[0])]
        jset = set()
        fr_list: List[Tuple[str, int, np.ndarray]] = []
        fr_index: Dict[str, int] = {}
        tl: List[Tuple[int, int, int]] = []
        for i in my_terms:
            t = terms[i]
            ids = []
            for link in (t.task_link, t.origin_link):
                if link is None:
                    ids.append(-1)
                    continue
                if link not in fr_index:
                    pj, offv = frame_of(link)
                    fr_index[link] = len(fr_list)
                    fr_list.append((link, pj, offv))
                    if pj >= 0:
                        jset.update(model.ancestors(pj))
                ids.append(fr_index[link])
            tl.append((ids[0], ids[1], t.ref_row))
        for v in vars_c:  # orphans and plain variables: make sure their own joint is present
            jset.add(pin_of_opt[v])
            jset.update(model.ancestors(pin_of_opt[v]))
        changed = True
        while changed:  # mimic joints need their source joint in the same table
            changed = False
            for p in list(jset):
                if src[p][0] == SRC_MIMIC and src[p][1] not in jset:
                    jset.add(src[p][1])
                    jset.update(model.ancestors(src[p][1]))
                    changed = True
        if not overflow:
            try:
                comps.append(_build_component(model, sorted(jset), fr_list, tl, src, lo, hi))
            except TableOverflow:
                overflow = True
        comp_vars.append(vars_c)

    generic = None
    if overflow:
        generic = _generic_of(model, terms, src, lo, hi, [int(p) for p in idx_pin2target], human_indices)
        comps, comp_vars = [], [list(range(n_opt))]

    n_ref = len(terms)
    inv_norm = 1.0 / (3 * n_ref) if kind == KIND_POSITION else 1.0 / max(n_ref, 1)
    header = np.zeros((), dtype=HEADER_DTYPE)
    header["magic"], header["version"] = MAGIC, VERSION
    header["kind"], header["n_opt"], header["n_fixed"], header["n_ref"] = kind, n_opt, n_fixed, n_ref
    header["n_comp"], header["num_fingers"], header["n_q"] = len(comps), num_fingers, model.dof
    header["comp_bytes"] = COMP_DTYPE.itemsize
    header["huber_delta"], header["norm_delta"], header["scaling"] = huber_delta, norm_delta, scaling
    header["inv_norm"] = inv_norm
    header["project_dist"], header["escape_dist"], header["eta1"], header["eta2"] = project_dist, escape_dist, eta1, eta2
    if human_indices is not None:  # target_link_human_indices: (2, n_ref) for vector/dexpilot, (n_ref,) for position
        hi = np.asarray(human_indices, dtype=np.int64)
        if hi.ndim == 1:
            origin, task = np.full(hi.shape[0], -1), hi
        else:
            origin, task = hi[0], hi[1]
        if task.shape[0] != n_ref:
            raise ValueError("target_link_human_indices does not match the number of reference rows")
        if task.max(initial=0) >= n_keypoints or origin.max(initial=-1) >= n_keypoints:
            raise ValueError("target_link_human_indices exceeds the keypoint count")
        header["n_keypoints"] = n_keypoints
        header["human_origin"][:min(n_ref, MAXT)] = origin[:MAXT]  # generic tables carry the full map per term
        header["human_task"][:min(n_ref, MAXT)] = task[:MAXT]
    if generic is not None:
        header["comp_bytes"] = 0
    return CompiledModel(kind, n_opt, n_fixed, n_ref, model.dof, header,
                         np.array(comps, dtype=COMP_DTYPE).reshape(-1), comp_vars, generic=generic)


def _human_maps(human_indices, n_ref):
    if human_indices is None:
        return None
    hi = np.asarray(human_indices, dtype=np.int64)
    return (np.full(hi.shape[0], -1), hi) if hi.ndim == 1 else (hi[0], hi[1])


def _generic_of(model: KinematicModel, terms: Sequence[TermSpec], src: Dict[int, tuple], lo, hi,
                opt_pins: Sequence[int], human_indices) -> bytes:
    """The whole model as ONE generic table: every joint any term or optimised variable needs, every target link."""
    jset = set()
    fr_list: List[Tuple[str, int, np.ndarray]] = []
    fr_index: Dict[str, int] = {}
    tl: List[Tuple[int, int, int]] = []
    for t in terms:
        ids = []
        for link in (t.task_link, t.origin_link):
            if link is None:
                ids.append(-1)
                continue
            if link not in fr_index:
                f = model.frames[model.body_frame_index(link)]
                fr_index[link] = len(fr_list)
                fr_list.append((link, f.parent, f.placement[:3, 3].copy()))
                if f.parent >= 0:
                    jset.update(model.ancestors(f.parent))
            ids.append(fr_index[link])
        tl.append((ids[0], ids[1], t.ref_row))
    for p in opt_pins:
        jset.add(p)
        jset.update(model.ancestors(p))
    changed = True
    while changed:  # mimic joints need their source joint in the table
        changed = False
        for p in list(jset):
            if src[p][0] == SRC_MIMIC and src[p][1] not in jset:
                jset.add(src[p][1])
                jset.update(model.ancestors(src[p][1]))
                changed = True
    return gt.build_generic(model, sorted(jset), fr_list, tl, src, lo, hi, _human_maps(human_indices, len(terms)))


def compile_fk(model: KinematicModel, link_names: Sequence[str]) -> CompiledModel:
    """Table for plain forward kinematics of a list of links from a full pinocchio-order qpos
    (RobotWrapper.compute_forward_kinematics + get_link_pose, robot_wrapper.py:82-87).  Output row l of the
    FK kernel is the world position of link_names[l]."""
    src = {p: (SRC_DIRECT, p) for p in range(model.dof)}
    comps = []
    try:
        return _compile_fk_fixed(model, link_names, src)
    except TableOverflow:
        pass
    # a chain of more than DEXR_MAXJ joints (arm + hand + free joints): one generic table, general kernel
    terms = [TermSpec(link, None, i) for i, link in enumerate(link_names)]
    z = np.zeros(1)
    fr_list = []
    jset = set()
    for i, link in enumerate(link_names):
        f = model.frames[model.body_frame_index(link)]
        fr_list.append((f"{link}#{i}", f.parent, f.placement[:3, 3].copy()))
        if f.parent >= 0:
            jset.update(model.ancestors(f.parent))
    generic = gt.build_generic(model, sorted(jset), fr_list, [(i, -1, i) for i in range(len(link_names))], src, z, z, None)
    header = np.zeros((), dtype=HEADER_DTYPE)
    header["magic"], header["version"], header["kind"] = MAGIC, VERSION, KIND_FKONLY
    header["n_ref"], header["n_q"], header["inv_norm"] = len(link_names), model.dof, 1.0
    header["n_comp"], header["comp_bytes"] = 0, 0
    return CompiledModel(KIND_FKONLY, 0, 0, len(link_names), model.dof, header, np.zeros(0, dtype=COMP_DTYPE), [],
                         generic=generic)


def _compile_fk_fixed(model: KinematicModel, link_names: Sequence[str], src) -> CompiledModel:
    comps = []
    # one component per chunk of <= MAXF links
    for c0 in range(0, len(link_names), MAXF):
        chunk = list(link_names[c0:c0 + MAXF])
        jset = set()
        frames = []
        terms = []
        for i, link in enumerate(chunk):
            f = model.frames[model.body_frame_index(link)]
            frames.append((f"{link}#{i}", f.parent, f.placement[:3, 3].copy()))
            if f.parent >= 0:
                jset.update(model.ancestors(f.parent))
            terms.append((i, -1, c0 + i))
        z = np.zeros(1)
        comps.append(_build_component(model, sorted(jset), frames, terms, src, z, z, direct=True))
    header = np.zeros((), dtype=HEADER_DTYPE)
    header["magic"], header["version"], header["kind"] = MAGIC, VERSION, KIND_FKONLY
    header["n_opt"], header["n_fixed"], header["n_ref"] = 0, 0, len(link_names)
    header["n_comp"], header["n_q"], header["comp_bytes"] = len(comps), model.dof, COMP_DTYPE.itemsize
    header["inv_norm"] = 1.0
    return CompiledModel(KIND_FKONLY, 0, 0, len(link_names), model.dof, header,
                         np.array(comps, dtype=COMP_DTYPE).reshape(-1), [])
