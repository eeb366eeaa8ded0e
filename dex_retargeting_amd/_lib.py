"""ctypes binding of libdexr.so (include/dexr.h).  Fails loudly when the HIP library is missing: there is no
CPU fallback anywhere in this package."""
from __future__ import annotations

import ctypes as C
import threading
import os
from typing import Optional

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("DEXR_LIB") or os.path.join(_HERE, "libdexr.so")  # DEXR_LIB: developer override

_lib: Optional[C.CDLL] = None


class DexrError(RuntimeError):
    pass


class SolveOptions(C.Structure):
    _fields_ = [("max_iter", C.c_int32), ("tol", C.c_float), ("lambda0", C.c_float), ("newton", C.c_int32),
                ("precision", C.c_int32), ("polish", C.c_int32), ("strict", C.c_int32)]


class Tuning(C.Structure):
    """Mirror of dexr_tuning (include/dexr.h): per-model launch / damping parameters."""
    _fields_ = [("struct_size", C.c_uint32), ("kernel", C.c_int32), ("chain", C.c_int32), ("persist_occ", C.c_int32),
                ("persist_from", C.c_int32), ("qchunk", C.c_int32), ("resident_waves", C.c_int32),
                ("max_blind", C.c_int32), ("stall_from", C.c_int32), ("stall_ratio", C.c_float),
                ("stall_cap", C.c_float), ("lam_jump", C.c_float), ("lam_fastdec", C.c_float),
                ("floor_scale", C.c_float), ("step_cap", C.c_float), ("blind_tol_scale", C.c_float),
                ("pivot_rule", C.c_int32), ("longest_first", C.c_int32), ("lam_recover", C.c_float),
                ("fork_streams", C.c_int32), ("user_mask", C.c_uint32), ("sprint_max_batch", C.c_int32), ("sprint_ladder", C.c_int32), ("tail_passes", C.c_int32)]


TUNE_LAM_JUMP, TUNE_LAM_FASTDEC = 1, 2
_TUNE_BITS = {"lam_jump": TUNE_LAM_JUMP, "lam_fastdec": TUNE_LAM_FASTDEC}


KERNEL_AUTO, KERNEL_REGISTER, KERNEL_QUAD, KERNEL_LDS, KERNEL_REDUCED, KERNEL_WIDE, KERNEL_GENERAL = -1, 0, 1, 2, 3, 4, 5

EXPORTS = ["dexr_last_error", "dexr_version", "dexr_device_count", "dexr_default_options", "dexr_model_create",
           "dexr_model_destroy", "dexr_model_info", "dexr_model_get_tuning", "dexr_model_set_tuning", "dexr_model_kernel",
           "dexr_model_lane_plan", "dexr_model_reserve",
           "dexr_retarget_dev", "dexr_retarget_seq_dev", "dexr_seq_compose_dev", "dexr_fleet_workspace_bytes",
           "dexr_retarget_multi_dev", "dexr_retarget_multi", "dexr_retarget", "dexr_retarget_f64",
           "dexr_retarget_kp_dev", "dexr_retarget_kp", "dexr_eval", "dexr_fk", "dexr_mano_keypoints_dev",
           "dexr_mano_keypoints", "dexr_comm_unique_id", "dexr_comm_create", "dexr_comm_destroy", "dexr_comm_info",
           "dexr_allgather", "dexr_comm_max_f64", "dexr_comm_barrier"]
UNIQUE_ID_BYTES = 128


def load() -> C.CDLL:
    global _lib
    if _lib is not None:
        return _lib
    if not os.environ.get("DEXR_NO_TORCH_PRELOAD"):
        # torch-ROCm bundles its own libamdhip64; whichever HIP runtime is loaded first serves the whole process.
        # Let torch's win when torch is installed, so tensors / streams created by torch and the kernels launched by
        # libdexr share one runtime (device-pointer entry points, bench.py).
        try:
            import torch  # noqa: F401
        except Exception:  # pragma: no cover - torch is optional for the host-pointer API
            pass
    if not os.path.exists(LIB_PATH):
        raise DexrError(f"{LIB_PATH} is missing: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
                        "(hipcc --offload-arch=gfx950). dex_retargeting_amd has no CPU fallback.")
    lib = C.CDLL(LIB_PATH)
    vp, i64, f32p, u32p, i32p, f64p = C.c_void_p, C.c_int64, C.POINTER(C.c_float), C.POINTER(C.c_uint32), \
        C.POINTER(C.c_int32), C.POINTER(C.c_double)
    optp = C.POINTER(SolveOptions)
    lib.dexr_last_error.restype = C.c_char_p
    lib.dexr_version.restype = C.c_char_p
    lib.dexr_device_count.restype = C.c_int
    lib.dexr_default_options.argtypes = [optp]
    lib.dexr_default_options.restype = None
    lib.dexr_model_create.argtypes = [C.c_void_p, C.c_size_t, C.POINTER(vp)]
    lib.dexr_model_destroy.argtypes = [vp]
    lib.dexr_model_destroy.restype = None
    lib.dexr_model_info.argtypes = [vp, C.c_void_p]
    lib.dexr_model_get_tuning.argtypes = [vp, C.POINTER(Tuning)]
    lib.dexr_model_set_tuning.argtypes = [vp, C.POINTER(Tuning)]
    lib.dexr_model_kernel.argtypes = [vp, i32p, i32p, i32p]
    lib.dexr_model_reserve.argtypes = [vp, C.c_int64]
    lib.dexr_model_lane_plan.argtypes = [vp, C.c_int32, i32p, i32p, vp, vp]
    lib.dexr_retarget_dev.argtypes = [vp, i64, vp, vp, vp, vp, vp, vp, vp, vp, optp, vp]
    lib.dexr_retarget_seq_dev.argtypes = [vp, i64, C.c_int32, vp, C.c_int32, vp, vp, vp, vp, vp, C.c_float, optp, vp]
    lib.dexr_seq_compose_dev.argtypes = [i64, C.c_int32, C.c_int32, C.c_int32, C.c_int32, i32p, i32p, f64p, f64p, vp, vp,
                                         C.c_double, vp, C.c_int32, vp, vp]
    lib.dexr_fleet_workspace_bytes.argtypes = [i64]
    lib.dexr_fleet_workspace_bytes.restype = C.c_size_t
    lib.dexr_retarget_multi_dev.argtypes = [C.POINTER(vp), C.c_int32, i64, vp, vp, vp, C.c_int32, vp, C.c_int32, vp, vp, vp,
                                            optp, vp, C.c_size_t, vp]
    lib.dexr_retarget_multi.argtypes = [C.POINTER(vp), C.c_int32, i64, i32p, f32p, f32p, C.c_int32, f32p, C.c_int32, u32p,
                                        f32p, i32p, optp]
    lib.dexr_retarget.argtypes = [vp, i64, f32p, f32p, f32p, u32p, f32p, i32p, i32p, f32p, optp]
    lib.dexr_retarget_kp_dev.argtypes = [vp, i64, vp, vp, vp, vp, vp, vp, vp, vp, optp, vp]
    lib.dexr_retarget_kp.argtypes = [vp, i64, f32p, f32p, f32p, u32p, f32p, i32p, i32p, f32p, optp]
    lib.dexr_retarget_f64.argtypes = [vp, i64, f32p, f32p, f32p, u32p, f64p, i32p, i32p, optp]
    lib.dexr_eval.argtypes = [vp, i64, f32p, f32p, f32p, f64p, u32p, f64p, f64p]
    lib.dexr_fk.argtypes = [vp, i64, f64p, f64p]
    lib.dexr_mano_keypoints_dev.argtypes = [i64, vp, f32p, vp, vp, vp]
    lib.dexr_mano_keypoints.argtypes = [i64, f32p, f32p, f32p, f32p]
    lib.dexr_comm_unique_id.argtypes = [vp]
    lib.dexr_comm_create.argtypes = [vp, C.c_int32, C.c_int32, C.POINTER(vp)]
    lib.dexr_comm_destroy.argtypes = [vp]
    lib.dexr_comm_destroy.restype = None
    lib.dexr_comm_info.argtypes = [vp, i32p, i32p, i32p]
    lib.dexr_allgather.argtypes = [vp, vp, vp, C.c_size_t, vp]
    lib.dexr_comm_max_f64.argtypes = [vp, f64p, C.c_int32, vp]
    lib.dexr_comm_barrier.argtypes = [vp, vp]
    _lib = lib
    return lib


def check(rc: int):
    if rc != 0:
        raise DexrError(f"libdexr error {rc}: {load().dexr_last_error().decode()}")


def default_options(**kw) -> SolveOptions:
    o = SolveOptions()
    load().dexr_default_options(C.byref(o))
    for k, v in kw.items():
        if v is not None:
            setattr(o, k, v)
    return o


def _ptr(a: Optional[np.ndarray], ctype):
    if a is None:
        return None
    return a.ctypes.data_as(C.POINTER(ctype))


class Model:
    """Owns one dexr_model (compiled tables resident in HBM)."""

    def __init__(self, blob: bytes):
        lib = load()
        self._h = C.c_void_p()
        self._one_lock = threading.Lock()
        buf = C.create_string_buffer(blob, len(blob))
        check(lib.dexr_model_create(buf, len(blob), C.byref(self._h)))

    def __del__(self):
        h = getattr(self, "_h", None)
        if h is not None and h.value and _lib is not None:
            _lib.dexr_model_destroy(h)
            self._h = None

    @property
    def handle(self) -> C.c_void_p:
        return self._h

    # launch / damping parameters of this handle (dexr_tuning) -----------------------------------
    def get_tuning(self) -> Tuning:
        t = Tuning()
        t.struct_size = C.sizeof(Tuning)
        check(load().dexr_model_get_tuning(self._h, C.byref(t)))
        return t

    def tune(self, **kw) -> Tuning:
        """Change fields of the handle's dexr_tuning, e.g. ``model.tune(kernel=KERNEL_REGISTER, persist_from=0)``.
        ``lam_jump`` / ``lam_fastdec``: a number is a caller override (holds for every kernel family), ``None`` drops it."""
        t = self.get_tuning()
        for k, v in kw.items():
            if k not in dict(Tuning._fields_):
                raise AttributeError(f"dexr_tuning has no field {k}")
            if k in _TUNE_BITS:  # family-dependent defaults: a value is an explicit override, None returns to the default
                if v is None:
                    t.user_mask &= ~_TUNE_BITS[k]
                    continue
                t.user_mask |= _TUNE_BITS[k]
            setattr(t, k, v)
        check(load().dexr_model_set_tuning(self._h, C.byref(t)))
        return t

    def reserve(self, max_batch: int):
        """Pre-allocate the lazily grown device workspaces for batches of up to `max_batch` frames (dexr_model_reserve):
        afterwards no device-pointer entry point allocates or synchronises with the host."""
        check(load().dexr_model_reserve(self._h, int(max_batch)))

    def kernel(self):
        """(family, bucket, chain) of the float32 solve kernel this handle launches."""
        f, b, c = C.c_int32(), C.c_int32(), C.c_int32()
        check(load().dexr_model_kernel(self._h, C.byref(f), C.byref(b), C.byref(c)))
        return int(f.value), int(b.value), int(c.value)  # chain: 0, 1 (serial-chain kernel) or 2 (with its tip pass)

    def lane_plan(self, comp: int = 0):
        """(n_chain, depth, chain (16,16) uint8, anc_rev (32,) uint32) of the sixteen-lane kernel for one component."""
        n, d = C.c_int32(), C.c_int32()
        chain = np.zeros((16, 16), np.uint8)
        anc = np.zeros(32, np.uint32)
        check(load().dexr_model_lane_plan(self._h, comp, C.byref(n), C.byref(d), chain.ctypes.data_as(C.c_void_p),
                                          anc.ctypes.data_as(C.c_void_p)))
        return int(n.value), int(d.value), chain, anc

    # host-pointer entry points -------------------------------------------------------------------
    def retarget(self, ref, fixed, last, state=None, opts: Optional[SolveOptions] = None, want_info=False,
                 keypoints: bool = False):
        """ref: (B,n_ref,3) ref_value rows, or with keypoints=True the raw (B,n_keypoints,3) hand keypoints."""
        lib = load()
        ref = np.ascontiguousarray(ref, dtype=np.float32)
        last = np.ascontiguousarray(last, dtype=np.float32)
        B = last.shape[0]
        fixed = None if fixed is None or fixed.size == 0 else np.ascontiguousarray(fixed, dtype=np.float32)
        if B == 1:
            return self._retarget_one(lib, ref, fixed, last, state, opts, want_info, keypoints)
        q = np.empty_like(last)
        status = np.zeros(B, dtype=np.int32)
        iters = np.zeros(B, dtype=np.int32)
        fval = np.zeros(B, dtype=np.float32)
        fn = lib.dexr_retarget_kp if keypoints else lib.dexr_retarget
        check(fn(self._h, B, _ptr(ref, C.c_float), _ptr(fixed, C.c_float), _ptr(last, C.c_float),
                 _ptr(state, C.c_uint32), _ptr(q, C.c_float), _ptr(status, C.c_int32),
                 _ptr(iters, C.c_int32), _ptr(fval, C.c_float), C.byref(opts) if opts is not None else None))
        if want_info:
            return q, dict(status=status, iters=iters, fval=fval)
        return q

    def _retarget_one(self, lib, ref, fixed, last, state, opts, want_info, keypoints):
        """ONE frame per call -- the reference's own calling pattern (SeqRetargeting.retarget, profile_online_retargeting.py:
        18-36): the arrays of the call live in per-handle buffers whose ctypes pointers are built once; a call copies a few
        dozen floats in and out instead of allocating four arrays and building eight pointer objects (numpy's
        `ctypes.data_as` costs 1-3 us each: a fifth of a 48 us call)."""
        with self._one_lock:  # (the buffers are shared by every caller of this handle; the C call is serialised per handle anyway)
            return self._retarget_one_locked(lib, ref, fixed, last, state, opts, want_info, keypoints)

    def _retarget_one_locked(self, lib, ref, fixed, last, state, opts, want_info, keypoints):
        key = (ref.shape, None if fixed is None else fixed.shape, last.shape, state is not None)
        c = getattr(self, "_one", None)
        if c is None or c[0] != key:
            b = dict(ref=np.empty(ref.shape, np.float32), fixed=None if fixed is None else np.empty(fixed.shape, np.float32),
                     last=np.empty(last.shape, np.float32), state=None if state is None else np.zeros(1, np.uint32),
                     q=np.empty(last.shape, np.float32), status=np.zeros(1, np.int32), iters=np.zeros(1, np.int32),
                     fval=np.zeros(1, np.float32))
            p = (_ptr(b["ref"], C.c_float), _ptr(b["fixed"], C.c_float), _ptr(b["last"], C.c_float), _ptr(b["state"], C.c_uint32),
                 _ptr(b["q"], C.c_float), _ptr(b["status"], C.c_int32), _ptr(b["iters"], C.c_int32), _ptr(b["fval"], C.c_float))
            c = self._one = (key, b, p)
        b, p = c[1], c[2]
        np.copyto(b["ref"], ref)
        np.copyto(b["last"], last)
        if fixed is not None:
            np.copyto(b["fixed"], fixed)
        if state is not None:
            b["state"][0] = state[0]
        fn = lib.dexr_retarget_kp if keypoints else lib.dexr_retarget
        check(fn(self._h, 1, p[0], p[1], p[2], p[3], p[4], p[5], p[6], p[7], C.byref(opts) if opts is not None else None))
        if state is not None:
            state[0] = b["state"][0]
        q = b["q"].copy()
        if want_info:
            return q, dict(status=b["status"].copy(), iters=b["iters"].copy(), fval=b["fval"].copy())
        return q

    def retarget_f64(self, ref, fixed, last, state=None, opts: Optional[SolveOptions] = None, want_info=False):
        lib = load()
        ref = np.ascontiguousarray(ref, dtype=np.float32)
        last = np.ascontiguousarray(last, dtype=np.float32)
        B = last.shape[0]
        fixed = None if fixed is None or fixed.size == 0 else np.ascontiguousarray(fixed, dtype=np.float32)
        q = np.empty(last.shape, dtype=np.float64)
        status = np.zeros(B, dtype=np.int32)
        iters = np.zeros(B, dtype=np.int32)
        check(lib.dexr_retarget_f64(self._h, B, _ptr(ref, C.c_float), _ptr(fixed, C.c_float), _ptr(last, C.c_float),
                                    _ptr(state, C.c_uint32), _ptr(q, C.c_double), _ptr(status, C.c_int32),
                                    _ptr(iters, C.c_int32), C.byref(opts) if opts is not None else None))
        if want_info:
            return q, dict(status=status, iters=iters)
        return q

    def eval(self, ref, fixed, last, x, state=None):
        lib = load()
        ref = np.ascontiguousarray(ref, dtype=np.float32)
        last = np.ascontiguousarray(last, dtype=np.float32)
        x = np.ascontiguousarray(x, dtype=np.float64)
        B = x.shape[0]
        fixed = None if fixed is None or fixed.size == 0 else np.ascontiguousarray(fixed, dtype=np.float32)
        f = np.zeros(B, dtype=np.float64)
        g = np.zeros_like(x)
        check(lib.dexr_eval(self._h, B, _ptr(ref, C.c_float), _ptr(fixed, C.c_float), _ptr(last, C.c_float),
                            _ptr(x, C.c_double), _ptr(state, C.c_uint32), _ptr(f, C.c_double), _ptr(g, C.c_double)))
        return f, g

    def fk(self, q, n_links: int):
        lib = load()
        q = np.ascontiguousarray(q, dtype=np.float64)
        B = q.shape[0]
        out = np.zeros((B, n_links, 3), dtype=np.float64)
        check(lib.dexr_fk(self._h, B, _ptr(q, C.c_double), _ptr(out, C.c_double)))
        return out

    def retarget_seq_dev(self, B: int, T: int, inputs_ptr: int, fixed_ptr: int, last_ptr: int, state_ptr: int,
                         qraw_ptr: int, status_ptr: int = 0, joint_limit_eps: float = 1e-3,
                         opts: Optional[SolveOptions] = None, stream: int = 0, keypoints: bool = True):
        """T frames of B sequences in one launch (dexr_retarget_seq_dev); device addresses of C-contiguous arrays."""
        check(load().dexr_retarget_seq_dev(self._h, B, T, inputs_ptr or None, 1 if keypoints else 0, fixed_ptr or None,
                                           last_ptr or None, state_ptr or None, qraw_ptr or None, status_ptr or None,
                                           joint_limit_eps, C.byref(opts) if opts is not None else None, stream or None))

    # device-pointer entry point (torch tensors on the current device) ------------------------------
    def retarget_dev(self, B: int, ref_ptr: int, fixed_ptr: int, last_ptr: int, state_ptr: int, q_ptr: int,
                     status_ptr: int = 0, iters_ptr: int = 0, fval_ptr: int = 0,
                     opts: Optional[SolveOptions] = None, stream: int = 0, keypoints: bool = False):
        """Enqueue on `stream`; all pointers are device addresses of C-contiguous arrays.  With keypoints=True
        `ref_ptr` addresses raw (B,n_keypoints,3) hand keypoints instead of ref_value rows."""
        fn = load().dexr_retarget_kp_dev if keypoints else load().dexr_retarget_dev
        check(fn(self._h, B, ref_ptr or None, fixed_ptr or None, last_ptr or None,
                 state_ptr or None, q_ptr or None, status_ptr or None, iters_ptr or None,
                 fval_ptr or None, C.byref(opts) if opts is not None else None, stream or None))


def seq_compose_dev(B: int, T: int, dof_kind, dof_idx, dof_mult, dof_off, n_opt: int, n_fixed: int, qraw_ptr: int,
                    fixed_ptr: int, alpha: float, filter_ptr: int, first_frame_initialises: bool, out_ptr: int,
                    stream: int = 0):
    """robot-qpos composition + mimic fill + low-pass filter for T x B frames (dexr_seq_compose_dev)."""
    kind = np.ascontiguousarray(dof_kind, dtype=np.int32)
    idx = np.ascontiguousarray(dof_idx, dtype=np.int32)
    mult = np.ascontiguousarray(dof_mult, dtype=np.float64)
    off = np.ascontiguousarray(dof_off, dtype=np.float64)
    check(load().dexr_seq_compose_dev(B, T, len(kind), n_opt, n_fixed, _ptr(kind, C.c_int32), _ptr(idx, C.c_int32),
                                      _ptr(mult, C.c_double), _ptr(off, C.c_double), qraw_ptr or None, fixed_ptr or None,
                                      float(alpha), filter_ptr or None, 1 if first_frame_initialises else 0,
                                      out_ptr or None, stream or None))


def fleet_workspace_bytes(B: int) -> int:
    return int(load().dexr_fleet_workspace_bytes(B))


def retarget_multi_dev(models, B: int, model_id_ptr: int, kp_ptr: int, last_ptr: int, ld: int, state_ptr: int,
                       q_ptr: int, status_ptr: int, ws_ptr: int, ws_bytes: int, opts: Optional[SolveOptions] = None,
                       stream: int = 0, fixed_ptr: int = 0, ld_fixed: int = 0):
    """Mixed-fleet batch (dexr_retarget_multi_dev): `models` is a list of Model handles; `fixed_ptr` addresses (B, ld_fixed)
    float32 rows of caller-supplied fixed-joint values (0: no model has any)."""
    arr = (C.c_void_p * len(models))(*[m.handle for m in models])
    check(load().dexr_retarget_multi_dev(arr, len(models), B, model_id_ptr or None, kp_ptr or None, fixed_ptr or None, ld_fixed,
                                         last_ptr or None, ld,
                                         state_ptr or None, q_ptr or None, status_ptr or None,
                                         C.byref(opts) if opts is not None else None, ws_ptr or None, ws_bytes,
                                         stream or None))


def retarget_multi(models, model_id: np.ndarray, keypoints: np.ndarray, last: np.ndarray, state: Optional[np.ndarray] = None,
                   qpos_out: Optional[np.ndarray] = None, opts: Optional[SolveOptions] = None, want_status: bool = False,
                   fixed: Optional[np.ndarray] = None):
    """Mixed-fleet batch on HOST arrays (dexr_retarget_multi): model_id (B,) int32, keypoints (B,21,3) f32, last (B,ld) f32,
    state (B,) uint32 in/out or None.  Returns qpos (B,ld) f32 [, status (B,) int32]; `qpos_out` (in-out) supplies the
    values of the rows / columns the call leaves untouched (default zeros)."""
    mid = np.ascontiguousarray(model_id, dtype=np.int32)
    kp = np.ascontiguousarray(keypoints, dtype=np.float32)
    la = np.ascontiguousarray(last, dtype=np.float32)
    B, ld = la.shape
    if kp.shape != (B, 21, 3) or mid.shape != (B,):
        raise ValueError(f"model_id must be ({B},), keypoints ({B}, 21, 3)")
    q = np.zeros((B, ld), np.float32) if qpos_out is None else np.ascontiguousarray(qpos_out, dtype=np.float32)
    if q.shape != (B, ld):
        raise ValueError(f"qpos_out must have shape ({B}, {ld})")
    if state is not None and (state.dtype != np.uint32 or state.shape != (B,) or not state.flags["C_CONTIGUOUS"]):
        raise ValueError(f"state must be a C-contiguous uint32 array of shape ({B},)")
    status = np.zeros(B, np.int32)
    fx = None if fixed is None else np.ascontiguousarray(fixed, dtype=np.float32).reshape(B, -1)
    arr = (C.c_void_p * len(models))(*[m.handle for m in models])
    check(load().dexr_retarget_multi(arr, len(models), B, _ptr(mid, C.c_int32), _ptr(kp, C.c_float), _ptr(fx, C.c_float),
                                     0 if fx is None else fx.shape[1], _ptr(la, C.c_float), ld,
                                     _ptr(state, C.c_uint32), _ptr(q, C.c_float), _ptr(status, C.c_int32),
                                     C.byref(opts) if opts is not None else None))
    return (q, status) if want_status else q


def comm_unique_id() -> bytes:
    """Rank 0: the RCCL unique id every rank of a communicator has to be handed (dexr_comm_unique_id)."""
    buf = C.create_string_buffer(UNIQUE_ID_BYTES)
    check(load().dexr_comm_unique_id(buf))
    return buf.raw


class Comm:
    """Owns one dexr_comm: an RCCL communicator bound to the current HIP device (collective constructor)."""

    def __init__(self, unique_id: bytes, rank: int, world: int):
        if len(unique_id) != UNIQUE_ID_BYTES:
            raise ValueError(f"unique id must be {UNIQUE_ID_BYTES} bytes")
        self._h = C.c_void_p()
        buf = C.create_string_buffer(unique_id, UNIQUE_ID_BYTES)
        check(load().dexr_comm_create(buf, rank, world, C.byref(self._h)))
        self.rank, self.world = rank, world

    def close(self):
        h = getattr(self, "_h", None)
        if h is not None and h.value and _lib is not None:
            _lib.dexr_comm_destroy(h)
            self._h = None

    __del__ = close

    def rccl_version(self) -> int:
        v = C.c_int32()
        check(load().dexr_comm_info(self._h, None, None, C.byref(v)))
        return int(v.value)

    def allgather(self, send_ptr: int, recv_ptr: int, bytes_per_rank: int, stream: int = 0):
        """recv[r] = rank r's send block; device addresses; enqueued on `stream` (no synchronisation)."""
        check(load().dexr_allgather(self._h, send_ptr or None, recv_ptr or None, bytes_per_rank, stream or None))

    def max_f64(self, values, stream: int = 0) -> np.ndarray:
        v = np.ascontiguousarray(values, dtype=np.float64).copy()
        check(load().dexr_comm_max_f64(self._h, _ptr(v, C.c_double), len(v), stream or None))
        return v

    def barrier(self, stream: int = 0):
        check(load().dexr_comm_barrier(self._h, stream or None))
