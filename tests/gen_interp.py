"""Test helper: a numpy reader + forward-kinematics interpreter of the GENERIC table format (include/dexr_tables.h,
dex_retargeting_amd/generic_tables.py), written from the header's layout description -- so the CPU suite can check what
the table compiler emits for models that only the general kernel serves, without a GPU."""
import numpy as np

from dex_retargeting_amd import generic_tables as gt
from dex_retargeting_amd import model_compiler as mc


def parse(blob: bytes) -> dict:
    hsz = mc.HEADER_DTYPE.itemsize
    header = np.frombuffer(blob[:hsz], dtype=mc.HEADER_DTYPE)[0]
    assert int(header["n_comp"]) == 0 and int(header["comp_bytes"]) == 0
    body = blob[hsz:]
    gh = np.frombuffer(body[:gt.GEN_HEADER_DTYPE.itemsize], dtype=gt.GEN_HEADER_DTYPE)[0]
    assert int(gh["magic"]) == gt.GEN_MAGIC
    nj, nf, nt, nv, nfam = (int(gh[k]) for k in ("n_joint", "n_frame", "n_term", "n_var", "n_fam"))
    off = [gt.GEN_HEADER_DTYPE.itemsize]

    def take(dtype, n, shape=None):
        a = np.frombuffer(body, dtype=dtype, count=n, offset=off[0])
        off[0] += n * np.dtype(dtype).itemsize
        return a.reshape(shape) if shape else a

    def take_i(n):
        a = take("<i4", n + (n & 1))
        return a[:n]

    t = dict(header=header, gh=gh, nj=nj, nf=nf, nt=nt, nv=nv)
    t["X"] = take("<f8", nj * 12, (nj, 12))
    t["axis"] = take("<f8", nj * 3, (nj, 3))
    t["jmul"], t["joff"] = take("<f8", nj), take("<f8", nj)
    t["lo"], t["hi"] = take("<f8", nv), take("<f8", nv)
    t["frame_off"] = take("<f8", nf * 3, (nf, 3))
    t["frame_anc"], t["joint_anc"] = take("<u8", nf), take("<u8", nj)
    for name, n in (("jtype", nj), ("parent", nj), ("depth", nj), ("src_idx", nj), ("var", nj), ("var_api", nv),
                    ("fam_off", nv + 1), ("fam", nfam), ("frame_joint", nf), ("term_task", nt), ("term_origin", nt),
                    ("term_ref", nt), ("row_ho", nt), ("row_ht", nt)):
        t[name] = take_i(n)
    assert off[0] == len(body)
    return t


def joint_values(t: dict, x=None, fixed=None, q_full=None) -> np.ndarray:
    q = np.zeros(t["nj"])
    for k in range(t["nj"]):
        if q_full is not None:
            q[k] = q_full[t["src_idx"][k]]
        elif t["var"][k] >= 0:
            q[k] = t["jmul"][k] * x[t["var"][k]] + t["joff"][k]
        else:
            q[k] = t["jmul"][k] * fixed[t["src_idx"][k]] + t["joff"][k]
    return q


def frame_positions(t: dict, q: np.ndarray) -> np.ndarray:
    nj = t["nj"]
    R, p = np.zeros((nj, 3, 3)), np.zeros((nj, 3))
    for k in range(nj):  # parents come first
        pa = int(t["parent"][k])
        Rp, pp = (R[pa], p[pa]) if pa >= 0 else (np.eye(3), np.zeros(3))
        Xr, Xp = t["X"][k, :9].reshape(3, 3), t["X"][k, 9:]
        Ra, pa3 = Rp @ Xr, Rp @ Xp + pp
        a = t["axis"][k]
        if t["jtype"][k] == 0:
            K = np.array([[0, -a[2], a[1]], [a[2], 0, -a[0]], [-a[1], a[0], 0]])
            R[k] = Ra @ (np.eye(3) + np.sin(q[k]) * K + (1 - np.cos(q[k])) * (K @ K))
            p[k] = pa3
        else:
            R[k], p[k] = Ra, pa3 + Ra @ a * q[k]
    out = np.zeros((t["nf"], 3))
    for f in range(t["nf"]):
        j = int(t["frame_joint"][f])
        out[f] = t["frame_off"][f] if j < 0 else R[j] @ t["frame_off"][f] + p[j]
    return out
