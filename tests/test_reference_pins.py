"""Pins to the REFERENCE'S OWN code that go beyond the objective closures: URDF reading / forward kinematics
(yourdfpy.py), SeqRetargeting.warm_start and the per-frame SeqRetargeting bookkeeping.  The fixtures under
tests/golden/ were produced by importing /root/reference (tests/golden/gen_golden.py, oracle/ref_urdf.py,
oracle/ref_harness.py); nothing here reads /root/reference at run time.

CPU tests cover the oracle (oracle/kin.py), the host-side kinematic model (dex_retargeting_amd/urdf.py), the table
compiler (through tests/table_interp.py) and the host mirrors of the reference API; the `-m gpu` tests push the
same fixtures through libdexr's kernels.
"""
import os
import tempfile

import numpy as np
import pytest

from testutil import REPO
from dex_retargeting_amd import model_compiler as mc
from dex_retargeting_amd.constants import DEFAULT_URDF_DIR, HandType
from dex_retargeting_amd.retargeting_config import RetargetingConfig
from dex_retargeting_amd.urdf import KinematicModel, parse_urdf
from oracle import cases
from oracle.kin import OracleRobot
import table_interp as ti

GOLD = os.path.join(REPO, "tests", "golden")
FK = np.load(os.path.join(GOLD, "fk_golden.npz"))
WS = np.load(os.path.join(GOLD, "warm_start_golden.npz"))
SQ = np.load(os.path.join(GOLD, "seq_wrapper_golden.npz"))
FK_KEYS = sorted(k[: -len("__links")] for k in FK.files if k.endswith("__links"))


def _urdf_of(key):
    free = key.endswith("__free")
    base = key[: -len("__free")] if free else key
    if base.startswith("testurdf__"):  # the "messy" URDFs under tests/urdf/
        return os.path.join(REPO, "tests", "urdf", base[len("testurdf__"):] + ".urdf"), free
    return os.path.join(cases.URDF_DIR, base.replace("__", "/") + ".urdf"), free


def _tol(path):
    # rounding level everywhere: the golden FK was run on normalised axes (gen_golden.py explains why), which is also what
    # this repo's reader does (the Shadow fixture carries one axis that is unit only to 6 digits)
    return 1e-12


def _q_by_name(dof_names, mimic, key, c):
    """golden configuration c (over the reference's actuated joints) -> full q in `dof_names` order, mimic joints
    filled the way the reference's FK fills them (yourdfpy.py:1017-1023)."""
    val = dict(zip(FK[key + "__joints"].tolist(), FK[key + "__cfg"][c].tolist()))
    mims = FK[key + "__mimic"].tolist()
    if mims != [""]:
        for n, s, a, b in zip(mims, FK[key + "__mimic_src"].tolist(), FK[key + "__mimic_mult"], FK[key + "__mimic_off"]):
            val[n] = val[s] * float(a) + float(b)
    return np.array([val[n] for n in dof_names])


def test_fk_golden_covers_every_fixture_urdf():
    import glob

    urdfs = glob.glob(os.path.join(cases.URDF_DIR, "*", "*.urdf"))
    messy = [k for k in FK_KEYS if k.startswith("testurdf__")]
    assert len(FK_KEYS) == 2 * len(urdfs) + len(messy) and len(urdfs) >= 13 and len(messy) == 2


@pytest.mark.parametrize("key", FK_KEYS)
def test_oracle_fk_equals_reference_fk(key):
    path, free = _urdf_of(key)
    r = OracleRobot(path, free)
    links = FK[key + "__links"].tolist()
    assert sorted(links) == sorted(r.links)
    # the reference's actuated joints (non-fixed, non-mimic) + its mimic joints == this model's dof joints
    mims = [m for m in FK[key + "__mimic"].tolist() if m]
    assert sorted(FK[key + "__joints"].tolist() + mims) == sorted(r.dof_joint_names)
    lim = dict(zip(FK[key + "__joints"].tolist(), zip(FK[key + "__lower"], FK[key + "__upper"])))
    for n, (lo, hi) in lim.items():
        assert np.allclose(r.joint_limits[r.qidx[n]], [lo, hi], atol=1e-12)
    if mims:  # mimic parameters as the reference parsed them (yourdfpy.py:1107-1115)
        got = {m[0]: m[1:] for m in r.mimic}
        for n, s, a, b in zip(mims, FK[key + "__mimic_src"].tolist(), FK[key + "__mimic_mult"], FK[key + "__mimic_off"]):
            assert got[n][0] == s and abs(got[n][1] - a) < 1e-12 and abs(got[n][2] - b) < 1e-12
    for c in range(FK[key + "__cfg"].shape[0]):
        q = _q_by_name(r.dof_joint_names, r.mimic, key, c)
        R, p = r.link_poses(q[None], links)
        T = FK[key + "__T"][c]
        assert np.abs(R[0] - T[:, :3, :3]).max() < _tol(path), key
        assert np.abs(p[0] - T[:, :3, 3]).max() < _tol(path), key


@pytest.mark.parametrize("key", FK_KEYS)
def test_host_model_and_compiled_tables_equal_reference_fk(key):
    path, free = _urdf_of(key)
    km = KinematicModel(parse_urdf(path, add_dummy_free_joints=free))
    links = FK[key + "__links"].tolist()
    comp = mc.compile_fk(km, links)
    for c in range(FK[key + "__cfg"].shape[0]):
        q = _q_by_name(km.dof_joint_names, None, key, c)
        T = FK[key + "__T"][c]
        # host float64 model (poses incl. rotation: used by warm_start and the local Jacobian)
        for li, name in enumerate(links):
            Th, _ = km.frame_pose_and_local_jacobian(q, km.body_frame_index(name))
            assert np.abs(Th - T[li]).max() < _tol(path), (key, name)
        # compiled float32 tables, evaluated by the test interpreter (the GPU test runs the same tables in dexr_fk)
        got = np.zeros((len(links), 3))
        for comp_rec in comp.comps:
            qj = ti.joint_values(comp_rec, q_full=q[None])
            P, _, _ = ti.frame_positions(comp_rec, qj)
            for t in range(int(comp_rec["n_term"])):
                got[int(comp_rec["term_ref"][t])] = P[0, int(comp_rec["term_task"][t])]
        assert np.abs(got - T[:, :3, 3]).max() < 2e-6, key  # float32 table entries


@pytest.mark.parametrize("key", FK_KEYS)
def test_rewritten_urdf_parses_to_the_same_model(key):
    """RetargetingConfig.build hands pinocchio the file yourdfpy WRITES (retargeting_config.py:176-186): origins go
    through euler_from_matrix and back, mimic tags are dropped (yourdfpy.py:1787-1802), dummy joints are materialised.
    Our reader must turn that text into the same kinematic model as the original file (+ add_dummy_free_joints)."""
    path, free = _urdf_of(key)
    a = KinematicModel(parse_urdf(path, add_dummy_free_joints=free))
    with tempfile.NamedTemporaryFile("w", suffix=".urdf", delete=False) as f:
        f.write(str(FK[key + "__rewritten"]))
        tmp = f.name
    try:
        b = KinematicModel(parse_urdf(tmp, add_dummy_free_joints=False))
    finally:
        os.remove(tmp)
    assert a.dof_joint_names == b.dof_joint_names
    assert a.frame_names == b.frame_names
    assert np.allclose(a.joint_limits, b.joint_limits, atol=1e-9)
    for ja, jb in zip(a.joints, b.joints):
        assert ja.type == jb.type and ja.parent == jb.parent
        assert np.abs(ja.placement - jb.placement).max() < 1e-9 and np.abs(ja.axis - jb.axis).max() < 1e-6
    for fa, fb in zip(a.frames, b.frames):
        assert fa.parent == fb.parent and np.abs(fa.placement - fb.placement).max() < 1e-9
    assert b.mimic_joints()[0] == []  # the written file carries no <mimic> (the reference re-reads them from the original)


def test_messy_urdf_constructs_are_read_like_the_reference_reads_them():
    """tests/urdf/messy_arm_hand.urdf: inertial / visual / collision / material / transmission / gazebo noise, non-unit and
    negative axes, missing <origin> / <axis>, a mimic joint declared before its source, fixed-joint chains between
    movable joints, sibling joints whose file order is not their name order.  Axes and mimic parameters as the
    reference's reader parsed them (golden), dof order = depth-first with siblings by joint name (pinocchio / urdfdom's
    std::map), fixed joints folded."""
    key = "testurdf__messy_arm_hand"
    path, _ = _urdf_of(key)
    robot = parse_urdf(path)
    jm = robot.joint_map
    for n, a in zip(FK[key + "__axis_names"].tolist(), FK[key + "__axis_raw"]):
        assert np.array_equal(jm[n].axis, a), n  # as written (0 0 2, 3 0 4, 1 1 0, default 1 0 0)
    km = KinematicModel(robot)
    assert km.dof_joint_names == ["arm_joint_1", "arm_joint_10", "arm_joint_2", "wrist_roll", "finger_A_joint_1",
                                  "finger_A_joint_2", "finger_B_joint_1", "finger_B_joint_2"]
    for j in km.joints:
        assert abs(np.linalg.norm(j.axis) - 1.0) < 1e-15
    assert km.mimic_joints() == (["finger_B_joint_1"], ["finger_B_joint_2"], [0.8], [0.05])
    # every link is a BODY frame; links behind fixed joints hang off the last movable joint above them
    byname = {f.name: f for f in km.frames}
    assert byname["camera_optical"].parent == km.dof_joint_names.index("wrist_roll")
    assert byname["finger_A.tip"].parent == km.dof_joint_names.index("finger_A_joint_2")
    assert byname["mount_plate"].parent == -1 and byname["adapter"].parent == km.dof_joint_names.index("arm_joint_2")


def test_continuous_joint_is_rejected_like_the_reference_rejects_it():
    """robot_wrapper.py:22-23: pinocchio gives a continuous joint nq = 2 != nv = 1 -> NotImplementedError."""
    from dex_retargeting_amd.robot_wrapper import RobotWrapper

    path = os.path.join(REPO, "tests", "urdf", "messy_continuous.urdf")
    with pytest.raises(NotImplementedError, match="Can not handle robot with special joint."):
        KinematicModel(parse_urdf(path))
    with pytest.raises(NotImplementedError, match="Can not handle robot with special joint."):
        RobotWrapper(path)


# ---- SeqRetargeting.warm_start (seq_retarget.py:45-110) ---------------------------------------------------------
WS_KEYS = sorted(k[: -len("__pos")] for k in WS.files if k.endswith("__pos"))


def _build(rel):
    RetargetingConfig.set_default_urdf_dir(str(DEFAULT_URDF_DIR))
    return RetargetingConfig.load_from_file(os.path.join(cases.CONFIG_DIR, rel)).build()


@pytest.mark.parametrize("key", WS_KEYS)
def test_warm_start_equals_reference(key, monkeypatch):
    from dex_retargeting_amd import robot_wrapper
    from dex_retargeting_amd.seq_retarget import BatchedSeqRetargeting

    # warm_start is host arithmetic; the link pose it needs comes from the host float64 model when no GPU is present
    def host_pose(self, link_id):
        T, _ = self.kin.frame_pose_and_local_jacobian(self._qpos[0], self.kin.body_of_frame_id(link_id))
        return T

    from testutil import gpu_available
    if not gpu_available():
        monkeypatch.setattr(robot_wrapper.RobotWrapper, "get_link_pose", host_pose)
    rel = key.replace("__", "/") + ".yml"
    seq = _build(rel)
    assert seq.optimizer.target_joint_names == WS[key + "__target_joint_names"].tolist()
    pos, quat = WS[key + "__pos"], WS[key + "__quat"]
    right, mano, want = WS[key + "__hand_is_right"], WS[key + "__mano"], WS[key + "__last_qpos"]
    for i in range(pos.shape[0]):
        seq.reset()
        seq.warm_start(pos[i], quat[i], HandType.right if right[i] else HandType.left, bool(mano[i]))
        assert seq.last_qpos.dtype == np.float32
        assert np.abs(seq.last_qpos - want[i]).max() < 2e-6, (key, i)  # float32 storage of a float64 pose
    # batched form, one call per (hand type, convention) group, no per-item Python loop inside
    bs = BatchedSeqRetargeting(seq.optimizer, pos.shape[0])
    for hr in (True, False):
        for mn in (True, False):
            sel = np.nonzero((right == hr) & (mano == mn))[0]
            if not len(sel):
                continue
            b2 = BatchedSeqRetargeting(seq.optimizer, len(sel))
            b2.warm_start(pos[sel], quat[sel], HandType.right if hr else HandType.left, mn)
            assert np.abs(b2.last_qpos - want[sel]).max() < 2e-6
    assert bs.last_qpos.shape == want.shape


# ---- SeqRetargeting.retarget bookkeeping (seq_retarget.py:112-134) around a replaying stub ------------------------
SQ_KEYS = sorted(k[: -len("__answers")] for k in SQ.files if k.endswith("__answers"))


class _Replay:
    def __init__(self, base, answers):
        self.__dict__.update(_base=base, answers=answers, calls=[])

    def __getattr__(self, name):
        return getattr(self._base, name)

    def retarget(self, ref_value, fixed_qpos, last_qpos):
        self.calls.append(np.array(last_qpos))
        return self.answers[len(self.calls) - 1].astype(np.float32)


@pytest.mark.parametrize("key", SQ_KEYS)
def test_seq_retargeting_bookkeeping_equals_reference(key):
    from dex_retargeting_amd.optimizer_utils import LPFilter
    from dex_retargeting_amd.seq_retarget import SeqRetargeting

    rel = key.replace("__", "/") + ".yml"
    built = _build(rel)
    assert built.joint_names == SQ[key + "__joint_names"].tolist()  # pinocchio-order names ([not-in-ref] DFS order)
    assert built.optimizer.target_joint_names == SQ[key + "__target_joint_names"].tolist()
    ans = SQ[key + "__answers"]
    stub = _Replay(built.optimizer, ans)
    alpha = float(SQ[key + "__alpha"])
    seq = SeqRetargeting(stub, has_joint_limits=True, lp_filter=LPFilter(alpha) if 0 <= alpha <= 1 else None)
    n_ref = int(built.optimizer.compiled_model().n_ref)
    out = np.array([seq.retarget(np.zeros((n_ref, 3)), fixed_qpos=np.zeros(len(built.optimizer.idx_pin2fixed)))
                    for _ in range(ans.shape[0])])
    assert np.abs(out - SQ[key + "__robot_qpos"]).max() < 1e-12
    assert np.abs(np.array(stub.calls) - SQ[key + "__last_given"]).max() < 1e-12


# ---- the same FK fixtures through the HIP kernel ---------------------------------------------------------------
@pytest.mark.gpu
@pytest.mark.parametrize("key", FK_KEYS)
def test_gpu_fk_equals_reference_fk(key, require_gpu):
    from dex_retargeting_amd.robot_wrapper import RobotWrapper

    path, free = _urdf_of(key)
    robot = RobotWrapper(path, add_dummy_free_joints=free)
    links = FK[key + "__links"].tolist()
    q = np.stack([_q_by_name(robot.dof_joint_names, None, key, c) for c in range(FK[key + "__cfg"].shape[0])])
    got = robot.link_positions(q, [robot.get_link_index(n) for n in links])
    want = FK[key + "__T"][:, :, :3, 3]
    assert np.abs(got - want).max() < 2e-6, key  # float64 kernel over float32 table entries
    robot.compute_forward_kinematics(q[2])
    T = robot.get_link_pose(robot.get_link_index(links[-1]))
    assert np.abs(T - FK[key + "__T"][2, -1]).max() < 2e-6
