"""GPU tests (-m gpu) of the GENERAL kernel (csrc/dexr_gen.hpp) and the generic table format:

* forced onto the shipped robots it must reproduce the reference's closures (golden vectors), the reference's FK, the
  float64 oracle minimiser (1e-4 rad, BASELINE.json north_star) and the specialised kernels' answers;
* models that only it can serve -- an arm + Shadow hand with 37 movable joints (position, 21 reference rows) and a
  20-vector problem on the same robot -- are solved to the oracle's minimum, through every entry point (host pointers,
  keypoint input, sequences), instead of raising ValueError.
"""
import os

import numpy as np
import pytest

from dex_retargeting_amd import _lib
from dex_retargeting_amd.constants import DEFAULT_URDF_DIR
from dex_retargeting_amd.retargeting_config import RetargetingConfig
from oracle import cases, solvers
from oracle.kin import OracleRobot
from oracle.objectives import OracleProblem
from test_generic_tables import ARM_HAND, arm_hand_config, comb_hand_config, comb_hand_urdf

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(__file__), "golden")
RetargetingConfig.set_default_urdf_dir(str(DEFAULT_URDF_DIR))


@pytest.fixture(scope="module", autouse=True)
def _gpu(require_gpu):
    yield


def build_generic(rel):
    seq = RetargetingConfig.load_from_file(os.path.join(cases.CONFIG_DIR, rel)).build()
    seq.optimizer.use_generic_tables = True
    assert seq.optimizer.device_model().kernel()[0] == _lib.KERNEL_GENERAL
    return seq, cases.problem_from_config(rel)


def dexpilot_kw(prob, ref):
    if prob.kind != "dexpilot":
        return {}
    w, rv, _ = prob.dexpilot_preamble(ref, np.zeros((ref.shape[0], prob.n_pair), bool))
    return dict(weights=w, dexpilot_ref=rv)


GOLD_CONFIGS = ["teleop/allegro_hand_right.yml", "teleop/shadow_hand_right_dexpilot.yml", "offline/leap_hand_right.yml",
                "teleop/ability_hand_right.yml", "teleop/inspire_hand_right_dexpilot.yml", "offline/schunk_svh_hand_right.yml",
                "teleop/panda_gripper.yml"]


@pytest.mark.parametrize("rel", GOLD_CONFIGS)
def test_general_kernel_objective_matches_reference_golden(rel):
    g = np.load(os.path.join(GOLD, "objective_golden.npz"))
    k = rel.replace("/", "__").replace(".yml", "")
    seq, prob = build_generic(rel)
    model = seq.optimizer.device_model()
    ref, fixed, last, x = g[k + "__ref"], g[k + "__fixed"], g[k + "__last"], g[k + "__x"]
    st = None
    if prob.kind == "dexpilot":
        proj = g[k + "__state_in"]
        st = (proj.astype(np.uint32) << np.arange(proj.shape[1], dtype=np.uint32)).sum(1).astype(np.uint32)
    f, grad = model.eval(ref, fixed, last, x, state=st)
    assert np.allclose(f, g[k + "__f"], rtol=2e-6, atol=1e-9)
    assert np.allclose(grad, g[k + "__grad"], rtol=2e-6, atol=2e-8)  # float32 ref rows / DexPilot targets as the reference
    if st is not None:
        so = g[k + "__state_out"]
        assert np.array_equal(st, (so.astype(np.uint32) << np.arange(so.shape[1], dtype=np.uint32)).sum(1).astype(np.uint32))


@pytest.mark.parametrize("rel", GOLD_CONFIGS)
def test_general_kernel_solves_to_the_oracle_minimum_and_agrees_with_the_specialised_kernels(rel):
    seq, prob = build_generic(rel)
    B = 96
    d = cases.reachable_set(prob, B, 0.05)
    st = np.zeros(B, np.uint32) if prob.kind == "dexpilot" else None
    q = seq.optimizer.retarget_batch(d["ref"], d["fixed"], d["last"], state=st)
    want = solvers.solve_lm_batched(prob, d["ref"], d["fixed"], d["last"], newton=True, max_iter=100, **dexpilot_kw(prob, d["ref"]))
    assert np.abs(q - want).max() < 1e-4, rel
    assert np.all(seq.optimizer.last_info["status"] == 0)
    fast = RetargetingConfig.load_from_file(os.path.join(cases.CONFIG_DIR, rel)).build()
    st2 = np.zeros(B, np.uint32) if prob.kind == "dexpilot" else None
    q2 = fast.optimizer.retarget_batch(d["ref"], d["fixed"], d["last"], state=st2)
    assert np.abs(q - q2).max() < 1e-4
    if st is not None:
        assert np.array_equal(st, st2)


def arm_hand(kind):
    cfg = arm_hand_config(kind)
    seq = RetargetingConfig.from_dict(cfg).build()
    free = kind == "position"
    r = OracleRobot(ARM_HAND, add_dummy_free_joints=free)
    if free:
        prob = OracleProblem(r, "position", None, target_link_names=cfg["target_link_names"])
        prob.target_link_human_indices = np.arange(21)
    else:
        prob = OracleProblem(r, "vector", None, target_origin_link_names=cfg["target_origin_link_names"],
                             target_task_link_names=cfg["target_task_link_names"], scaling=cfg["scaling_factor"])
        prob.target_link_human_indices = np.array(cfg["target_link_human_indices"])
    assert list(prob.idx_pin2target) == list(seq.optimizer.idx_pin2target)
    return seq, prob


@pytest.mark.parametrize("kind", ["position", "vector"])
def test_arm_plus_hand_model_is_solved_not_refused(kind):
    """31 movable joints (+ 6 free joints) in one component and 20 / 21 reference rows: beyond every fixed-size table
    (DEXR_MAXJ = 32, DEXR_MAXT = 16).  The reference accepts such a model (optimizer.py:18-52); so does this library."""
    seq, prob = arm_hand(kind)
    opt = seq.optimizer
    assert opt.device_model().kernel()[0] == _lib.KERNEL_GENERAL
    B = 64
    d = cases.reachable_set(prob, B, 0.03)
    q, info = opt.device_model().retarget(d["ref"], None, d["last"], want_info=True)
    want = solvers.solve_lm_batched(prob, d["ref"], None, d["last"], newton=True, max_iter=100)
    dq = np.abs(q - want).max(1)
    F_got = prob.total(q.astype(np.float64), d["ref"], None, d["last"].astype(np.float64))
    F_want = prob.total(want, d["ref"], None, d["last"].astype(np.float64))
    # 37 coupled variables: a few frames may settle in another (certified not worse) minimum than the oracle's LM
    far = dq >= 1e-4
    assert far.mean() <= 0.05 and np.all(F_got[far] <= F_want[far] + 1e-9), (kind, dq.max())
    assert np.all(info["status"] <= 1)
    # objective closure through the same tables == the oracle's closure
    f, g = opt.device_model().eval(d["ref"][:8], None, d["last"][:8], want[:8])
    fo, go, _ = prob.evaluate(want[:8], d["ref"][:8], None, d["last"][:8])
    # (the header carries huber_delta / inv_norm / norm_delta as float32: 6e-8 relative, like every other kernel)
    assert np.allclose(f, fo, rtol=1e-6, atol=1e-12) and np.allclose(g, go, rtol=1e-6, atol=1e-9)
    # single-frame API + raw keypoint input
    kp = cases.human_keypoints(4)
    ref = cases.ref_from_keypoints(prob, kp).astype(np.float32)
    last = np.repeat(seq.last_qpos[None], 4, 0).astype(np.float32)
    a = opt.retarget_batch(ref, None, last)
    b = opt.retarget_keypoints_batch(kp, None, last)
    assert np.array_equal(a, b)
    out = seq.retarget(ref[0])
    assert out.shape == (opt.robot.dof,) and np.all(np.isfinite(out))


def test_forward_kinematics_of_a_37_joint_robot():
    from dex_retargeting_amd.robot_wrapper import RobotWrapper

    robot = RobotWrapper(ARM_HAND, add_dummy_free_joints=True)
    r = OracleRobot(ARM_HAND, add_dummy_free_joints=True)
    names = [f.name for f in robot.kin.frames]  # every link: more than one generic FK table holds -> chunked
    rng = np.random.default_rng(2)
    q = rng.uniform(r.joint_limits[:, 0], r.joint_limits[:, 1], size=(5, r.dof))
    got = robot.link_positions(q, [robot.get_link_index(n) for n in names])
    want = r.link_positions(q, names)
    # chunks whose chains fit the fixed-size records run on float32 table entries (2e-6, as in test_reference_pins.py);
    # the finger chains (37 joints above them) go through the generic float64 table
    assert np.abs(got - want).max() < 2e-6
    tips = ["thtip", "fftip", "mftip", "rftip", "lftip"]
    got_t = robot.link_positions(q, [robot.get_link_index(n) for n in tips])
    assert np.abs(got_t - r.link_positions(q, tips)).max() < 2e-6


def test_general_kernel_sequence_mode_equals_frame_by_frame():
    """dexr_retarget_seq_dev on a generic model: T frames per sequence inside one launch, carrying the clipped float32
    answer (seq_retarget.py:118-124) == T single-frame calls."""
    torch = pytest.importorskip("torch")
    seq, prob = arm_hand("vector")
    model = seq.optimizer.device_model()
    B, T = 6, 5
    kp = np.stack([cases.human_keypoints(B, seed=40 + t) for t in range(T)])
    lim = seq.joint_limits
    last = np.repeat(lim.mean(1)[None], B, 0).astype(np.float32)
    cur = last.copy()
    want = []
    for t in range(T):
        cur = np.clip(cur, lim[:, 0], lim[:, 1]).astype(np.float32)
        cur = model.retarget(kp[t], None, cur, keypoints=True)
        want.append(cur.copy())
    dev = torch.device("cuda:0")
    t_kp, t_last = torch.from_numpy(kp).to(dev), torch.from_numpy(last).to(dev)
    t_raw = torch.empty((T, B, prob.n_opt), dtype=torch.float32, device=dev)
    model.retarget_seq_dev(B, T, t_kp.data_ptr(), 0, t_last.data_ptr(), 0, t_raw.data_ptr(),
                           stream=torch.cuda.current_stream().cuda_stream)
    torch.cuda.synchronize()
    assert np.abs(t_raw.cpu().numpy() - np.stack(want)).max() < 2e-6


@pytest.mark.parametrize("kind", ["position", "vector"])
@pytest.mark.parametrize("fingers,joints", [(6, 7), (7, 7), (8, 7), (10, 6)])
def test_models_of_46_to_64_variables_run_on_the_64_variable_instantiation(tmp_path, kind, fingers, joints):
    """A 4-joint wrist + `fingers` chains of `joints` revolute joints: 46 / 53 / 60 / 64 variables in one component -- the
    general kernel's second instantiation (8 x 8 lane grid with 6, 7 or 8 tile rows: up to 36 Hessian entries per lane in
    registers, 64 rows for the register factorisation).  Solved to the oracle's minimum from near starts; objective
    closure == the oracle's."""
    urdf = comb_hand_urdf(str(tmp_path / "comb_hand.urdf"), fingers=fingers, joints=joints)
    cfg = comb_hand_config(urdf, kind, fingers=fingers, joints=joints)
    seq = RetargetingConfig.from_dict(cfg).build()
    opt = seq.optimizer
    r = OracleRobot(urdf)
    if kind == "position":
        prob = OracleProblem(r, "position", None, target_link_names=cfg["target_link_names"])
        prob.target_link_human_indices = np.arange(2 * fingers)
    else:
        prob = OracleProblem(r, "vector", None, target_origin_link_names=cfg["target_origin_link_names"],
                             target_task_link_names=cfg["target_task_link_names"], scaling=cfg["scaling_factor"])
        prob.target_link_human_indices = np.array(cfg["target_link_human_indices"])
    assert opt.device_model().kernel()[0] == _lib.KERNEL_GENERAL and opt.opt_dof == 4 + fingers * joints
    B = 64
    d = cases.reachable_set(prob, B, 0.03)
    q, info = opt.device_model().retarget(d["ref"], None, d["last"], want_info=True)
    want = solvers.solve_lm_batched(prob, d["ref"], None, d["last"], newton=True, max_iter=100)
    l64 = d["last"].astype(np.float64)
    dq = np.abs(q - want).max(1)
    F_got, F_want = prob.total(q.astype(np.float64), d["ref"], None, l64), prob.total(want, d["ref"], None, l64)
    far = dq >= 1e-4
    assert far.mean() <= 0.05 and np.all(F_got[far] <= F_want[far] + 1e-9), (kind, dq.max(), int(far.sum()))
    assert np.all(info["status"] <= 1)
    f, g = opt.device_model().eval(d["ref"][:8], None, d["last"][:8], want[:8])
    fo, go, _ = prob.evaluate(want[:8], d["ref"][:8], None, d["last"][:8])
    assert np.allclose(f, fo, rtol=1e-6, atol=1e-12) and np.allclose(g, go, rtol=1e-6, atol=1e-9)
