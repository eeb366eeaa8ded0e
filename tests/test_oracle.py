"""CPU tests that pin the ORACLE before anything is compared with it (no GPU).

* objective value/gradient vs the golden vectors produced by the reference's own closures
  (tests/golden/objective_golden.npz, see tests/golden/gen_golden.py);
* the two docstring known-answers the reference holds (optimizer.py:409-413, 432-439);
* FK restatement: finite differences + hand-derived poses (parity with pinocchio itself is UNPINNED: it is not
  installed and the reference holds no FK vectors);
* solvers: batched LM == scipy tight minimiser; the reference's round-trip property
  (/root/reference/tests/test_optimizer.py:141,209,278: mean error < 1e-2 m) for the as-configured SLSQP path.
"""
import json
import os
import sys

import numpy as np
import pytest

from oracle import cases, solvers
from oracle.kin import OracleRobot
from oracle.objectives import dexpilot_cache, generate_link_indices

GOLD = os.path.join(os.path.dirname(__file__), "golden")
OBJ_CONFIGS = [
    "teleop/allegro_hand_right.yml", "teleop/shadow_hand_right_dexpilot.yml", "offline/leap_hand_right.yml",
    "teleop/ability_hand_right.yml", "teleop/inspire_hand_right_dexpilot.yml", "offline/schunk_svh_hand_right.yml",
    "teleop/panda_gripper.yml", "teleop/shadow_hand_left.yml", "teleop/allegro_hand_left_dexpilot.yml",
]


def _key(rel):
    return rel.replace("/", "__").replace(".yml", "")


def test_docstring_known_answers():
    assert generate_link_indices(4) == ([2, 3, 4, 3, 4, 4, 0, 0, 0, 0], [1, 1, 1, 2, 2, 3, 1, 2, 3, 4])
    n_pair, s2o, s2t, dist = dexpilot_cache(4, 0.1, 0.2)
    assert n_pair == 6 and s2o == [1, 2, 2] and s2t == [0, 0, 1]
    assert np.allclose(dist, [0.1, 0.1, 0.1, 0.2, 0.2, 0.2])


@pytest.mark.parametrize("rel", OBJ_CONFIGS)
def test_objective_matches_reference_closures(rel):
    g = np.load(os.path.join(GOLD, "objective_golden.npz"))
    k = _key(rel)
    prob = cases.problem_from_config(rel)
    ref, fixed, last, x = g[k + "__ref"], g[k + "__fixed"], g[k + "__last"], g[k + "__x"]
    kw = {}
    if prob.kind == "dexpilot":
        w, rv, st = prob.dexpilot_preamble(ref, g[k + "__state_in"])
        assert np.array_equal(st, g[k + "__state_out"])
        kw = dict(weights=w, dexpilot_ref=rv)
    f, grad, _ = prob.evaluate(x, ref, fixed, last, **kw)
    assert np.allclose(f, g[k + "__f"], rtol=1e-12, atol=1e-14)
    assert np.allclose(grad, g[k + "__grad"], rtol=1e-10, atol=1e-13)


def test_dexpilot_projection_fires_in_golden():
    g = np.load(os.path.join(GOLD, "objective_golden.npz"))
    k = _key("teleop/shadow_hand_right_dexpilot.yml")
    so, si = g[k + "__state_out"], g[k + "__state_in"]
    assert so.any() and not so.all() and (so != si).any()  # set, cleared and hysteresis branches are all exercised


def test_fk_hand_derived_poses():
    r = OracleRobot(os.path.join(cases.URDF_DIR, "allegro_hand/allegro_hand_right.urdf"))
    # zero pose: the middle finger is a straight stack of its link offsets above the wrist->base offset
    z = 0.095 + 0.0007 + 0.0164 + 0.054 + 0.0384 + 0.0267
    p = r.link_positions(np.zeros((1, 16)), ["link_7.0_tip", "wrist", "base_link"])[0]
    assert np.allclose(p[0], [0, 0, z], atol=1e-12)
    assert np.allclose(p[1], 0) and np.allclose(p[2], [0, 0, 0.095])
    # bend joint_5.0 (axis +y at height 0.095+0.0007+0.0164) by 90 deg: the rest of the finger points along +x
    q = np.zeros((1, 16))
    q[0, r.qidx["joint_5.0"]] = np.pi / 2
    p = r.link_positions(q, ["link_7.0_tip"])[0, 0]
    assert np.allclose(p, [0.054 + 0.0384 + 0.0267, 0, 0.095 + 0.0007 + 0.0164], atol=1e-12)
    # panda: prismatic fingers move along +-y
    pr = OracleRobot(os.path.join(cases.URDF_DIR, "panda_gripper/panda_gripper_glb.urdf"))
    p = pr.link_positions(np.array([[0.03, 0.01]]), ["panda_leftfinger", "panda_rightfinger"])[0]
    assert np.allclose(p, [[0, 0.03, 0.0584], [0, -0.01, 0.0584]])


def test_pinocchio_joint_order_is_lexicographic_dfs():
    r = OracleRobot(os.path.join(cases.URDF_DIR, "allegro_hand/allegro_hand_right.urdf"))
    want = [f"joint_{i}.0" for i in (0, 1, 2, 3, 12, 13, 14, 15, 4, 5, 6, 7, 8, 9, 10, 11)]
    assert r.dof_joint_names == want
    rf = OracleRobot(os.path.join(cases.URDF_DIR, "leap_hand/leap_hand_right.urdf"), add_dummy_free_joints=True)
    assert rf.dof == 22 and all("dummy" in n for n in rf.dof_joint_names[:6])
    assert np.allclose(rf.joint_limits[:3], [[-5, 5]] * 3) and np.allclose(rf.joint_limits[3:6], [[-2 * np.pi, 2 * np.pi]] * 3)


@pytest.mark.parametrize("urdf,free", [("shadow_hand/shadow_hand_right.urdf", False),
                                       ("schunk_hand/schunk_svh_hand_left.urdf", True),
                                       ("leap_hand/leap_hand_left.urdf", True)])
def test_jacobian_finite_differences(urdf, free):
    r = OracleRobot(os.path.join(cases.URDF_DIR, urdf), add_dummy_free_joints=free)
    rng = np.random.default_rng(3)
    lim = r.joint_limits
    q = rng.uniform(lim[:, 0], lim[:, 1], (3, r.dof))
    links = [l for l in r.links if "tip" in l or l.endswith("_c") or l.endswith("_q")][:6]
    J = r.point_jacobians(q, links)
    eps = 1e-6
    for i in range(r.dof):
        dq = np.zeros(r.dof)
        dq[i] = eps
        Jn = (r.link_positions(q + dq, links) - r.link_positions(q - dq, links)) / (2 * eps)
        assert np.abs(J[..., i] - Jn).max() < 1e-8
    # LOCAL frame Jacobian consistency: R_link @ J_local[:3] == world point Jacobian (optimizer.py:279-284)
    R, _ = r.link_poses(q[:1], links[:1])
    Jl = r.frame_jacobian_local(q[0], links[0])
    assert np.allclose(R[0, 0] @ Jl[:3], J[0, 0], atol=1e-12)


@pytest.mark.parametrize("rel", ["teleop/allegro_hand_right.yml", "teleop/shadow_hand_right_dexpilot.yml",
                                 "offline/leap_hand_right.yml", "teleop/inspire_hand_right.yml"])
def test_lm_equals_scipy_tight_in_tracking_regime(rel):
    prob = cases.problem_from_config(rel)
    B = 6
    d = cases.reachable_set(prob, B, 0.05)
    kw = {}
    if prob.kind == "dexpilot":
        w, rv, _ = prob.dexpilot_preamble(d["ref"], np.zeros((B, prob.n_pair), bool))
        kw = dict(weights=w, dexpilot_ref=rv)
    x_lm = solvers.solve_lm_batched(prob, d["ref"], d["fixed"], d["last"], newton=True, **kw)
    x_t = solvers.solve_tight(prob, d["ref"], d["fixed"], d["last"], **kw)
    F_lm = prob.total(x_lm, d["ref"], d["fixed"], d["last"].astype(np.float64), **kw)
    F_t = prob.total(x_t, d["ref"], d["fixed"], d["last"].astype(np.float64), **kw)
    assert np.abs(x_lm - x_t).max() < 5e-6
    assert np.all(F_lm <= F_t + 1e-12)


MULTIMODAL = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "multimodal_frames.npz")


@pytest.mark.parametrize("key", ["teleop__shadow_hand_right_dexpilot", "offline__inspire_hand_right",
                                 "teleop__schunk_svh_hand_left", "teleop__allegro_hand_right_dexpilot"])
def test_lm_oracle_stays_in_the_basin_slsqp_converges_to(key):
    """Frames of the round-3 MI355X run whose answers were >= 1e-4 rad from the LM oracle of rounds 1-3 ("other minimum"
    rows; tests/golden/multimodal_frames.npz = gpurun_out/all_configs_far_frames.npz of that run, <= 12 frames per config:
    inputs, the library's answers, the old oracle's answers).  The arbiter is solve_tight: scipy's SLSQP -- the
    reference's own algorithm (optimizer.py:41,96-99) -- driven to convergence from the same start.  It converges to the
    library's recorded answer and never to the old oracle's: that oracle solved INDEFINITE damped models with a general
    linear solver (the stationary point of such a model is a saddle) and hopped basins when F happened to be lower there.
    Requiring a positive-definite model (solve_lm_batched(require_pd=True), the default since round 4) removes the hops.
    Over all 27 configs of the fixture (234 frames): SLSQP-to-convergence lands at the library's answer in 203, at the old
    oracle's in 19; the new oracle agrees with SLSQP-to-convergence in 208."""
    import warnings

    d = np.load(MULTIMODAL)
    rel = key.replace("__", "/") + ".yml"
    prob = cases.problem_from_config(rel)
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

    assert same(old, q_old).all()              # the fixture's "old oracle" column is what require_pd=False computes
    assert not same(q_old, q_lib).any()        # ... and every frame of the fixture was an "other minimum" frame
    assert same(tight, q_lib).all(), np.abs(tight - q_lib).max(1)   # SLSQP-to-convergence: the library's basin
    assert not same(tight, q_old).any()
    assert same(new, tight).all(), np.abs(new - tight).max(1)       # the positive-definite LM oracle agrees with it


ARBITER = json.load(open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "arbiter_counts.json")))


@pytest.mark.parametrize("key", sorted(ARBITER["per_config"]))
def test_arbiter_counts_of_every_fixture_config(key):
    """the independent arbiter over ALL 27 configs of the multimodal fixture, not four hand-picked ones.  Where
    SLSQP-to-convergence (solve_tight) lands -- the library's recorded answer / the rounds 1-3 oracle's -- and what the two LM
    oracles agree with is recomputed here and must equal the pinned counts (tests/golden/gen_arbiter_counts.py): a change
    of oracle/solvers.py that moves the checker shows up config by config."""
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden"))
    import gen_arbiter_counts

    k, got = gen_arbiter_counts.counts(key)
    assert got == ARBITER["per_config"][key], (key, got, ARBITER["per_config"][key])


def test_arbiter_aggregate_sides_with_the_library():
    """The aggregate the round-4 oracle repair rests on, with its loose ends stated: of the 234 recorded frames the arbiter
    converges to the library's answer in >= 200 and to the rounds 1-3 oracle's in <= 20; the positive-definite LM oracle
    agrees with the arbiter in >= 205; the rounds 1-3 rule (require_pd=False) still reproduces every recorded old answer
    (the comparison column `far_r3` of tests/test_gpu_all_configs.py is that unchanged oracle)."""
    t = ARBITER["total"]
    assert sum(v["frames"] for v in ARBITER["per_config"].values()) == t["frames"] == 234 and len(ARBITER["per_config"]) == 27
    assert t["tight_at_library"] >= 200 and t["tight_at_old_oracle"] <= 20
    assert t["new_oracle_at_tight"] >= 205 and t["new_oracle_at_library"] >= 205
    assert t["old_rule_reproduces_old_oracle"] == t["frames"]
    for f in t:
        assert t[f] == sum(v[f] for v in ARBITER["per_config"].values())


@pytest.mark.parametrize("rel,thr", [("teleop/allegro_hand_right.yml", 1e-2), ("offline/leap_hand_right.yml", 1e-2)])
def test_reference_round_trip_property_as_configured(rel, thr):
    """tests/test_optimizer.py:83-209 re-enacted on the oracle: normal_delta=0, scaling 1, 12 seeded solves."""
    over = dict(normal_delta=0)
    if "teleop" in rel:
        over.update(scaling_factor=1.0, low_pass_alpha=0)
    prob = cases.problem_from_config(rel, **over)
    d = cases.reachable_set(prob, 12, 0.5, seed=1, divide_scaling=False)
    x, evals = solvers.solve_ref_as_configured(prob, d["ref"].astype(np.float64), d["fixed"], d["last"])
    got = cases.fk_reference_values(prob, prob.full_qpos(x.astype(np.float64), d["fixed"]))
    err = np.linalg.norm(got - d["ref"], axis=-1).mean()
    assert err < thr
    assert evals.mean() > 3


def test_refsolve_golden_regression():
    """oracle.solvers.solve_ref_as_configured reproduces what the reference's own Optimizer.retarget returned
    (through the scipy stand-in for nlopt) on the first frames of the human sequence."""
    g = np.load(os.path.join(GOLD, "refsolve_golden.npz"))
    rel = "teleop/allegro_hand_right.yml"
    prob = cases.problem_from_config(rel)
    kp = np.load(cases.HUMAN_FIXTURE)[:4].astype(np.float64)
    refs = cases.ref_from_keypoints(prob, kp).astype(np.float32)  # seq_retarget.py:116
    last = prob.joint_limits.mean(1).astype(np.float32)
    for t in range(4):
        last = np.clip(last, prob.joint_limits[:, 0], prob.joint_limits[:, 1]).astype(np.float32)  # seq_retarget.py:118-120
        x, _ = solvers.solve_ref_as_configured(prob, refs[t:t + 1], None, last[None])
        assert np.abs(x[0] - g[_key(rel) + "__last_qpos"][t]).max() < 1e-5
        last = x[0]


def test_keypoint_preprocessing_oracle_matches_reference_golden():
    """oracle/preprocess.py == the reference's own SingleHandDetector.estimate_frame_from_hand_points and the lines
    around its call (single_hand_detector.py:102-104,129-158), as captured by tests/golden/gen_golden.py."""
    from oracle import preprocess

    g = np.load(os.path.join(GOLD, "mano_frame_golden.npz"))
    for hand, raw in (("right", g["raw"]), ("left", g["raw_left"])):
        jp, rot = preprocess.mano_joint_pos(raw, right=(hand == "right"))
        assert np.abs(jp - g[f"joint_pos_{hand}"]).max() < 1e-12
        assert np.abs(rot - g[f"wrist_rot_{hand}"]).max() < 1e-12
        # a rotation (det +1), and the wrist ends up at the origin
        assert np.allclose(np.einsum("bij,bkj->bik", rot, rot), np.eye(3), atol=1e-12)
        assert np.allclose(np.linalg.det(rot), 1.0, atol=1e-12)
        assert np.abs(jp[:, 0]).max() == 0.0


def test_keypoint_preprocessing_undoes_a_rigid_motion():
    """Property: the MANO-frame joint positions do not depend on where the detector's camera frame was."""
    from oracle import preprocess

    kp = cases.human_keypoints(40, seed=3).astype(np.float64)
    rng = np.random.default_rng(5)
    a, _ = preprocess.mano_joint_pos(kp)
    q, _ = np.linalg.qr(rng.standard_normal((3, 3)))
    q *= np.sign(np.linalg.det(q))
    b, _ = preprocess.mano_joint_pos(kp @ q.T + rng.uniform(-1, 1, 3))
    assert np.abs(a - b).max() < 1e-9
