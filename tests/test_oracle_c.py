"""CPU tests that pin the oracle's plain-C restatement (oracle/csrc/dexr_oracle.c through oracle/cport.py) before
anything is timed or compared with it:

* one closure evaluation vs the golden vectors produced by the REFERENCE'S OWN closures
  (tests/golden/objective_golden.npz, optimizer.py:146-198, 249-304, 510-575);
* its forward kinematics vs the link positions of the reference's own URDF reader / FK (tests/golden/fk_golden.npz);
* the as-configured SLSQP solve driven by the C closure == the numpy oracle's, bit for bit.
"""
import os

import numpy as np
import pytest

from oracle import cases, cport, solvers
from oracle.kin import OracleRobot
from oracle.objectives import OracleProblem

GOLD = os.path.join(os.path.dirname(__file__), "golden")
OBJ_CONFIGS = [
    "teleop/allegro_hand_right.yml", "teleop/shadow_hand_right_dexpilot.yml", "offline/leap_hand_right.yml",
    "teleop/ability_hand_right.yml", "teleop/inspire_hand_right_dexpilot.yml", "offline/schunk_svh_hand_right.yml",
    "teleop/panda_gripper.yml", "teleop/shadow_hand_left.yml", "teleop/allegro_hand_left_dexpilot.yml",
]


def _key(rel):
    return rel.replace("/", "__").replace(".yml", "")


def test_c_restatement_builds_with_gcc():
    assert os.path.exists(cport.build())
    cport.load()


@pytest.mark.parametrize("rel", OBJ_CONFIGS)
def test_c_closure_matches_reference_closures(rel):
    g = np.load(os.path.join(GOLD, "objective_golden.npz"))
    k = _key(rel)
    prob = cases.problem_from_config(rel)
    cp = cport.CProblem(prob)
    ref, fixed, last, x = g[k + "__ref"], g[k + "__fixed"], g[k + "__last"], g[k + "__x"]
    w = rv = None
    if prob.kind == "dexpilot":  # the per-frame pre-amble stays in numpy (it is not part of an evaluation)
        w, rv, _ = prob.dexpilot_preamble(ref, g[k + "__state_in"])
    for b in range(x.shape[0]):
        tgt = cp.target(ref[b], None if rv is None else rv[b])
        f, grad = cp.evaluate(x[b], tgt, fixed[b] if fixed.size else None, last[b].astype(np.float64),
                              None if w is None else w[b])
        assert np.isclose(f, g[k + "__f"][b], rtol=1e-12, atol=1e-14)
        assert np.allclose(grad, g[k + "__grad"][b], rtol=1e-10, atol=1e-13)
        f2, none = cp.evaluate(x[b], tgt, fixed[b] if fixed.size else None, None, None if w is None else w[b], need_grad=False)
        assert f2 == f and none is None  # the value never carries the regulariser


FK = np.load(os.path.join(GOLD, "fk_golden.npz"))


@pytest.mark.parametrize("key", ["shadow_hand__shadow_hand_right__free", "schunk_hand__schunk_svh_hand_left",
                                 "testurdf__messy_arm_hand", "panda_gripper__panda_gripper_glb__free"])
def test_c_forward_kinematics_matches_reference_fk(key):
    free = key.endswith("__free")
    base = key[: -len("__free")] if free else key
    if base.startswith("testurdf__"):
        path = os.path.join(os.path.dirname(__file__), "urdf", base[len("testurdf__"):] + ".urdf")
    else:
        path = os.path.join(cases.URDF_DIR, base.replace("__", "/") + ".urdf")
    r = OracleRobot(path, free)
    links = FK[key + "__links"].tolist()
    prob = OracleProblem(r, "position", target_link_names=links)
    cp = cport.CProblem(prob)
    for c in range(FK[key + "__cfg"].shape[0]):
        val = dict(zip(FK[key + "__joints"].tolist(), FK[key + "__cfg"][c].tolist()))
        q = np.array([val.get(n, 0.0) for n in r.dof_joint_names])
        q = r.mimic_forward(q)
        got = cp.link_positions(q)
        assert np.abs(got - FK[key + "__T"][c][:, :3, 3]).max() < 1e-12, key


@pytest.mark.parametrize("rel", ["teleop/allegro_hand_right.yml", "teleop/inspire_hand_right_dexpilot.yml",
                                 "offline/leap_hand_right.yml"])
def test_c_driven_slsqp_equals_numpy_driven_slsqp(rel):
    prob = cases.problem_from_config(rel)
    cp = cport.CProblem(prob)
    d = cases.human_set(prob, 6, seed=5, sigma=0.1)
    kw = {}
    if prob.kind == "dexpilot":
        w, rv, _ = prob.dexpilot_preamble(d["ref"], np.zeros((6, prob.n_pair), bool))
        kw = dict(weights=w, dexpilot_ref=rv)
    q1, e1 = solvers.solve_ref_as_configured(prob, d["ref"], d["fixed"], d["last"], **kw)
    q2, e2 = cport.solve_ref_as_configured_c(cp, d["ref"], d["fixed"], d["last"], **kw)
    assert np.abs(q1.astype(np.float64) - q2).max() < 1e-6  # same iterates up to the rounding of the two closures
    assert np.abs(e1 - e2).max() <= 2
