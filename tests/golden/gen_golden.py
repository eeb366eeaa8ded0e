#!/usr/bin/env python3
"""Generate the committed golden fixtures under tests/golden/ (run in the BUILD container only).

* human_joint_right_f32.npy -- float32 copy of the reference's only data fixture,
  /root/reference/example/profiling/human_joint_right.pkl (621 x (21,3) MANO-frame keypoints).
* objective_golden.npz -- (x, last, ref, fixed[, state]) -> (value, grad[, state']) produced by the REFERENCE'S
  OWN objective closures (/root/reference/src/dex_retargeting/optimizer.py:138-200, 241-306, 456-577 and
  kinematics_adaptor.py), imported from /root/reference through oracle/ref_harness.py (pinocchio replaced by
  oracle.kin, nlopt by a scipy stand-in).  These pin oracle/objectives.py and the GPU `dexr_eval` kernel.
* refsolve_golden.npz -- qpos returned by the reference's own ``Optimizer.retarget`` / ``SeqRetargeting.retarget``
  (optimizer.py:77-102, seq_retarget.py:112-134) with the scipy-SLSQP stand-in for nlopt: "reference as
  configured" answers for a short human-keypoint sequence (information + regression of oracle.solvers).

* mano_frame_golden.npz -- raw detector-style keypoints (the human fixture under random rigid motions + noise)
  -> ``mediapipe_wrist_rot`` / ``joint_pos`` computed by the REFERENCE'S OWN
  ``SingleHandDetector.estimate_frame_from_hand_points`` and the three lines around its call
  (example/vector_retargeting/single_hand_detector.py:102-104,129-158), right and left hand.  Pins
  oracle/preprocess.py and the GPU ``dexr_mano_keypoints`` kernel.

Usage: python tests/golden/gen_golden.py [--only-mano]
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, REPO)

from oracle import cases, ref_harness  # noqa: E402

CONFIGS = [
    "teleop/allegro_hand_right.yml", "teleop/shadow_hand_right_dexpilot.yml", "offline/leap_hand_right.yml",
    "teleop/ability_hand_right.yml", "teleop/inspire_hand_right_dexpilot.yml", "offline/schunk_svh_hand_right.yml",
    "teleop/panda_gripper.yml", "teleop/shadow_hand_left.yml", "teleop/allegro_hand_left_dexpilot.yml",
]
N_SAMPLES = 6


def build_reference_optimizer(rel):
    """RetargetingConfig.build() of the reference (retargeting_config.py:167-257) re-enacted with the stand-in robot."""
    opt_mod, ka_mod, sr_mod, ou_mod = ref_harness.import_reference()
    cfg = cases.load_cfg(rel)
    kind = cfg["type"].lower()
    free = bool(cfg.get("add_dummy_free_joint", False))
    robot = ref_harness.FakeRobotWrapper(os.path.join(cases.URDF_DIR, cfg["urdf_path"]), free)
    tj = cfg.get("target_joint_names")
    if free and tj is not None:
        tj = [f"dummy_{n}_translation_joint" for n in "xyz"] + [f"dummy_{n}_rotation_joint" for n in "xyz"] + tj
    names = tj if tj is not None else robot.dof_joint_names
    if kind == "position":
        o = opt_mod.PositionOptimizer(robot, names, target_link_names=cfg["target_link_names"],
                                      target_link_human_indices=np.array(cfg["target_link_human_indices"]),
                                      norm_delta=cfg.get("normal_delta", 4e-3), huber_delta=cfg.get("huber_delta", 0.02))
    elif kind == "vector":
        o = opt_mod.VectorOptimizer(robot, names, target_origin_link_names=cfg["target_origin_link_names"],
                                    target_task_link_names=cfg["target_task_link_names"],
                                    target_link_human_indices=np.array(cfg["target_link_human_indices"]),
                                    scaling=cfg.get("scaling_factor", 1.0), norm_delta=cfg.get("normal_delta", 4e-3),
                                    huber_delta=cfg.get("huber_delta", 0.02))
    else:
        o = opt_mod.DexPilotOptimizer(robot, names, finger_tip_link_names=cfg["finger_tip_link_names"],
                                      wrist_link_name=cfg["wrist_link_name"], scaling=cfg.get("scaling_factor", 1.0))
    if robot.kin.mimic and not cfg.get("ignore_mimic_joint", False):
        mim = robot.kin.mimic
        ad = ka_mod.MimicJointKinematicAdaptor(robot, target_joint_names=names,
                                               source_joint_names=[m[1] for m in mim],
                                               mimic_joint_names=[m[0] for m in mim],
                                               multipliers=[m[2] for m in mim], offsets=[m[3] for m in mim])
        o.set_kinematic_adaptor(ad)
    lp = ou_mod.LPFilter(cfg.get("low_pass_alpha", 0.1)) if 0 <= cfg.get("low_pass_alpha", 0.1) <= 1 else None
    seq = sr_mod.SeqRetargeting(o, has_joint_limits=True, lp_filter=lp)
    return o, seq


def random_rotations(n, rng):
    q = rng.standard_normal((n, 4))
    q /= np.linalg.norm(q, axis=1, keepdims=True)
    w, x, y, z = q.T
    return np.stack([np.stack([1 - 2 * (y * y + z * z), 2 * (x * y - z * w), 2 * (x * z + y * w)], -1),
                     np.stack([2 * (x * y + z * w), 1 - 2 * (x * x + z * z), 2 * (y * z - x * w)], -1),
                     np.stack([2 * (x * z - y * w), 2 * (y * z + x * w), 1 - 2 * (x * x + y * y)], -1)], 1)


def gen_mano_frame():
    det = ref_harness.import_reference_detector()
    rng = np.random.default_rng(31)
    fixture = np.load(os.path.join(HERE, "human_joint_right_f32.npy")).astype(np.float64)
    n = 96
    idx = rng.integers(0, fixture.shape[0], n)
    R = random_rotations(n, rng)
    t = rng.uniform(-0.5, 0.5, (n, 1, 3))
    raw = (np.einsum("bij,bkj->bki", R, fixture[idx]) + t + 1e-3 * rng.standard_normal((n, 21, 3))).astype(np.float32)
    out = {"raw": raw}
    for hand, op in (("right", det.OPERATOR2MANO_RIGHT), ("left", det.OPERATOR2MANO_LEFT)):
        kp_in = raw.astype(np.float64)
        if hand == "left":
            kp_in = kp_in * np.array([1.0, -1.0, 1.0])  # mirrored hand
            out["raw_left"] = kp_in.astype(np.float32)
            kp_in = out["raw_left"].astype(np.float64)
        jp, rot = [], []
        for b in range(n):
            c = kp_in[b] - kp_in[b][0:1, :]
            r = det.SingleHandDetector.estimate_frame_from_hand_points(c)
            jp.append(c @ r @ op)
            rot.append(r)
        out[f"joint_pos_{hand}"], out[f"wrist_rot_{hand}"] = np.array(jp), np.array(rot)
    np.savez_compressed(os.path.join(HERE, "mano_frame_golden.npz"), **out)
    print("mano_frame_golden.npz:", {k: v.shape for k, v in out.items()})


def main():
    if "--only-mano" in sys.argv:
        gen_mano_frame()
        return
    kp = np.load("/root/reference/example/profiling/human_joint_right.pkl", allow_pickle=True)
    np.save(os.path.join(HERE, "human_joint_right_f32.npy"), np.stack(kp).astype(np.float32))

    out = {}
    for rel in CONFIGS:
        key = rel.replace("/", "__").replace(".yml", "")
        o, _ = build_reference_optimizer(rel)
        prob = cases.problem_from_config(rel)
        assert list(o.idx_pin2target) == list(prob.idx_pin2target), rel
        assert list(o.idx_pin2fixed) == list(prob.idx_pin2fixed), rel
        half = N_SAMPLES // 2
        d1 = cases.reachable_set(prob, half, 0.3, seed=7)
        d2 = cases.human_set(prob, N_SAMPLES - half, seed=7, sigma=0.2)
        ref = np.concatenate([d1["ref"], d2["ref"]]).astype(np.float32)
        fixed = np.concatenate([d1["fixed"], d2["fixed"]]).astype(np.float32)
        last = np.concatenate([d1["last"], d2["last"]]).astype(np.float32)
        rng = np.random.default_rng(11)
        lim = prob.joint_limits
        x = np.clip(last.astype(np.float64) + 0.1 * rng.standard_normal(last.shape), lim[:, 0], lim[:, 1])
        if prob.kind == "dexpilot":  # shrink some pair vectors so the projection logic (optimizer.py:466-476) fires
            ref[1, 0] *= 0.1
            ref[2, :2] *= 0.1
            ref[4, : prob.n_pair] *= 0.15
            ref[5, 0] *= 0.3  # between project_dist and escape_dist for most frames: keeps the previous state
        fs, gs, st_in, st_out = [], [], [], []
        for b in range(N_SAMPLES):
            if prob.kind == "dexpilot":  # alternate the incoming projection state
                o.projected[:] = (b % 2 == 1)
                st_in.append(o.projected.copy())
            fn = o.get_objective_function(ref[b], fixed[b], last[b])  # float32 inputs, as SeqRetargeting passes them
            g = np.zeros(prob.n_opt)
            fs.append(fn(x[b].copy(), g))
            gs.append(g)
            if prob.kind == "dexpilot":
                st_out.append(o.projected.copy())
        out[key + "__ref"], out[key + "__fixed"], out[key + "__last"], out[key + "__x"] = ref, fixed, last, x
        out[key + "__f"], out[key + "__grad"] = np.array(fs), np.array(gs)
        if prob.kind == "dexpilot":
            out[key + "__state_in"], out[key + "__state_out"] = np.array(st_in), np.array(st_out)
        print(f"{rel:45s} f={np.array(fs)}")
    np.savez_compressed(os.path.join(HERE, "objective_golden.npz"), **out)

    # reference-as-configured answers on a short real sequence (stand-in SLSQP)
    sol = {}
    for rel in ["teleop/allegro_hand_right.yml", "teleop/shadow_hand_right_dexpilot.yml", "offline/leap_hand_right.yml"]:
        key = rel.replace("/", "__").replace(".yml", "")
        o, seq = build_reference_optimizer(rel)
        prob = cases.problem_from_config(rel)
        kpf = np.load(os.path.join(HERE, "human_joint_right_f32.npy"))[:24].astype(np.float64)
        refs = cases.ref_from_keypoints(prob, kpf)
        outs, raw = [], []
        for t in range(refs.shape[0]):
            outs.append(seq.retarget(refs[t]))
            raw.append(np.array(seq.last_qpos, dtype=np.float32))
        sol[key + "__robot_qpos"] = np.array(outs)
        sol[key + "__last_qpos"] = np.array(raw)
        print(f"{rel:45s} seq done, evals={o.opt.n_evals}")
    np.savez_compressed(os.path.join(HERE, "refsolve_golden.npz"), **sol)
    gen_mano_frame()


if __name__ == "__main__":
    main()
