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

* fk_golden.npz -- link global transforms of every fixture URDF (with and without dummy free joints) at seeded
  configurations, computed by the REFERENCE'S OWN URDF reader and forward kinematics
  (/root/reference/src/dex_retargeting/yourdfpy.py: URDF.load :896-959, _forward_kinematics_joint :1013-1050,
  build_tree / update_kinematics / get_link_global_transform :1862-1939, _add_dummy_joints :1942-1984) through
  oracle/ref_urdf.py, plus the URDF text the reference's ``write_xml_file`` (:1098-1105) emits -- the file pinocchio
  actually parses in RetargetingConfig.build (retargeting_config.py:176-186).  Pins oracle/kin.py,
  dex_retargeting_amd/urdf.py, the table compiler and the GPU ``dexr_fk`` kernel (keyed by joint / link NAME).
* warm_start_golden.npz -- ``last_qpos`` after the REFERENCE'S OWN ``SeqRetargeting.warm_start``
  (seq_retarget.py:45-110) for seeded wrist poses, both hand types and both conventions, on every offline config.
* seq_wrapper_golden.npz -- the REFERENCE'S OWN ``SeqRetargeting.retarget`` (seq_retarget.py:112-134) + ``LPFilter``
  (optimizer_utils.py:7-13) + mimic adaptor wrapped around a deterministic stub optimizer that replays recorded
  solver outputs: pins the per-frame bookkeeping (clip of the carried qpos, unfiltered carry, robot-qpos composition,
  mimic fill, low-pass filter) independently of any solver.

Usage: python tests/golden/gen_golden.py [--only-mano | --only-fk | --only-seq]
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


def gen_fk_golden():
    import glob

    from oracle import ref_urdf

    out = {}
    n_cfg = 4
    # the shipped robot fixtures + the "messy" test URDFs (tests/urdf/: constructs exported real-world URDFs carry that the
    # authored robot fixtures do not use; keys prefixed "testurdf__")
    paths = [(p, os.path.relpath(p, cases.URDF_DIR).replace("/", "__").replace(".urdf", ""))
             for p in sorted(glob.glob(os.path.join(cases.URDF_DIR, "*", "*.urdf")))]
    paths += [(os.path.join(REPO, "tests", "urdf", n + ".urdf"), "testurdf__" + n) for n in ("messy_arm_hand",)]
    for path, base_key in paths:
        for dummy in (False, True):
            key = base_key + ("__free" if dummy else "")
            u = ref_urdf.load_reference_urdf(path, dummy)
            # the axes exactly as the reference's reader parsed them (yourdfpy.py:1631-1643) ...
            moving = [j for j in u.robot.joints if j.type != "fixed"]
            out[key + "__axis_names"] = np.array([j.name for j in moving])
            out[key + "__axis_raw"] = np.array([np.asarray(j.axis, dtype=np.float64) for j in moving])
            # ... and, for the FK below, normalised: yourdfpy feeds the axis to Rodrigues' formula as written (a non-unit
            # axis gives a non-orthonormal "rotation"), whereas the model the reference actually solves with comes from
            # pinocchio's URDF parser, which normalises it [not-in-ref: third-party behaviour; urdfdom / pinocchio
            # JointModel{Revolute,Prismatic}Unaligned take axis.normalized()]
            for j in moving:
                j.axis = np.asarray(j.axis, dtype=np.float64) / np.linalg.norm(j.axis)
            lo = np.array([j.limit.lower for j in u.actuated_joints], dtype=np.float64)
            hi = np.array([j.limit.upper for j in u.actuated_joints], dtype=np.float64)
            rng = np.random.default_rng(len(key) * 1000 + len(u.robot.joints) + (7 if dummy else 0))
            cfg = rng.uniform(lo, hi, size=(n_cfg, len(lo)))
            cfg[0] = 0.0  # the zero configuration (may lie outside the limits: FK does not care)
            cfg[1] = lo
            out[key + "__links"] = np.array([l.name for l in u.robot.links])
            out[key + "__joints"] = np.array(u.actuated_joint_names)
            out[key + "__lower"], out[key + "__upper"] = lo, hi
            mim = [j for j in u.robot.joints if j.mimic is not None]
            out[key + "__mimic"] = np.array([j.name for j in mim] or [""])
            out[key + "__mimic_src"] = np.array([j.mimic.joint for j in mim] or [""])
            out[key + "__mimic_mult"] = np.array([j.mimic.multiplier for j in mim], dtype=np.float64)
            out[key + "__mimic_off"] = np.array([j.mimic.offset for j in mim], dtype=np.float64)
            out[key + "__cfg"] = cfg
            out[key + "__T"] = np.stack([ref_urdf.reference_link_transforms(u, c) for c in cfg])
            # what pinocchio is given: the re-written URDF (mimic tags are not written, yourdfpy.py:1787-1802)
            tmp = os.path.join("/tmp", f"dexr_rewrite_{os.getpid()}.urdf")
            u.write_xml_file(tmp)
            out[key + "__rewritten"] = np.array(open(tmp).read())
            os.remove(tmp)
            print(f"{key:40s} links={len(u.robot.links)} actuated={len(lo)} mimic={len(mim)}")
    np.savez_compressed(os.path.join(HERE, "fk_golden.npz"), **out)


class _ReplayOptimizer:
    """Deterministic stub with the attribute surface SeqRetargeting touches (seq_retarget.py:12-134): ``retarget``
    records its arguments and replays a prepared answer."""

    def __init__(self, base, answers):
        self.__dict__["_base"] = base
        self.__dict__["answers"] = answers
        self.__dict__["calls"] = []

    def __getattr__(self, name):
        return getattr(self._base, name)

    def retarget(self, ref_value, fixed_qpos, last_qpos):
        self.calls.append((np.array(ref_value), np.array(fixed_qpos), np.array(last_qpos)))
        return self.answers[len(self.calls) - 1].astype(np.float32)


def gen_seq_and_warm_start():
    opt_mod, ka_mod, sr_mod, ou_mod = ref_harness.import_reference()
    import dex_retargeting.constants as rc  # noqa: E402  (pure-python constants of the reference)

    ws, sq = {}, {}
    offline = sorted(f for f in os.listdir(os.path.join(cases.CONFIG_DIR, "offline")) if f.endswith(".yml"))
    for f in offline:
        rel = "offline/" + f
        key = rel.replace("/", "__").replace(".yml", "")
        o, seq = build_reference_optimizer(rel)
        rng = np.random.default_rng(len(key))
        n = 6
        pos = rng.uniform(-0.4, 0.4, (n, 3))
        quat = rng.standard_normal((n, 4))
        hand = [rc.HandType.right if i % 2 == 0 else rc.HandType.left for i in range(n)]
        mano = [i % 3 == 0 for i in range(n)]
        outs = []
        for i in range(n):
            seq.reset()
            seq.warm_start(pos[i], quat[i], hand[i], mano[i])
            outs.append(np.array(seq.last_qpos, dtype=np.float32))
        ws[key + "__pos"], ws[key + "__quat"] = pos, quat
        ws[key + "__hand_is_right"] = np.array([h == rc.HandType.right for h in hand])
        ws[key + "__mano"] = np.array(mano)
        ws[key + "__last_qpos"] = np.array(outs)
        ws[key + "__target_joint_names"] = np.array(o.target_joint_names)
        print(f"warm_start {rel}")
    np.savez_compressed(os.path.join(HERE, "warm_start_golden.npz"), **ws)

    # SeqRetargeting bookkeeping around a replaying stub: one config per structural case
    for rel in ["teleop/allegro_hand_right.yml",          # no mimic, alpha 0.2
                "teleop/ability_hand_right.yml",          # mimic joints filled after the solve
                "offline/inspire_hand_right.yml",         # free joints + mimic, alpha 1 (filter is the identity)
                "teleop/panda_gripper.yml"]:              # 1 target joint + mimic finger
        key = rel.replace("/", "__").replace(".yml", "")
        o, _ = build_reference_optimizer(rel)
        cfg = cases.load_cfg(rel)
        T = 12
        rng = np.random.default_rng(5 + len(key))
        lim = o.robot.joint_limits[o.idx_pin2target]
        # recorded "solver outputs": inside the optimiser's widened box, some beyond the joint limits by < 1e-3 so that
        # the clip of the carried value (seq_retarget.py:118-120) matters
        ans = rng.uniform(lim[:, 0], lim[:, 1], (T, len(lim)))
        ans[3] = lim[:, 0] - 9e-4
        ans[7] = lim[:, 1] + 9e-4
        stub = _ReplayOptimizer(o, ans)
        alpha = cfg.get("low_pass_alpha", 0.1)
        seq = sr_mod.SeqRetargeting(stub, has_joint_limits=True, lp_filter=ou_mod.LPFilter(alpha) if 0 <= alpha <= 1 else None)
        n_ref = len(o.target_link_human_indices[0]) if o.retargeting_type != "POSITION" else len(o.target_link_human_indices)
        outs = []
        for t in range(T):
            outs.append(seq.retarget(rng.standard_normal((n_ref, 3)), fixed_qpos=np.zeros(len(o.idx_pin2fixed))))
        sq[key + "__answers"] = ans.astype(np.float32)
        sq[key + "__alpha"] = np.array(alpha)
        sq[key + "__robot_qpos"] = np.array(outs)
        sq[key + "__last_given"] = np.array([c[2] for c in stub.calls])
        sq[key + "__joint_names"] = np.array(seq.joint_names)
        sq[key + "__target_joint_names"] = np.array(o.target_joint_names)
        print(f"seq wrapper {rel}: alpha={alpha}")
    np.savez_compressed(os.path.join(HERE, "seq_wrapper_golden.npz"), **sq)


def main():
    if "--only-mano" in sys.argv:
        gen_mano_frame()
        return
    if "--only-fk" in sys.argv:
        gen_fk_golden()
        return
    if "--only-seq" in sys.argv:
        gen_seq_and_warm_start()
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
    gen_fk_golden()
    gen_seq_and_warm_start()


if __name__ == "__main__":
    main()
