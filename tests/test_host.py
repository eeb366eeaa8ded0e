"""CPU tests of the host side: URDF reader, table compiler, config/API surface, C-ABI exports (no GPU compute).

The config tests mirror the reference's /root/reference/tests/test_retargeting_config.py:36-125.
"""
import ctypes
import glob
import os
import re

import numpy as np
import pytest

import table_interp as ti
from dex_retargeting_amd import _lib, model_compiler as mc
from dex_retargeting_amd.constants import (DEFAULT_URDF_DIR, ROBOT_NAMES, HandType, RetargetingType, RobotName,
                                           get_default_config_path)
from dex_retargeting_amd.optimizer import DexPilotOptimizer
from dex_retargeting_amd.optimizer_utils import LPFilter
from dex_retargeting_amd.retargeting_config import RetargetingConfig
from dex_retargeting_amd.urdf import DUMMY_JOINT_NAMES, KinematicModel, parse_urdf
from oracle import cases
from oracle.kin import OracleRobot

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
RetargetingConfig.set_default_urdf_dir(str(DEFAULT_URDF_DIR))
URDFS = sorted(glob.glob(os.path.join(str(DEFAULT_URDF_DIR), "*", "*.urdf")))


def _has_gpu():
    return _lib.load().dexr_device_count() > 0


# ---- C ABI ----------------------------------------------------------------------------------------------
def test_library_loads_and_exports_every_declared_symbol():
    lib = _lib.load()
    header = open(os.path.join(REPO, "include", "dexr.h")).read()
    declared = set(re.findall(r"\b(dexr_[a-z0-9_]+)\s*\(", header))
    assert declared >= set(_lib.EXPORTS)
    for name in declared:
        assert hasattr(lib, name), f"libdexr.so does not export {name}"
    assert b"gfx950" in lib.dexr_version()


def test_solve_options_struct_matches_header():
    # the ctypes mirror of dexr_solve_options must list the header's fields in the header's order (all 4-byte)
    h = open(os.path.join(REPO, "include", "dexr.h")).read()
    body = re.search(r"typedef struct dexr_solve_options \{(.*?)\} dexr_solve_options;", h, re.S).group(1)
    body = re.sub(r"/\*.*?\*/", "", body, flags=re.S)
    fields = re.findall(r"\b(?:int32_t|float)\s+(\w+)\s*;", body)
    assert fields == [f[0] for f in _lib.SolveOptions._fields_]
    import ctypes

    assert ctypes.sizeof(_lib.SolveOptions) == 4 * len(fields)
    o = _lib.default_options()
    assert (o.max_iter, o.newton, o.precision, o.polish, o.strict) == (64, 1, 0, -1, 0) and abs(o.tol - 2e-6) < 1e-12


def test_tuning_struct_matches_header_and_library_reads_no_environment():
    h = open(os.path.join(REPO, "include", "dexr.h")).read()
    body = re.search(r"typedef struct dexr_tuning \{(.*?)\} dexr_tuning;", h, re.S).group(1)
    body = re.sub(r"/\*.*?\*/", "", body, flags=re.S)
    fields = []
    for decl in re.findall(r"\b(?:uint32_t|int32_t|float)\s+([\w\s,]+);", body):
        fields += [f.strip() for f in decl.split(",")]
    assert fields == [f[0] for f in _lib.Tuning._fields_]
    import ctypes

    assert ctypes.sizeof(_lib.Tuning) == 4 * len(fields)
    # developer knobs are fields of dexr_tuning / dexr_solve_options, not environment variables
    for src in ("dexr_api.hip", "dexr_kernel.hpp", "dexr_quad.hpp", "dexr_big.hpp", "dexr_prep.hip"):
        path = os.path.join(REPO, "dex_retargeting_amd", "csrc", src)
        if os.path.exists(path):
            assert "getenv" not in open(path).read(), src


def test_table_struct_sizes_match_header():
    # numpy dtypes in model_compiler.py must be byte-identical to the C structs in include/dexr_tables.h
    h = open(os.path.join(REPO, "include", "dexr_tables.h")).read()
    assert int(re.search(r"#define DEXR_MAXJ (\d+)", h).group(1)) == mc.MAXJ
    assert int(re.search(r"#define DEXR_MAXF (\d+)", h).group(1)) == mc.MAXF
    assert int(re.search(r"#define DEXR_MAXT (\d+)", h).group(1)) == mc.MAXT
    assert int(re.search(r"#define DEXR_TABLE_VERSION (\d+)u", h).group(1)) == mc.VERSION
    words = 4 + mc.MAXJ * 12 + 8 * mc.MAXJ + 4 * mc.MAXJ + mc.MAXF * 5 + 3 * mc.MAXT + 1 + 3 * mc.MAXJ
    assert mc.COMP_DTYPE.itemsize == 4 * words
    assert mc.HEADER_DTYPE.itemsize == 4 * (18 + 1 + 2 * mc.MAXT)


def test_model_create_rejects_malformed_blobs():
    km = KinematicModel(parse_urdf(URDFS[0]))
    blob = bytearray(mc.compile_fk(km, [km.frames[-1].name]).to_blob())
    with pytest.raises(_lib.DexrError, match="blob"):
        _lib.Model(bytes(blob[:10]))
    bad = bytearray(blob)
    bad[0] ^= 0xFF
    with pytest.raises(_lib.DexrError, match="magic"):
        _lib.Model(bytes(bad))
    bad = bytearray(blob)
    bad[4] = 99
    with pytest.raises(_lib.DexrError, match="version"):
        _lib.Model(bytes(bad))
    with pytest.raises(_lib.DexrError, match="size"):
        _lib.Model(bytes(blob) + b"\0\0\0\0")


@pytest.mark.skipif(_has_gpu(), reason="only meaningful on a box without a GPU")
def test_no_silent_cpu_fallback():
    km = KinematicModel(parse_urdf(URDFS[0]))
    with pytest.raises(_lib.DexrError):  # a well-formed model cannot be created without a device: fail loudly
        _lib.Model(mc.compile_fk(km, [km.frames[-1].name]).to_blob())


# ---- URDF reader + table compiler ---------------------------------------------------------------------------
@pytest.mark.parametrize("path", URDFS, ids=[os.path.basename(p) for p in URDFS])
@pytest.mark.parametrize("free", [False, True])
def test_compiled_tables_reproduce_oracle_kinematics(path, free):
    km = KinematicModel(parse_urdf(path, free))
    orc = OracleRobot(path, free)
    assert km.dof_joint_names == orc.dof_joint_names
    assert np.allclose(km.joint_limits, orc.joint_limits)
    links = [f.name for f in km.frames]
    cm = mc.compile_fk(km, links)
    rng = np.random.default_rng(5)
    lim = km.joint_limits
    q = rng.uniform(lim[:, 0], lim[:, 1], (4, km.dof))
    want = orc.link_positions(q, links)
    wantJ = orc.point_jacobians(q, links)
    for c in cm.comps:
        qq = ti.joint_values(c, q_full=q)
        P, axes, orgs = ti.frame_positions(c, qq)
        for t in range(int(c["n_term"])):
            f, row = int(c["term_task"][t]), int(c["term_ref"][t])
            assert np.abs(P[:, f] - want[:, row]).max() < 2e-7
            # ancestor masks + world axes/origins give the point Jacobian the kernels use
            for k in range(int(c["n_joint"])):
                pin = int(c["src_idx"][k])
                if (int(c["frame_anc"][f]) >> k) & 1:
                    col = np.cross(axes[k], P[:, f] - orgs[k]) if int(c["jtype"][k]) == 0 else axes[k]
                else:
                    col = np.zeros((4, 3))
                assert np.abs(col - wantJ[:, row, :, pin]).max() < 2e-6


def test_deep_fork_tree_uses_slots():
    import tempfile
    xml = ['<robot name="tree"><link name="l0"/>']
    # chain a0-a1 with a fork after a0 (two branches) and a nested fork after b0
    joints = [("a0", "l0", "l1"), ("b0", "l1", "l2"), ("c0", "l2", "l3"), ("c1", "l2", "l4"), ("b1", "l1", "l5"),
              ("d0", "l5", "l6")]
    for i in range(1, 7):
        xml.append(f'<link name="l{i}"/>')
    for n, (name, p, c) in enumerate(joints):
        xml.append(f'<joint name="{name}" type="revolute"><parent link="{p}"/><child link="{c}"/>'
                   f'<origin xyz="0.0{n+1} 0.02 0.1" rpy="0.{n} 0.2 -0.{n}"/><axis xyz="{(n%3==0)*1} {(n%3==1)*1} {(n%3==2)*1}"/>'
                   f'<limit lower="-1" upper="1"/></joint>')
    xml.append("</robot>")
    with tempfile.NamedTemporaryFile("w", suffix=".urdf", delete=False) as f:
        f.write("".join(xml))
    km = KinematicModel(parse_urdf(f.name))
    orc = OracleRobot(f.name)
    links = [f"l{i}" for i in range(7)]
    cm = mc.compile_fk(km, links)
    assert int(cm.comps[0]["save"].max()) >= 1  # nested fork -> two live slots
    q = np.random.default_rng(0).uniform(-1, 1, (3, km.dof))
    P, _, _ = ti.frame_positions(cm.comps[0], ti.joint_values(cm.comps[0], q_full=q))
    want = orc.link_positions(q, links)
    for t in range(7):
        assert np.abs(P[:, int(cm.comps[0]["term_task"][t])] - want[:, t]).max() < 2e-7
    os.unlink(f.name)


def test_components_partition_the_variables():
    cfg = RetargetingConfig.load_from_file(get_default_config_path(RobotName.allegro, RetargetingType.vector, HandType.right))
    opt = cfg._build_optimizer()
    opt.set_joint_limit(opt.robot.joint_limits[opt.idx_pin2target])
    cm = opt.compiled_model()
    assert cm.n_comp == 4 and sorted(v for vs in cm.comp_vars for v in vs) == list(range(16))
    assert [int(c["n_joint"]) for c in cm.comps] == [4, 4, 4, 4]
    # DexPilot couples every finger: one component
    cfg = RetargetingConfig.load_from_file(get_default_config_path(RobotName.allegro, RetargetingType.dexpilot, HandType.right))
    opt = cfg._build_optimizer()
    assert opt.compiled_model().n_comp == 1
    # mimic hand: mimic joints ride in the component of their source
    cfg = RetargetingConfig.load_from_file(get_default_config_path(RobotName.ability, RetargetingType.vector, HandType.right))
    opt = cfg._build_optimizer()
    cm = opt.compiled_model()
    assert cm.n_comp == 5 and cm.n_opt == 6
    assert sum(int((c["src_kind"][: int(c["n_joint"])] == mc.SRC_MIMIC).sum()) for c in cm.comps) == 4


# ---- config / API surface (mirrors reference tests/test_retargeting_config.py) ---------------------------------
ALL_CONFIGS = sorted(glob.glob(os.path.join(cases.CONFIG_DIR, "*", "*.yml")))


@pytest.mark.parametrize("config_path", ALL_CONFIGS, ids=[os.path.relpath(p, cases.CONFIG_DIR) for p in ALL_CONFIGS])
def test_every_shipped_config_parses_and_compiles(config_path):
    cfg = RetargetingConfig.load_from_file(config_path)
    opt = cfg._build_optimizer()
    lim = opt.robot.joint_limits[opt.idx_pin2target]
    opt.set_joint_limit(lim)
    cm = opt.compiled_model()
    assert cm.n_opt == len(opt.target_joint_names)
    assert cm.max_joints <= 32
    prob = cases.problem_from_config(os.path.relpath(config_path, cases.CONFIG_DIR))
    assert list(prob.idx_pin2target) == list(opt.idx_pin2target)
    assert list(prob.idx_pin2fixed) == list(opt.idx_pin2fixed)
    assert prob.n_ref == cm.n_ref


def test_default_config_paths_exist():
    for rn in ROBOT_NAMES:
        for rt in RetargetingType:
            for ht in HandType:
                assert get_default_config_path(rn, rt, ht).exists()


def test_dict_config_and_mixed_case_type():
    cfg_dict = {
        "type": "DexPilot", "urdf_path": "allegro_hand/allegro_hand_right.urdf", "wrist_link_name": "wrist",
        "finger_tip_link_names": ["link_15.0_tip", "link_3.0_tip", "link_7.0_tip", "link_11.0_tip"],
        "scaling_factor": 1.6, "low_pass_alpha": 0.2,
    }
    cfg = RetargetingConfig.from_dict(cfg_dict)
    assert cfg.type == "dexpilot"
    opt = cfg._build_optimizer()
    assert isinstance(opt, DexPilotOptimizer) and opt.retargeting_type == "DEXPILOT"
    assert opt.target_link_human_indices.tolist() == [[8, 12, 16, 12, 16, 16, 0, 0, 0, 0], [4, 4, 4, 8, 8, 12, 4, 8, 12, 16]]
    override = RetargetingConfig.from_dict(dict(cfg_dict), override={"scaling_factor": 1.1})
    assert override.scaling_factor == 1.1


def test_config_validation_errors():
    with pytest.raises(ValueError, match="type must be one of"):
        RetargetingConfig(type="nope", urdf_path="allegro_hand/allegro_hand_right.urdf")
    with pytest.raises(ValueError, match="Vector retargeting requires"):
        RetargetingConfig(type="vector", urdf_path="allegro_hand/allegro_hand_right.urdf")
    with pytest.raises(ValueError, match="dim mismatch"):
        RetargetingConfig(type="vector", urdf_path="allegro_hand/allegro_hand_right.urdf",
                          target_origin_link_names=["wrist"], target_task_link_names=["link_3.0_tip", "link_7.0_tip"],
                          target_link_human_indices=np.zeros((2, 1), int))
    with pytest.raises(ValueError, match="does not exist"):
        RetargetingConfig(type="dexpilot", urdf_path="nope.urdf", wrist_link_name="w", finger_tip_link_names=["a", "b"])
    cfg = RetargetingConfig(type="vector", urdf_path="allegro_hand/allegro_hand_right.urdf",
                            target_origin_link_names=["wrist"], target_task_link_names=["not_a_link"],
                            target_link_human_indices=np.zeros((2, 1), int))
    with pytest.raises(ValueError, match="is not a link name"):
        cfg._build_optimizer()
    cfg = RetargetingConfig(type="vector", urdf_path="allegro_hand/allegro_hand_right.urdf",
                            target_joint_names=["no_such_joint"], target_origin_link_names=["wrist"],
                            target_task_link_names=["link_3.0_tip"], target_link_human_indices=np.zeros((2, 1), int))
    with pytest.raises(ValueError, match="does not appear to be in robot XML"):
        cfg._build_optimizer()


@pytest.mark.parametrize("robot_name", ROBOT_NAMES)
def test_add_dummy_free_joint(robot_name):
    """reference tests/test_retargeting_config.py:106-125"""
    path = get_default_config_path(robot_name, RetargetingType.vector, HandType.right)
    base = RetargetingConfig.load_from_file(path)._build_optimizer().robot
    free = RetargetingConfig.load_from_file(path, override={"add_dummy_free_joint": True})._build_optimizer().robot
    assert free.dof == base.dof + 6
    assert free.joint_limits.shape == (base.dof + 6, 2)
    assert all("dummy" in n for n in free.dof_joint_names[:6])
    assert free.dof_joint_names[:6] == DUMMY_JOINT_NAMES


def test_dexpilot_index_tables_known_answers():
    assert DexPilotOptimizer.generate_link_indices(4) == ([2, 3, 4, 3, 4, 4, 0, 0, 0, 0], [1, 1, 1, 2, 2, 3, 1, 2, 3, 4])
    proj, o, t, d = DexPilotOptimizer.set_dexpilot_cache(4, 0.1, 0.2)
    assert not proj.any() and o == [1, 2, 2] and t == [0, 0, 1] and np.allclose(d, [0.1] * 3 + [0.2] * 3)


def test_lpfilter_semantics():
    f = LPFilter(0.25)
    a = f.next(np.array([1.0, 2.0]))
    assert np.allclose(a, [1, 2])
    b = f.next(np.array([2.0, 4.0]))
    assert np.allclose(b, [1.25, 2.5])
    f.reset()
    assert np.allclose(f.next(np.array([5.0])), [5.0])
    frozen = LPFilter(0.0)  # quirk Q5: alpha = 0 freezes the output
    frozen.next(np.array([1.0]))
    assert np.allclose(frozen.next(np.array([9.0])), [1.0])


def test_seq_retargeting_wrapper_bookkeeping(monkeypatch):
    """seq_retarget.py:112-134: clip -> retarget -> carry UNFILTERED qpos -> compose -> mimic fill -> filter."""
    cfg = RetargetingConfig.load_from_file(get_default_config_path(RobotName.inspire, RetargetingType.vector, HandType.right))
    seq = cfg.build()
    opt = seq.optimizer
    calls = []

    def fake_retarget(ref_value, fixed_qpos, last_qpos):
        calls.append((ref_value.dtype, np.array(last_qpos)))
        return (np.array(last_qpos) * 0 + 0.3).astype(np.float32)

    monkeypatch.setattr(opt, "retarget", fake_retarget)
    seq.last_qpos = np.full(6, 99.0, dtype=np.float32)  # out of limits: must be clipped before the solve
    out1 = seq.retarget(np.zeros((5, 3)))
    assert calls[0][0] == np.float32 and np.all(calls[0][1] <= seq.joint_limits[:, 1] + 1e-9)
    assert out1.shape == (opt.robot.dof,) and out1.dtype == np.float64
    ad = opt.adaptor
    assert np.allclose(out1[ad.idx_pin2mimic], out1[ad.idx_pin2source] * ad.multipliers + ad.offsets)
    assert np.allclose(seq.last_qpos, 0.3)
    out2 = seq.retarget(np.zeros((5, 3)))
    assert np.allclose(out2, out1)  # constant solver output -> filter stays put
    assert seq.num_retargeting == 2
    seq.reset()
    assert np.allclose(seq.last_qpos, seq.joint_limits.mean(1))


def test_table_blob_round_trip(tmp_path):
    cfg = RetargetingConfig.load_from_file(get_default_config_path(RobotName.shadow, RetargetingType.dexpilot, HandType.right))
    opt = cfg._build_optimizer()
    opt.set_joint_limit(opt.robot.joint_limits[opt.idx_pin2target])
    cm = opt.compiled_model()
    path = str(tmp_path / "shadow_dexpilot.dexr")
    cm.save(path)
    back = mc.CompiledModel.load(path)
    assert back.to_blob() == cm.to_blob()
    assert (back.kind, back.n_opt, back.n_ref, back.n_comp) == (cm.kind, cm.n_opt, cm.n_ref, cm.n_comp)
    assert int(back.header["n_keypoints"]) == 21 and back.header["human_task"][:3].tolist() == [4, 4, 4]
    with pytest.raises(ValueError):
        mc.CompiledModel.from_blob(cm.to_blob()[:-4])


@pytest.mark.parametrize("rel", ["teleop/inspire_hand_right_dexpilot.yml", "teleop/schunk_svh_hand_right.yml",
                                 "offline/ability_hand_right.yml", "offline/schunk_svh_hand_right.yml",
                                 "teleop/allegro_hand_right_dexpilot.yml", "teleop/panda_gripper.yml"])
def test_reduced_variable_assembly_matches_oracle_model(rel):
    """The reduced-variable formulation of csrc/dexr_red.hpp (mimic joints folded while the Jacobian columns are formed;
    second-order term from running per-variable axis sums; both kinematic chains of a vector term swept separately),
    stated in numpy over the compiled tables (tests/table_interp.reduced_model), reproduces the oracle's gradient and
    full Newton Hessian of the data term in the optimiser's variables."""
    import table_interp as ti
    from oracle import solvers

    seq = RetargetingConfig.load_from_file(os.path.join(cases.CONFIG_DIR, rel)).build()
    prob = cases.problem_from_config(rel)
    cm = seq.optimizer.compiled_model()
    B = 3
    d = cases.human_set(prob, B, seed=5, sigma=0.3)
    x = d["last"].astype(np.float64)
    kw = {}
    targets_all = (d["ref"].astype(np.float32) * np.float32(prob.scaling)).astype(np.float64) if prob.kind == "vector" else d["ref"].astype(np.float64)
    w_all = np.full((B, prob.n_ref), 1.0 / (3 * prob.n_ref) if prob.kind == "position" else 1.0 / prob.n_ref)
    if prob.kind == "dexpilot":
        w, rv, _ = prob.dexpilot_preamble(d["ref"], np.zeros((B, prob.n_pair), bool))
        kw = dict(weights=w, dexpilot_ref=rv)
        targets_all, w_all = rv, w / prob.n_ref
    F, g, H = solvers._model(prob, x, d["ref"], None, x, kw, newton=True)  # last = x: no regulariser in g
    H = H - 2 * prob.norm_delta * np.eye(prob.n_opt)[None]
    for b in range(B):
        gg, HH, FF = np.zeros(prob.n_opt), np.zeros((prob.n_opt, prob.n_opt)), 0.0
        for comp in cm.comps:
            q = ti.joint_values(comp, x=x[b:b + 1], fixed=np.zeros((1, 1)))
            P, axes, orgs = ti.frame_positions(comp, q)
            nt, nv = int(comp["n_term"]), int(comp["n_var"])
            rows = comp["term_ref"][:nt].astype(int)
            f_, g_, H_ = ti.reduced_model(comp, cm.header, P, axes, orgs, targets_all[b][rows], w_all[b][rows])
            api = np.array([int(comp["api"][int(comp["var_joint"][v])]) for v in range(nv)])
            gg[api] += g_
            HH[np.ix_(api, api)] += H_
            FF += f_
        assert abs(FF - F[b]) < 1e-6 * max(1.0, abs(F[b]))
        assert np.abs(gg - g[b]).max() < 2e-6 * max(1.0, np.abs(g[b]).max()), rel   # float32 table entries
        assert np.abs(HH - H[b]).max() < 2e-5 * max(1.0, np.abs(H[b]).max()), rel


# ---- literal mirrors of the reference's tests/test_retargeting_config.py:54-125 ------------------------------------------
def test_dict_config_parsing_like_the_reference():
    import yaml

    from dex_retargeting_amd.seq_retarget import SeqRetargeting

    cfg_str = """
    type: position
    urdf_path: ability_hand/ability_hand_right.urdf
    wrist_link_name: "base_link"

    target_joint_names: ['index_q1', 'middle_q1', 'pinky_q1', 'ring_q1', 'thumb_q1', 'thumb_q2']
    target_link_names: [ "thumb_tip",  "index_tip", "middle_tip", "ring_tip", "pinky_tip" ]

    target_link_human_indices: [ 4, 8, 12, 16, 20 ]

    low_pass_alpha: 1
    """
    retargeting = RetargetingConfig.from_dict(yaml.safe_load(cfg_str)).build()
    assert isinstance(retargeting, SeqRetargeting)
    assert retargeting.optimizer.retargeting_type == "POSITION" and retargeting.optimizer.opt_dof == 6


def test_multi_dict_config_parsing_like_the_reference():
    import yaml

    from dex_retargeting_amd.seq_retarget import SeqRetargeting

    cfg_str = """
    - type: vector
      urdf_path: allegro_hand/allegro_hand_right.urdf
      wrist_link_name: "wrist"

      target_joint_names: null
      target_origin_link_names: [ "wrist", "wrist", "wrist", "wrist" ]
      target_task_link_names: [ "link_15.0_tip", "link_3.0_tip", "link_7.0_tip", "link_11.0_tip" ]
      scaling_factor: 1.6

      # The joint indices of human hand joint which corresponds to each link in the target_link_names
      target_link_human_indices: [ [ 0, 0, 0, 0 ], [ 4, 8, 12, 16 ] ]

      low_pass_alpha: 0.2

    - type: DexPilot
      urdf_path: leap_hand/leap_hand_right.urdf
      wrist_link_name: "base"

      target_joint_names: null
      finger_tip_link_names: [ "thumb_tip_head", "index_tip_head", "middle_tip_head", "ring_tip_head" ]
      scaling_factor: 1.6

      low_pass_alpha: 0.2
    """
    kinds = []
    for cfg_dict in yaml.safe_load(cfg_str):
        retargeting = RetargetingConfig.from_dict(cfg_dict).build()
        assert isinstance(retargeting, SeqRetargeting)
        kinds.append(retargeting.optimizer.retargeting_type)
        # the tables compile on the host (no GPU needed for that): 4 vector components / one DexPilot component
        cm = retargeting.optimizer.compiled_model()
        assert cm.n_ref == (4 if kinds[-1] == "VECTOR" else 10)
    assert kinds == ["VECTOR", "DEXPILOT"]


@pytest.mark.parametrize("robot_name", ROBOT_NAMES)
def test_add_dummy_joint_like_the_reference(robot_name):
    """tests/test_retargeting_config.py:106-125, on the offline (position) configs as there."""
    path = get_default_config_path(robot_name, RetargetingType.position, HandType.right)
    r0 = RetargetingConfig.load_from_file(path, {"add_dummy_free_joint": False}).build()
    dof0, act0 = r0.optimizer.robot.dof, len(r0.optimizer.target_joint_names)
    r1 = RetargetingConfig.load_from_file(path, {"add_dummy_free_joint": True}).build()
    robot = r1.optimizer.robot
    assert robot.dof == dof0 + 6
    assert r1.joint_limits.shape == (act0 + 6, 2)
    assert all("dummy" in n for n in robot.dof_joint_names[:6])


def test_sincos_table_of_the_sixteen_lane_kernel_is_exact_and_the_scheme_accurate():
    """csrc/dexr_math.hpp sincos_f64_tab (round 6): the 32 table rows are the correctly rounded (sin, cos)(k pi / 16), and the
    scheme -- k = rint(16 a / pi), two-constant Cody-Waite remainder, degree-9 / -10 Taylor sums, angle-sum recombination --
    restated here in numpy float64 with the header's own constants stays within 2.5e-16 of numpy's sin / cos on float32-valued
    angles (what the kernel feeds it: joint values are float32)."""
    import re

    src = open(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "dex_retargeting_amd", "csrc", "dexr_math.hpp")).read()
    body = src[src.index("SINCOS_TAB16[32][2] = {"):]
    body = body[:body.index("};")]
    rows = re.findall(r"\{\s*([-+0-9.eE]+),\s*([-+0-9.eE]+)\}", body)
    assert len(rows) == 32
    tab = np.array(rows, dtype=np.float64)
    k = np.arange(32)
    # exact multiples of pi/2 are exact in the table; everywhere else within 1 ulp of numpy's value at the rounded argument
    assert np.abs(tab[:, 0] - np.sin(k * np.pi / 16)).max() < 7e-16 and np.abs(tab[:, 1] - np.cos(k * np.pi / 16)).max() < 7e-16
    assert tab[0, 0] == 0.0 and tab[8, 1] == 0.0 and tab[16, 0] == 0.0 and tab[24, 1] == 0.0 and tab[8, 0] == 1.0 and tab[16, 1] == -1.0
    fn = src[src.index("static __device__ __forceinline__ void sincos_f64_tab"):]
    inv, hi, lo = (float(x) for x in (re.search(r"a \* ([0-9.eE+-]+)\)", fn).group(1), re.search(r"fma\(-kf, ([0-9.eE+-]+), a\)", fn).group(1),
                                      re.search(r"fma\(-kf, ([0-9.eE+-]+), r\)", fn).group(1)))
    assert abs(inv - 16 / np.pi) < 1e-15 and abs(hi + lo - np.pi / 16) < 1e-17
    a = np.random.default_rng(0).uniform(-8, 8, 200000).astype(np.float32).astype(np.float64)
    kf = np.rint(a * inv)
    r = (a - kf * hi) - kf * lo
    kk = kf.astype(np.int64) & 31
    S, C = tab[kk, 0], tab[kk, 1]
    z = r * r
    sr = r + r * z * (-1.6666666666666666e-01 + z * (8.333333333333333e-03 + z * (-1.984126984126984e-04 + z * 2.7557319223985893e-06)))
    cm = z * (-0.5 + z * (4.1666666666666664e-02 + z * (-1.388888888888889e-03 + z * (2.48015873015873e-05 + z * -2.755731922398589e-07))))
    s_, c_ = S + (S * cm + C * sr), C + (C * cm - S * sr)
    assert np.abs(s_ - np.sin(a)).max() < 2.5e-16 and np.abs(c_ - np.cos(a)).max() < 2.5e-16
