#!/usr/bin/env python3
"""Author the URDF kinematic fixtures shipped under dex_retargeting_amd/assets/.

The reference expects its robots in the `dex-urdf` git submodule
(/root/reference/.gitmodules:1-4, tests/test_optimizer.py:22-23), which is not
checked out in this environment.  The fixtures written here are AUTHORED FROM
PUBLIC KINEMATIC SPECS (approximate link lengths / joint limits), not copied
from dex-urdf.  They only have to satisfy the names and topology that the
reference's YAML problem definitions require (configs/{teleop,offline}/*.yml):
link names, joint names, mimic structure, DoF counts.

Left hands are produced by mirroring the right hand across the XZ plane
(y -> -y): origin xyz -> (x,-y,z), rpy -> (-r,p,-y), axis -> (-ax,ay,-az).

Usage: python tools/make_fixtures.py   (rewrites the .urdf files in place)
"""
from __future__ import annotations

import math
import os
from typing import List, Optional, Tuple

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..",
                    "dex_retargeting_amd", "assets", "robots", "hands")

# joint tuple: (name, type, parent, child, xyz, rpy, axis, lower, upper, mimic)
# mimic = None | (source_joint, multiplier, offset)
J = Tuple[str, str, str, str, Tuple[float, float, float], Tuple[float, float, float],
          Tuple[float, float, float], float, float, Optional[Tuple[str, float, float]]]


def rev(name, parent, child, xyz, axis, lo, hi, rpy=(0, 0, 0), mimic=None) -> J:
    return (name, "revolute", parent, child, tuple(xyz), tuple(rpy), tuple(axis), lo, hi, mimic)


def pri(name, parent, child, xyz, axis, lo, hi, rpy=(0, 0, 0), mimic=None) -> J:
    return (name, "prismatic", parent, child, tuple(xyz), tuple(rpy), tuple(axis), lo, hi, mimic)


def fix(name, parent, child, xyz, rpy=(0, 0, 0)) -> J:
    return (name, "fixed", parent, child, tuple(xyz), tuple(rpy), (1, 0, 0), 0.0, 0.0, None)


# --------------------------------------------------------------------------- #
# Allegro hand (16 DoF): 4 fingers x (yaw + 3 flexion), thumb mounted sideways.
# --------------------------------------------------------------------------- #
def allegro() -> Tuple[str, List[J]]:
    js: List[J] = [fix("wrist_to_base", "wrist", "base_link", (0, 0, 0.095))]
    lim = [(-0.47, 0.47), (-0.196, 1.61), (-0.174, 1.709), (-0.227, 1.618)]
    bases = [((0, 0.0435, -0.001542), (-0.08726646255, 0, 0)),
             ((0, 0, 0.0007), (0, 0, 0)),
             ((0, -0.0435, -0.001542), (0.08726646255, 0, 0))]
    for f, (xyz, rpy) in enumerate(bases):
        b = 4 * f
        js += [
            rev(f"joint_{b}.0", "base_link", f"link_{b}.0", xyz, (0, 0, 1), *lim[0], rpy=rpy),
            rev(f"joint_{b+1}.0", f"link_{b}.0", f"link_{b+1}.0", (0, 0, 0.0164), (0, 1, 0), *lim[1]),
            rev(f"joint_{b+2}.0", f"link_{b+1}.0", f"link_{b+2}.0", (0, 0, 0.054), (0, 1, 0), *lim[2]),
            rev(f"joint_{b+3}.0", f"link_{b+2}.0", f"link_{b+3}.0", (0, 0, 0.0384), (0, 1, 0), *lim[3]),
            fix(f"joint_{b+3}.0_tip", f"link_{b+3}.0", f"link_{b+3}.0_tip", (0, 0, 0.0267)),
        ]
    js += [
        rev("joint_12.0", "base_link", "link_12.0", (-0.0182, 0.019333, -0.045987), (-1, 0, 0),
            0.263, 1.396, rpy=(0, -1.65806278845, -1.5707963259)),
        rev("joint_13.0", "link_12.0", "link_13.0", (-0.027, 0.005, 0.0399), (0, 0, 1), -0.105, 1.163),
        rev("joint_14.0", "link_13.0", "link_14.0", (0, 0, 0.0177), (0, 1, 0), -0.189, 1.644),
        rev("joint_15.0", "link_14.0", "link_15.0", (0, 0, 0.0514), (0, 1, 0), -0.162, 1.719),
        fix("joint_15.0_tip", "link_15.0", "link_15.0_tip", (0, 0, 0.0423)),
    ]
    return "allegro_hand", js


# --------------------------------------------------------------------------- #
# Shadow hand (24 DoF): 2 wrist + FF/MF/RF (4) + LF (5) + TH (5).
# --------------------------------------------------------------------------- #
def shadow() -> Tuple[str, List[J]]:
    js: List[J] = [
        rev("WRJ2", "forearm", "wrist", (0, -0.010, 0.213), (0, 1, 0), -0.524, 0.175),
        rev("WRJ1", "wrist", "palm", (0, 0, 0.034), (1, 0, 0), -0.698, 0.489),
        fix("ee_fixed_joint", "palm", "ee_link", (0, 0, 0.05)),
    ]

    def finger(p, knuckle_parent, xyz, j4_axis):
        return [
            rev(f"{p.upper()}J4", knuckle_parent, f"{p}knuckle", xyz, j4_axis, -0.349, 0.349),
            rev(f"{p.upper()}J3", f"{p}knuckle", f"{p}proximal", (0, 0, 0), (1, 0, 0), -0.262, 1.571),
            rev(f"{p.upper()}J2", f"{p}proximal", f"{p}middle", (0, 0, 0.045), (1, 0, 0), 0.0, 1.571),
            rev(f"{p.upper()}J1", f"{p}middle", f"{p}distal", (0, 0, 0.025), (1, 0, 0), 0.0, 1.571),
            fix(f"{p.upper()}tip", f"{p}distal", f"{p}tip", (0, 0, 0.026)),
        ]

    js += finger("ff", "palm", (0.033, 0, 0.095), (0, -1, 0))
    js += finger("mf", "palm", (0.011, 0, 0.099), (0, -1, 0))
    js += finger("rf", "palm", (-0.011, 0, 0.095), (0, 1, 0))
    js += [rev("LFJ5", "palm", "lfmetacarpal", (-0.033, 0, 0.02071), (0.573576, 0, 0.819152), 0.0, 0.785)]
    js += finger("lf", "lfmetacarpal", (0, 0, 0.06579), (0, 1, 0))
    js += [
        rev("THJ5", "palm", "thbase", (0.034, -0.0085, 0.029), (0, 0, -1), -1.047, 1.047,
            rpy=(0, 0.785398, 0)),
        rev("THJ4", "thbase", "thproximal", (0, 0, 0), (1, 0, 0), 0.0, 1.222),
        rev("THJ3", "thproximal", "thhub", (0, 0, 0.038), (1, 0, 0), -0.209, 0.209),
        rev("THJ2", "thhub", "thmiddle", (0, 0, 0), (0, -1, 0), -0.698, 0.698),
        rev("THJ1", "thmiddle", "thdistal", (0, 0, 0.032), (0, -1, 0), 0.0, 1.571),
        fix("THtip", "thdistal", "thtip", (0, 0, 0.0275)),
    ]
    return "shadow_hand", js


# --------------------------------------------------------------------------- #
# LEAP hand (16 DoF).  Joint names are the motor ids "0".."15"; within a finger
# the kinematic order is (1,0,2,3), (5,4,6,7), (9,8,10,11), thumb (12,13,14,15).
# --------------------------------------------------------------------------- #
def leap() -> Tuple[str, List[J]]:
    js: List[J] = []
    sfx = ["", "_2", "_3"]
    ybase = [0.0454, 0.0, -0.0454]
    for f in range(3):
        s, b = sfx[f], 4 * f
        js += [
            rev(f"{b+1}", "base", f"mcp_joint{s}", (-0.0070, ybase[f], 0.0230), (0, 1, 0), -0.314, 2.23),
            rev(f"{b}", f"mcp_joint{s}", f"pip{s}", (0.0, 0.0, 0.0120), (1, 0, 0), -1.047, 1.047),
            rev(f"{b+2}", f"pip{s}", f"dip{s}", (0.0, 0.0, 0.0360), (0, 1, 0), -0.506, 1.885),
            rev(f"{b+3}", f"dip{s}", f"fingertip{s}", (0.0, 0.0, 0.0360), (0, 1, 0), -0.366, 2.042),
        ]
    js += [
        fix("index_tip", "fingertip", "index_tip_head", (0, 0, 0.0480)),
        fix("middle_tip", "fingertip_2", "middle_tip_head", (0, 0, 0.0480)),
        fix("ring_tip", "fingertip_3", "ring_tip_head", (0, 0, 0.0480)),
        rev("12", "base", "pip_4", (-0.0693, 0.0495, -0.0012), (0, 0, 1), -0.349, 2.094,
            rpy=(0, 1.5707963, 0)),
        rev("13", "pip_4", "thumb_pip", (0, 0.0143, -0.013), (0, 1, 0), -0.47, 2.443),
        rev("14", "thumb_pip", "thumb_dip", (0, 0.0145, -0.017), (1, 0, 0), -1.20, 1.90),
        rev("15", "thumb_dip", "thumb_fingertip", (0, 0.0466, 0.0002), (1, 0, 0), -1.34, 1.88),
        fix("thumb_tip", "thumb_fingertip", "thumb_tip_head", (0, 0.0600, 0.0)),
    ]
    return "leap_hand", js


# --------------------------------------------------------------------------- #
# PSYONIC Ability hand: 6 actuated + 4 mimic (finger q2 = a*q1 + b).
# --------------------------------------------------------------------------- #
def ability() -> Tuple[str, List[J]]:
    js: List[J] = []
    a, b = 1.05851325, 0.72349796
    fingers = [("index", (0.0280, 0.0097, 0.0935), 0.08), ("middle", (0.0090, 0.0100, 0.0978), 0.0),
               ("ring", (-0.0100, 0.0095, 0.0935), -0.08), ("pinky", (-0.0290, 0.0085, 0.0862), -0.16)]
    for name, xyz, splay in fingers:
        js += [
            rev(f"{name}_q1", "base_link", f"{name}_L1", xyz, (1, 0, 0), 0.0, 2.0944, rpy=(0, splay, 0)),
            rev(f"{name}_q2", f"{name}_L1", f"{name}_L2", (0, 0.0, 0.0385), (1, 0, 0), 0.0, 2.6586,
                mimic=(f"{name}_q1", a, b)),
            fix(f"{name}_tip_joint", f"{name}_L2", f"{name}_tip", (0, -0.012, 0.0330)),
        ]
    js += [
        rev("thumb_q1", "base_link", "thumb_L1", (0.0240, 0.0070, 0.0330), (0, 0, 1), -2.0944, 0.0,
            rpy=(0, 0.35, 0)),
        rev("thumb_q2", "thumb_L1", "thumb_L2", (0.0278, 0.0, 0.0148), (0, 1, 0), 0.0, 2.0944),
        fix("thumb_tip_joint", "thumb_L2", "thumb_tip", (0.0650, 0.0, 0.0150)),
    ]
    return "ability_hand", js


# --------------------------------------------------------------------------- #
# Inspire hand: 6 actuated + 6 mimic.
# --------------------------------------------------------------------------- #
def inspire() -> Tuple[str, List[J]]:
    js: List[J] = [fix("base_joint", "base", "hand_base_link", (0, 0, 0))]
    fingers = [("index", (0.0320, 0.0, 0.1360)), ("middle", (0.0130, 0.0, 0.1400)),
               ("ring", (-0.0060, 0.0, 0.1360)), ("pinky", (-0.0250, 0.0, 0.1300))]
    for name, xyz in fingers:
        js += [
            rev(f"{name}_proximal_joint", "hand_base_link", f"{name}_proximal", xyz, (1, 0, 0), 0.0, 1.47),
            rev(f"{name}_intermediate_joint", f"{name}_proximal", f"{name}_intermediate", (0, -0.002, 0.032),
                (1, 0, 0), 0.0, 1.56, mimic=(f"{name}_proximal_joint", 1.06399, 0.0)),
            fix(f"{name}_tip_joint", f"{name}_intermediate", f"{name}_tip", (0, -0.004, 0.0450)),
        ]
    js += [
        rev("thumb_proximal_yaw_joint", "hand_base_link", "thumb_proximal_base", (0.0270, -0.010, 0.0690),
            (0, 0, -1), -0.1, 1.3),
        rev("thumb_proximal_pitch_joint", "thumb_proximal_base", "thumb_proximal", (0.0110, 0.0, 0.0044),
            (0, 1, 0), 0.0, 0.5),
        rev("thumb_intermediate_joint", "thumb_proximal", "thumb_intermediate", (0.0440, 0.0, 0.0030),
            (0, 1, 0), 0.0, 0.8, mimic=("thumb_proximal_pitch_joint", 1.6, 0.0)),
        rev("thumb_distal_joint", "thumb_intermediate", "thumb_distal", (0.0200, 0.0, 0.0010),
            (0, 1, 0), 0.0, 1.2, mimic=("thumb_proximal_pitch_joint", 2.4, 0.0)),
        fix("thumb_tip_joint", "thumb_distal", "thumb_tip", (0.0250, 0.0, 0.0020)),
    ]
    return "inspire_hand", js


# --------------------------------------------------------------------------- #
# Schunk SVH: 9 actuated + 11 mimic.  Link letters follow the public
# schunk_svh_description naming (z,a,b,c thumb; l,p,t index; k,o,s middle;
# j,n,r ring; i,m,q pinky).
# --------------------------------------------------------------------------- #
def svh(side: str) -> Tuple[str, List[J]]:
    P = f"{side}_hand_"
    js: List[J] = [
        rev(P + "Thumb_Opposition", P + "base_link", P + "z", (-0.0169, 0.0200, 0.0450), (0, 0, -1), 0.0, 0.9879),
        rev(P + "Thumb_Flexion", P + "z", P + "a", (0, 0, 0.04596), (1, 0, 0), 0.0, 0.9704,
            rpy=(0, 0, -0.9704)),
        rev(P + "j3", P + "a", P + "b", (0, 0, 0.0485), (1, 0, 0), 0.0, 0.98506,
            mimic=(P + "Thumb_Flexion", 1.01511, 0.0)),
        rev(P + "j4", P + "b", P + "c", (0, 0, 0.030), (1, 0, 0), 0.0, 1.406,
            mimic=(P + "Thumb_Flexion", 1.44889, 0.0)),
        fix(P + "thtip_joint", P + "c", "thtip", (0, 0, 0.0275)),
        # index
        rev(P + "index_spread", P + "base_link", P + "virtual_l", (-0.025, 0.0, 0.110), (0, 1, 0), 0.0, 0.28833,
            mimic=(P + "Finger_Spread", 0.5, 0.0)),
        rev(P + "Index_Finger_Proximal", P + "virtual_l", P + "l", (0, 0, 0), (1, 0, 0), 0.0, 0.79849),
        rev(P + "Index_Finger_Distal", P + "l", P + "p", (0, 0, 0.04804), (1, 0, 0), 0.0, 1.334),
        rev(P + "j14", P + "p", P + "t", (0, 0, 0.026), (1, 0, 0), 0.0, 1.394,
            mimic=(P + "Index_Finger_Distal", 1.0450, 0.0)),
        fix(P + "fftip_joint", P + "t", "fftip", (0, 0, 0.0180)),
        # middle
        rev(P + "Middle_Finger_Proximal", P + "base_link", P + "k", (-0.003, 0.0, 0.115), (1, 0, 0), 0.0, 0.79849),
        rev(P + "Middle_Finger_Distal", P + "k", P + "o", (0, 0, 0.05004), (1, 0, 0), 0.0, 1.334),
        rev(P + "j15", P + "o", P + "s", (0, 0, 0.032), (1, 0, 0), 0.0, 1.334,
            mimic=(P + "Middle_Finger_Distal", 1.0454, 0.0)),
        fix(P + "mftip_joint", P + "s", "mftip", (0, 0, 0.0200)),
        # palm arch + ring + pinky
        rev(P + "j5", P + "base_link", P + "e2", (0.0184, 0.006, 0.0375), (0, 0, 1), 0.0, 0.98786,
            mimic=(P + "Thumb_Opposition", 1.0, 0.0)),
        rev(P + "ring_spread", P + "e2", P + "virtual_j", (0.003855, -0.006, 0.0655), (0, 1, 0), 0.0, 0.28833,
            mimic=(P + "Finger_Spread", 0.5, 0.0), rpy=(0, 0, 3.14159)),
        rev(P + "Ring_Finger", P + "virtual_j", P + "j", (0, 0, 0), (1, 0, 0), 0.0, 0.98175),
        rev(P + "j12", P + "j", P + "n", (0, 0, 0.05004), (1, 0, 0), 0.0, 1.334,
            mimic=(P + "Ring_Finger", 1.3588, 0.0)),
        rev(P + "j16", P + "n", P + "r", (0, 0, 0.032), (1, 0, 0), 0.0, 1.395,
            mimic=(P + "Ring_Finger", 1.42093, 0.0)),
        fix(P + "rftip_joint", P + "r", "rftip", (0, 0, 0.0180)),
        rev(P + "Finger_Spread", P + "e2", P + "virtual_i", (0.025355, -0.006, 0.056), (0, 1, 0), 0.0, 0.5829,
            rpy=(0, 0, 3.14159)),
        rev(P + "Pinky", P + "virtual_i", P + "i", (0, 0, 0), (1, 0, 0), 0.0, 0.98175),
        rev(P + "j13", P + "i", P + "m", (0, 0, 0.04454), (1, 0, 0), 0.0, 1.334,
            mimic=(P + "Pinky", 1.35880, 0.0)),
        rev(P + "j17", P + "m", P + "q", (0, 0, 0.022), (1, 0, 0), 0.0, 1.3971,
            mimic=(P + "Pinky", 1.42307, 0.0)),
        fix(P + "lftip_joint", P + "q", "lftip", (0, 0, 0.0150)),
    ]
    return "schunk_hand", js


# --------------------------------------------------------------------------- #
# Panda gripper: 1 actuated prismatic + 1 mimic prismatic.
# --------------------------------------------------------------------------- #
def panda() -> Tuple[str, List[J]]:
    js: List[J] = [
        pri("panda_finger_joint1", "panda_hand", "panda_leftfinger", (0, 0, 0.0584), (0, 1, 0), 0.0, 0.04),
        pri("panda_finger_joint2", "panda_hand", "panda_rightfinger", (0, 0, 0.0584), (0, -1, 0), 0.0, 0.04,
            mimic=("panda_finger_joint1", 1.0, 0.0)),
    ]
    return "panda_gripper", js


# --------------------------------------------------------------------------- #
def mirror(js: List[J]) -> List[J]:
    out = []
    for (name, typ, parent, child, xyz, rpy, axis, lo, hi, mimic) in js:
        out.append((name, typ, parent, child, (xyz[0], -xyz[1], xyz[2]), (-rpy[0], rpy[1], -rpy[2]),
                    (-axis[0], axis[1], -axis[2]) if typ == "revolute" else (axis[0], -axis[1], axis[2]),
                    lo, hi, mimic))
    return out


def fmt(v) -> str:
    return " ".join(repr(float(x)) if x != 0 else "0" for x in v)


def write_urdf(path: str, robot_name: str, js: List[J]):
    links: List[str] = []
    for j in js:
        for l in (j[2], j[3]):
            if l not in links:
                links.append(l)
    lines = ['<?xml version="1.0"?>',
             "<!-- AUTHORED FIXTURE (tools/make_fixtures.py): approximate public kinematics, "
             "NOT a copy of dex-urdf. Kinematics only: no meshes, no inertias. -->",
             f'<robot name="{robot_name}">']
    for l in links:
        lines.append(f'  <link name="{l}"/>')
    for (name, typ, parent, child, xyz, rpy, axis, lo, hi, mimic) in js:
        lines.append(f'  <joint name="{name}" type="{typ}">')
        lines.append(f'    <parent link="{parent}"/>')
        lines.append(f'    <child link="{child}"/>')
        lines.append(f'    <origin xyz="{fmt(xyz)}" rpy="{fmt(rpy)}"/>')
        if typ != "fixed":
            lines.append(f'    <axis xyz="{fmt(axis)}"/>')
            lines.append(f'    <limit lower="{lo!r}" upper="{hi!r}" effort="10" velocity="3.14"/>')
        if mimic is not None:
            lines.append(f'    <mimic joint="{mimic[0]}" multiplier="{mimic[1]!r}" offset="{mimic[2]!r}"/>')
        lines.append("  </joint>")
    lines.append("</robot>")
    os.makedirs(os.path.dirname(path), exist_ok=True)
    with open(path, "w") as f:
        f.write("\n".join(lines) + "\n")


def arm_shadow() -> List[J]:
    """TEST fixture (tests/urdf/arm_shadow_hand_right.urdf): a 7-DoF arm carrying the Shadow hand fixture = 31 movable
    joints in ONE kinematic chain family (37 with the 6 dummy free joints): more than the fixed-size component tables
    hold, i.e. a model only the general kernel serves.  Arm geometry: alternating roll / pitch joints, authored."""
    arm = [
        rev("arm_j1", "arm_base", "arm_l1", (0, 0, 0.15), (0, 0, 1), -2.9, 2.9),
        rev("arm_j2", "arm_l1", "arm_l2", (0, 0, 0.18), (0, 1, 0), -1.8, 1.8),
        rev("arm_j3", "arm_l2", "arm_l3", (0, 0, 0.20), (0, 0, 1), -2.9, 2.9, rpy=(0, 0, 0.3)),
        rev("arm_j4", "arm_l3", "arm_l4", (0.03, 0, 0.20), (0, -1, 0), -0.1, 2.6),
        rev("arm_j5", "arm_l4", "arm_l5", (-0.03, 0, 0.19), (0, 0, 1), -2.9, 2.9),
        rev("arm_j6", "arm_l5", "arm_l6", (0, 0, 0.19), (0, 1, 0), -1.9, 1.9),
        rev("arm_j7", "arm_l6", "arm_l7", (0, 0, 0.08), (0, 0, 1), -3.0, 3.0),
        fix("arm_to_hand", "arm_l7", "forearm", (0, 0, 0.02), rpy=(0, 0, 1.5707963267948966)),
    ]
    return arm + shadow()[1]


def main():
    tests_urdf = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests", "urdf")
    write_urdf(os.path.join(tests_urdf, "arm_shadow_hand_right.urdf"), "arm_shadow_hand_right", arm_shadow())
    for fn in (allegro, shadow, leap, ability, inspire):
        d, js = fn()
        write_urdf(os.path.join(ROOT, d, f"{d}_right.urdf"), f"{d}_right", js)
        write_urdf(os.path.join(ROOT, d, f"{d}_left.urdf"), f"{d}_left", mirror(js))
    d, js = svh("right")
    write_urdf(os.path.join(ROOT, d, "schunk_svh_hand_right.urdf"), "schunk_svh_hand_right", js)
    d, js = svh("left")
    write_urdf(os.path.join(ROOT, d, "schunk_svh_hand_left.urdf"), "schunk_svh_hand_left", mirror(js))
    d, js = panda()
    write_urdf(os.path.join(ROOT, d, "panda_gripper_glb.urdf"), "panda_gripper", js)


if __name__ == "__main__":
    main()
