#!/usr/bin/env python3
"""Write the per-robot retargeting problem definitions (YAML) under dex_retargeting_amd/configs/.

Same schema, file names and values as the reference's input specs
(/root/reference/src/dex_retargeting/configs/{teleop,offline}/*.yml) so that
``get_default_config_path`` + ``RetargetingConfig.load_from_file`` behave identically.  The files are data
(link/joint names, human keypoint indices, scaling, low-pass alpha), regenerated from the tables below.
"""
import os

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "dex_retargeting_amd", "configs")

TIP4, MID4 = [4, 8, 12, 16], [2, 6, 10, 14]
TIP5, MID5 = [4, 8, 12, 16, 20], [2, 6, 10, 14, 18]


def svh_joints(s):
    return [f"{s}_hand_" + n for n in ["Thumb_Opposition", "Thumb_Flexion", "Index_Finger_Proximal",
                                      "Index_Finger_Distal", "Finger_Spread", "Pinky", "Ring_Finger",
                                      "Middle_Finger_Proximal", "Middle_Finger_Distal"]]


ABILITY_J = ["thumb_q1", "thumb_q2", "index_q1", "middle_q1", "pinky_q1", "ring_q1"]
INSPIRE_J = ["pinky_proximal_joint", "ring_proximal_joint", "middle_proximal_joint", "index_proximal_joint",
             "thumb_proximal_pitch_joint", "thumb_proximal_yaw_joint"]
TIPS5 = ["thumb_tip", "index_tip", "middle_tip", "ring_tip", "pinky_tip"]
SH_TIPS = ["thtip", "fftip", "mftip", "rftip", "lftip"]
SH_MID = ["thmiddle", "ffmiddle", "mfmiddle", "rfmiddle", "lfmiddle"]
LEAP_TIPS = ["thumb_tip_head", "index_tip_head", "middle_tip_head", "ring_tip_head"]
LEAP_DIP = ["thumb_dip", "dip", "dip_2", "dip_3"]


def allegro_tips(side):
    return ["link_15.0_tip", "link_3.0_tip", "link_7.0_tip", "link_11.0_tip"] if side == "right" else \
        ["link_15.0_tip", "link_11.0_tip", "link_7.0_tip", "link_3.0_tip"]


def allegro_mid(side):
    return ["link_14.0", "link_2.0", "link_6.0", "link_10.0"] if side == "right" else \
        ["link_14.0", "link_10.0", "link_6.0", "link_2.0"]


def hands(side):
    """name -> dict(urdf, joints(None=all), wrist, tips, mids(or None), scaling, human tip idx, human mid idx,
    offline extras)"""
    return {
        "allegro_hand": dict(urdf=f"allegro_hand/allegro_hand_{side}.urdf", joints=None, wrist="wrist",
                             tips=allegro_tips(side), scaling=1.6, htips=TIP4,
                             off_links=allegro_tips(side) + allegro_mid(side), off_idx=TIP4 + MID4),
        "shadow_hand": dict(urdf=f"shadow_hand/shadow_hand_{side}.urdf", joints=None, wrist="palm",
                            dp_wrist="ee_link", tips=SH_TIPS, vec_extra=SH_MID, scaling=1.2, htips=TIP5,
                            hextra=MID5, off_links=SH_TIPS + SH_MID, off_idx=TIP5 + MID5),
        "leap_hand": dict(urdf=f"leap_hand/leap_hand_{side}.urdf", joints=None, wrist="base", tips=LEAP_TIPS,
                          scaling=1.6, htips=TIP4, off_links=LEAP_TIPS + LEAP_DIP, off_idx=TIP4 + MID4),
        "ability_hand": dict(urdf=f"ability_hand/ability_hand_{side}.urdf", joints=ABILITY_J, wrist="base_link",
                             tips=TIPS5, scaling=1.0, htips=TIP5, off_links=TIPS5, off_idx=TIP5, mimic=True),
        "inspire_hand": dict(urdf=f"inspire_hand/inspire_hand_{side}.urdf", joints=INSPIRE_J, wrist="base",
                             tips=TIPS5, scaling=1.15, htips=TIP5, off_links=TIPS5, off_idx=TIP5, mimic=True),
        "schunk_svh_hand": dict(urdf=f"schunk_hand/schunk_svh_hand_{side}.urdf", joints=svh_joints(side),
                                wrist=f"{side}_hand_base_link", tips=SH_TIPS, scaling=1.2, htips=TIP5,
                                off_links=[f"{side}_hand_{c}" for c in "ctsrqbponi"], off_idx=TIP5 + MID5),
    }


def flow(lst):
    return "[ " + ", ".join(f'"{v}"' if isinstance(v, str) else str(v) for v in lst) + " ]"


def write(path, lines):
    os.makedirs(os.path.dirname(path), exist_ok=True)
    with open(path, "w") as f:
        f.write("retargeting:\n" + "\n".join("  " + l for l in lines) + "\n")


def emit(name, fname_side, h):
    tj = [] if h["joints"] is None else [f"target_joint_names: {flow(h['joints'])}"]
    vec_links = h["tips"] + h.get("vec_extra", [])
    vec_idx = h["htips"] + h.get("hextra", [])
    write(f"{ROOT}/teleop/{name}{fname_side}.yml", [
        "type: vector", f"urdf_path: {h['urdf']}", *tj,
        f"target_origin_link_names: {flow([h['wrist']] * len(vec_links))}",
        f"target_task_link_names: {flow(vec_links)}",
        f"scaling_factor: {h['scaling']}",
        f"target_link_human_indices: [ {flow([0] * len(vec_links))}, {flow(vec_idx)} ]",
        "low_pass_alpha: 0.2"])
    write(f"{ROOT}/teleop/{name}{fname_side}_dexpilot.yml", [
        "type: DexPilot", f"urdf_path: {h['urdf']}", *tj,
        f'wrist_link_name: "{h.get("dp_wrist", h["wrist"])}"',
        f"finger_tip_link_names: {flow(h['tips'])}",
        f"scaling_factor: {h['scaling']}", "low_pass_alpha: 0.2"])
    tjo = ["target_joint_names: null"] if h["joints"] is None else tj
    write(f"{ROOT}/offline/{name}{fname_side}.yml", [
        "type: position", f"urdf_path: {h['urdf']}", *tjo,
        f"target_link_names: {flow(h['off_links'])}",
        f"target_link_human_indices: {flow(h['off_idx'])}",
        "add_dummy_free_joint: True", "low_pass_alpha: 1",
        *(["ignore_mimic_joint: False"] if h.get("mimic") else [])])


def main():
    for side in ("right", "left"):
        for name, h in hands(side).items():
            emit(name, f"_{side}", h)
    panda = "panda_gripper/panda_gripper_glb.urdf"
    write(f"{ROOT}/teleop/panda_gripper.yml", [
        "type: vector", f"urdf_path: {panda}", 'target_joint_names: [ "panda_finger_joint1" ]',
        'target_origin_link_names: [ "panda_leftfinger" ]', 'target_task_link_names: [ "panda_rightfinger" ]',
        "scaling_factor: 1.5", "target_link_human_indices: [ [ 4 ], [ 8 ] ]", "low_pass_alpha: 0.2"])
    write(f"{ROOT}/teleop/panda_gripper_dexpilot.yml", [
        "type: DexPilot", f"urdf_path: {panda}", 'target_joint_names: [ "panda_finger_joint1" ]',
        'wrist_link_name: "panda_hand"', 'finger_tip_link_names: [ "panda_leftfinger", "panda_rightfinger" ]',
        "scaling_factor: 1.5", "low_pass_alpha: 0.2"])
    write(f"{ROOT}/offline/panda_gripper.yml", [
        "type: position", f"urdf_path: {panda}", 'target_joint_names: [ "panda_finger_joint1" ]',
        'target_link_names: [ "panda_leftfinger", "panda_rightfinger" ]', "target_link_human_indices: [ 4, 8 ]",
        "add_dummy_free_joint: True", "low_pass_alpha: 1", "ignore_mimic_joint: False"])


if __name__ == "__main__":
    main()
