"""GPU: the reference's own benchmark script (example/profiling/profile_online_retargeting.py:39-77), staged byte for byte
(stage.py) and run UNMODIFIED -- its main(), its loop, its prints -- with `dex_retargeting` aliased to the drop-in
(run_profile_script.py, in a process of its own): 7 robots x {vector, DexPilot} = 14 "fps" rows over the 621-frame fixture."""
import hashlib
import json
import os
import subprocess
import sys

import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(os.path.dirname(HERE))
ROBOTS = ["allegro_hand", "shadow_hand", "schunk_svh_hand", "leap_hand", "ability_hand", "inspire_hand", "panda_gripper"]


def test_profile_script_is_staged_unmodified():
    import stage

    stage.stage()
    if not stage.profile_script_staged():
        pytest.skip("reference profiling script not staged and /root/reference not here")
    manifest = json.load(open(os.path.join(stage.STAGED, "MANIFEST.json")))
    for name in stage.PROFILE_FILES:
        data = open(os.path.join(stage.PROFILE_DIR, name), "rb").read()
        assert hashlib.sha256(data).hexdigest() == manifest["example/profiling/" + name]["sha256"], name
        if os.path.isdir(stage.REF_PROFILING):
            assert data == open(os.path.join(stage.REF_PROFILING, name), "rb").read(), f"{name} differs from the reference's"


@pytest.mark.gpu
def test_reference_profile_script_runs_unmodified_against_the_drop_in():
    import stage

    stage.stage()
    assert stage.profile_script_staged(), "run __graft_entry__.build() where /root/reference is: the script travels with the tree"
    r = subprocess.run([sys.executable, os.path.join(HERE, "run_profile_script.py"), "--json"], capture_output=True, text=True,
                       timeout=600, cwd=REPO)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = r.stdout.splitlines()
    assert lines[0].startswith("Being retargeting profiling with a trajectory of 621 hand poses.")  # (the script's own first line)
    rec = json.loads([l for l in lines if l.startswith("REFERENCE_PROFILE_SCRIPT ")][0].split(" ", 1)[1])
    rows = rec["rows"]
    assert [(x["kind"], x["robot"]) for x in rows] == [("vector", n) for n in ROBOTS] + [("dexpilot", n) for n in ROBOTS]
    for x in rows:
        assert x["fps"] > 0 and abs(x["fps"] * x["seconds"] - 621) < 1e-6 * 621
    out = os.path.join(REPO, "gpurun_out")
    os.makedirs(out, exist_ok=True)
    with open(os.path.join(out, "reference_profile_script.txt"), "w") as f:
        f.write("# /root/reference/example/profiling/profile_online_retargeting.py, unmodified, `dex_retargeting` = the drop-in\n")
        f.write("\n".join(l for l in lines if not l.startswith("REFERENCE_PROFILE_SCRIPT ")) + "\n")
