"""Guards around the reference's own test files (see stage.py): they must be present when the suite runs, be the
reference's bytes, and import the drop-in -- not a second copy of it -- under the name `dex_retargeting`."""
import hashlib
import json
import os
import sys

import pytest

HERE = os.path.dirname(os.path.abspath(__file__))


def test_reference_tests_are_staged_unmodified():
    import stage

    if not stage.stage():  # (fresh clone away from the reference: nothing to guard -- the _ref tests are then not collected either)
        pytest.skip("tests/reference_suite/_ref/ is empty and /root/reference is not here: run `python tests/reference_suite/stage.py` "
                    "(or __graft_entry__.build()) where the reference is")
    manifest = json.load(open(os.path.join(stage.STAGED, "MANIFEST.json")))
    for name in stage.FILES:
        data = open(os.path.join(stage.STAGED, name), "rb").read()
        assert hashlib.sha256(data).hexdigest() == manifest[name]["sha256"], name
        if os.path.isdir(stage.REF_TESTS):
            assert data == open(os.path.join(stage.REF_TESTS, name), "rb").read(), f"{name} differs from the reference's"


def test_dex_retargeting_resolves_to_the_drop_in():
    import dex_retargeting
    import dex_retargeting_amd
    from dex_retargeting.constants import RobotName
    from dex_retargeting.optimizer import VectorOptimizer
    from dex_retargeting_amd.constants import RobotName as RobotName2
    from dex_retargeting_amd.optimizer import VectorOptimizer as VectorOptimizer2

    assert dex_retargeting is dex_retargeting_amd
    assert RobotName is RobotName2 and VectorOptimizer is VectorOptimizer2  # one module object each, not a re-import
    assert "dex_retargeting.seq_retarget" in sys.modules


def test_layout_the_reference_tests_expect():
    for rel in ("assets/robots/hands/allegro_hand/allegro_hand_right.urdf", "dex_retargeting/configs/teleop/allegro_hand_right.yml",
                "src/dex_retargeting/configs/offline/leap_hand_right.yml"):
        assert os.path.exists(os.path.join(HERE, rel)), rel
