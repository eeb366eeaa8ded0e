"""Runs the reference's own tests (tests/reference_suite/_ref/, staged unmodified by stage.py) against the drop-in:
`dex_retargeting` and every `dex_retargeting.<module>` the tests import are aliased to `dex_retargeting_amd` in
sys.modules BEFORE collection, and the directory layout the tests expect is created.  test_optimizer.py solves on the
GPU (marked `gpu` here); test_retargeting_config.py only builds objects (host side: runs in the CPU suite too)."""
import importlib
import os
import pkgutil
import sys

import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(os.path.dirname(HERE))
if REPO not in sys.path:
    sys.path.insert(0, REPO)
sys.path.insert(0, HERE)

import stage as _stage  # noqa: E402


def alias_package():
    import dex_retargeting_amd as pkg

    sys.modules["dex_retargeting"] = pkg
    for m in pkgutil.iter_modules(pkg.__path__):
        if m.name.startswith("_") or m.ispkg or not os.path.exists(os.path.join(pkg.__path__[0], m.name + ".py")):
            continue  # python modules only (libdexr.so sits in the package directory too)
        sys.modules[f"dex_retargeting.{m.name}"] = importlib.import_module(f"dex_retargeting_amd.{m.name}")


alias_package()
_stage.lay_out()
HAVE_REF_TESTS = _stage.stage()


def pytest_collection_modifyitems(config, items):
    for item in items:
        p = str(item.fspath)
        if os.sep + "_ref" + os.sep in p and os.path.basename(p) == "test_optimizer.py":
            item.add_marker(pytest.mark.gpu)
