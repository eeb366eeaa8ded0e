import os
import sys

import pytest

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if REPO not in sys.path:
    sys.path.insert(0, REPO)
sys.path.insert(0, os.path.join(REPO, "tests"))


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: test needs an MI355X (run with -m gpu on the GPU box)")


def pytest_sessionstart(session):
    """The tests exercise the in-tree libdexr.so (the only compute path).  It is normally built beforehand by
    __graft_entry__.build(); if it is not there -- fresh checkout, built artefacts are not in git -- build it now
    (hipcc cross-compiles gfx950 without a GPU, ~3 min) rather than failing every test."""
    from dex_retargeting_amd import _build, _lib

    if not os.path.exists(_lib.LIB_PATH):
        print(f"\n[conftest] {_lib.LIB_PATH} missing: building it (hipcc --offload-arch=gfx950) ...", file=sys.stderr)
        _build.build_library(verbose=False)


from testutil import gpu_available  # noqa: E402,F401


@pytest.fixture(scope="session")
def require_gpu():
    if not gpu_available():
        pytest.fail("GPU test selected but no HIP device / libdexr.so available (no CPU fallback exists)")
