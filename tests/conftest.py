import os
import sys

import pytest

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if REPO not in sys.path:
    sys.path.insert(0, REPO)
sys.path.insert(0, os.path.join(REPO, "tests"))


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: test needs an MI355X (run with -m gpu on the GPU box)")


def gpu_available() -> bool:
    try:
        from dex_retargeting_amd import _lib

        return _lib.load().dexr_device_count() > 0
    except Exception:
        return False


@pytest.fixture(scope="session")
def require_gpu():
    if not gpu_available():
        pytest.fail("GPU test selected but no HIP device / libdexr.so available (no CPU fallback exists)")
