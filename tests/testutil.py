"""Helpers shared by the test modules (kept out of conftest.py: pytest imports every directory's conftest under the
module name `conftest`, so `from conftest import ...` depends on which one was loaded last)."""
import os

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def gpu_available() -> bool:
    try:
        from dex_retargeting_amd import _lib

        return _lib.load().dexr_device_count() > 0
    except Exception:
        return False
