"""dex_retargeting_amd -- MI355X-native batched retargeting solver behind dex_retargeting's
SeqRetargeting / RetargetingConfig API.  The per-frame solve runs in hand-written HIP kernels (libdexr.so, gfx950)
called through a ctypes C-ABI; there is no CPU fallback."""
from .constants import DEFAULT_URDF_DIR  # noqa: F401

__version__ = "0.1.0"
