"""ams_hip: Python binding of libams_hip.so, the gfx950 HIP implementation of the separation hot path.

`ams_hip.ops` are thin tensor wrappers over the C ABI (include/ams.h); `ams_hip.functional` wraps them as
torch.autograd.Function so the host-side mirror of the reference's model classes can compose them.
"""
from ._lib import load, AmsError, LIB_PATH, parse_header  # noqa: F401
