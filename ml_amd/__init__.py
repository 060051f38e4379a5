"""ml_amd -- MI355X (gfx950) native Gaussian-mixture EM / K-means hot path behind the ML++ / cppyml API.

The compute path is hand-written HIP behind the C ABI in include/mlhip.h (ml_amd/libmlhip.so).
There is no CPU fallback: importing works anywhere the shared library loads, but every compute call
needs an AMD GPU and fails loudly otherwise.
"""
from . import _lib  # noqa: F401  (loads libmlhip.so or raises)

__all__ = ["_lib"]
