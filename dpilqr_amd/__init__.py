"""dpilqr_amd -- MI355X-native batched iLQR for DP-iLQR (hot path of labicon/dp-ilqr).

Host-side mirror of the reference's plugin/solver interface over hand-written HIP kernels
(libdpilqr_hip.so, C ABI in include/dpilqr_hip.h).  There is no CPU fallback.
"""
from . import _lib  # noqa: F401
from .batch import ProblemBatch, backward_pass_tiles, pack_tiles  # noqa: F401
