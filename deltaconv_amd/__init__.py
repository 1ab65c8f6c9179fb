"""deltaconv_amd -- MI355X-native DeltaConv message-passing path behind the reference's
``deltaconv.nn`` / ``deltaconv.models`` / ``deltaconv.geometry`` operator API.

    import deltaconv_amd as deltaconv      # experiments/train_*.py drop-in

All graph / operator work runs in libdeltaconv_hip.so (hand-written HIP for gfx950, C ABI in
include/deltaconv_hip.h).  There is no CPU fallback; the CPU restatement in oracle/ is test
infrastructure.
"""
from . import geometry, nn, models, transforms   # noqa: F401
from .data import Batch              # noqa: F401

__version__ = (0, 1, 0)
