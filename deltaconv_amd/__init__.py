"""deltaconv_amd -- MI355X-native DeltaConv message-passing path behind the reference's
``deltaconv.nn`` / ``deltaconv.models`` / ``deltaconv.geometry`` operator API.

    import deltaconv_amd as deltaconv      # experiments/train_*.py drop-in

All graph / operator work runs in libdeltaconv_hip.so (hand-written HIP for gfx950, C ABI in
include/deltaconv_hip.h).  There is no CPU fallback; the CPU restatement in oracle/ is test
infrastructure.
"""
import os as _os

# HIP-graph replays (graph_step.py) are only correct on ROCm 7.2 with the runtime's AQL-packet capture
# off; the flag is read when the HIP runtime initialises, so it is defaulted here, at import.
_os.environ.setdefault("DEBUG_CLR_GRAPH_PACKET_CAPTURE", "0")

from . import geometry, nn, models, transforms, datasets, optim   # noqa: F401
from .data import Batch              # noqa: F401

__version__ = (0, 1, 0)
