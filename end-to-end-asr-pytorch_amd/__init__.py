"""MI355X-native LAS training hot path (Listener / Attention / Speller / joint CTC+CE step).

Host side mirrors the reference's interface (src/asr.py Seq2Seq, src/solver.py Trainer); all
arithmetic runs in hand-written gfx950 kernels behind the C ABI of include/las_hip.h.
"""
from . import _lib
from ._lib import LasError, build

__all__ = ['_lib', 'LasError', 'build']

# Multi-process GPU work on this driver needs dmabuf IPC (RCCL / sharing device tensors across processes fail with
# `hipIpcGetMemHandle: invalid argument` otherwise); harmless for one process.  Must be in the environment before the HIP
# runtime initialises, hence at package import.
import os as _os
_os.environ.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')
