"""MI355X-native LAS training hot path (Listener / Attention / Speller / joint CTC+CE step).

Host side mirrors the reference's interface (src/asr.py Seq2Seq, src/solver.py Trainer); all
arithmetic runs in hand-written gfx950 kernels behind the C ABI of include/las_hip.h.
"""
from . import _lib
from ._lib import LasError, build

__all__ = ['_lib', 'LasError', 'build']
