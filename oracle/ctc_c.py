"""ctypes wrapper over oracle/ctc_ref.c  --  TEST INFRASTRUCTURE."""
import ctypes
import os
import subprocess
import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None


def build():
    so = os.path.join(_HERE, 'libctc_ref.so')
    src = os.path.join(_HERE, 'ctc_ref.c')
    if not os.path.exists(so) or os.path.getmtime(so) < os.path.getmtime(src):
        subprocess.check_call(['make', '-s', '-C', _HERE])
    return so


def _lib():
    global _LIB
    if _LIB is None:
        _LIB = ctypes.CDLL(build())
        _LIB.ctc_ref.restype = ctypes.c_int
    return _LIB


def ctc_ref(logits, label, enc_len, tgt_len, blank=0, want_grad=True):
    logits = np.ascontiguousarray(logits, dtype=np.float32)
    label = np.ascontiguousarray(label, dtype=np.int32)
    enc_len = np.ascontiguousarray(enc_len, dtype=np.int32)
    tgt_len = np.ascontiguousarray(tgt_len, dtype=np.int32)
    B, T, V = logits.shape
    L = label.shape[1]
    nll = np.empty(B, np.float32)
    la = np.empty((B, T, 2 * L + 1), np.float32)
    grad = np.empty((B, T, V), np.float32) if want_grad else None
    p = lambda a: a.ctypes.data_as(ctypes.c_void_p) if a is not None else None
    rc = _lib().ctc_ref(p(logits), p(label), p(enc_len), p(tgt_len), B, T, V, L, blank, p(nll), p(la), p(grad))
    assert rc == 0
    return nll, la, grad
