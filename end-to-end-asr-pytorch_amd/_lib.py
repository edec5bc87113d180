"""ctypes binding of liblas_hip.so (the C ABI declared in include/las_hip.h).

The product path has no fallback: if the shared library is missing or a symbol is absent this module
raises, and every op raises if its tensors are not on a HIP device.
"""
import ctypes
import os
import re
import subprocess

_HERE = os.path.dirname(os.path.abspath(__file__))
_CSRC = os.path.join(_HERE, 'csrc')
_SO = os.path.join(_HERE, os.environ.get('LAS_HIP_LIB', 'liblas_hip.so'))      # (LAS_HIP_LIB: diagnostic builds, tools/ only)
_HEADER = os.path.join(os.path.dirname(_HERE), 'include', 'las_hip.h')
_lib = None

P, I, Z, F = ctypes.c_void_p, ctypes.c_int, ctypes.c_size_t, ctypes.c_float

ERR = {-1: 'LAS_E_BADARG', -2: 'LAS_E_UNSUPPORTED', -3: 'LAS_E_WORKSPACE', -4: 'LAS_E_TIMEOUT'}


class LasError(RuntimeError):
    pass


def build(force=False, jobs=8):
    """Compile every HIP source for gfx950 into liblas_hip.so (cross-compiles without a GPU)."""
    if force:
        subprocess.check_call(['make', '-s', '-C', _CSRC, 'clean'])
    subprocess.check_call(['make', '-s', f'-j{jobs}', '-C', _CSRC])
    return _SO


def declared_symbols():
    """Every function name include/las_hip.h declares."""
    txt = open(_HEADER).read()
    txt = re.sub(r'/\*.*?\*/', '', txt, flags=re.S)
    return sorted(set(re.findall(r'\b(las_\w+)\s*\(', txt)))


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(_SO):
            raise LasError(f'{_SO} is missing: run __graft_entry__.build() (hipcc --offload-arch=gfx950); '
                           'there is no fallback path')
        # torch must be imported first: it loads its bundled HIP runtime (same soname, libamdhip64.so.7); loading
        # liblas_hip.so first would pull /opt/rocm's copy into the process and the two runtimes do not share devices
        # or streams.
        import torch  # noqa: F401
        L = ctypes.CDLL(_SO)
        for name in declared_symbols():
            if not hasattr(L, name):
                raise LasError(f'liblas_hip.so does not export {name} declared in include/las_hip.h')
            fn = getattr(L, name)
            fn.restype = Z if name.endswith('_bytes') else I
        L.las_error_string.restype = ctypes.c_char_p
        _lib = L
        _granule_selftest(L)
    return _lib


SELFTEST = None      # dict(torn=, finished=, timeouts=, ok=) of this process's start-up granule self-test (None: no GPU / skipped)


def _granule_selftest(L, iters=2000):
    """Once per process, on the current device: the 16-byte-granule property every persistent kernel's hand-off rests on
    (csrc/selftest.hip).  If an observation is ever inconsistent -- or the test does not finish -- the flag-protocol
    LSTM kernels and the per-step decoder kernels are selected for the rest of the process (the documented fallbacks)."""
    global SELFTEST
    import torch
    if os.environ.get('LAS_SKIP_SELFTEST') or not torch.cuda.is_available():
        return
    ws = torch.empty(int(L.las_granule_selftest_bytes()), dtype=torch.uint8, device='cuda')
    res = (ctypes.c_uint * 3)()
    rc = L.las_granule_selftest(I(iters), P(ws.data_ptr()), res, P(torch.cuda.current_stream().cuda_stream))
    expect = 2 * 64 * 64                      # lanes that must finish: 64 pairs x 2 sides x 64 lanes, both flavours summed -> x 2 below
    SELFTEST = dict(rc=int(rc), torn=int(res[0]), finished=int(res[1]), timeouts=int(res[2]))
    SELFTEST['ok'] = rc == 0 and res[0] == 0 and res[2] == 0 and res[1] == 2 * expect
    if not SELFTEST['ok']:
        import warnings
        warnings.warn(f'liblas_hip: the tagged-granule self-test failed on this device ({SELFTEST}); using the flag-protocol LSTM '
                      'kernels and the per-step decoder kernels (LAS_LSTM_NO_GR=1, LAS_DEC_NO_PK=1)')
        os.environ['LAS_LSTM_NO_GR'] = '1'
        os.environ['LAS_DEC_NO_PK'] = '1'


def check(rc, what=''):
    if rc != 0:
        msg = ERR.get(rc) or lib().las_error_string(int(rc)).decode()
        raise LasError(f'{what} failed: {rc} ({msg})')


def ptr(t):
    """Device pointer of a contiguous HIP tensor (None passes NULL)."""
    if t is None:
        return None
    if not t.is_cuda:
        raise LasError('liblas_hip ops need HIP device tensors (no CPU fallback)')
    if not t.is_contiguous():
        raise LasError('liblas_hip ops need contiguous tensors')
    return P(t.data_ptr())


def cur_stream():
    import torch
    return P(torch.cuda.current_stream().cuda_stream)
