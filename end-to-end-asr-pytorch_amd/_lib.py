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
    return _lib


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
