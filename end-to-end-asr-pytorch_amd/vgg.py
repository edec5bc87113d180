"""Host side of the VGG front-end (reference src/asr.py:507-558): ctypes mirrors of las_vgg_* (include/las_hip.h)
and the autograd Function that runs the four convolutions + two poolings (and their backward) behind one C-ABI call
each."""
import ctypes
import torch

from . import _lib, ops
from ._lib import P, I, ptr, check, cur_stream

NAMES = [f'conv{i}.{k}' for i in range(1, 5) for k in ('weight', 'bias')]


class VggDims(ctypes.Structure):
    _fields_ = ([(n, ctypes.c_int) for n in ('C_in', 'F', 'Tt', 'T2', 'F2', 'T4', 'F4', 'out_dim')] +
                [('Kp', ctypes.c_int * 4)] +
                [(n, ctypes.c_int64) for n in ('R1', 'R2', 'R3', 'col_floats', 'wr_floats')])


class VggParams(ctypes.Structure):
    _fields_ = [('w', P * 4), ('b', P * 4)]


class VggGrads(ctypes.Structure):
    _fields_ = [('dw', P * 4), ('db', P * 4)]


class VggState(ctypes.Structure):
    _fields_ = [(n, P) for n in ('y1', 'y2', 'p1', 'y3', 'y4', 'idx1', 'idx2', 'col', 'wr', 'dwr', 'ga', 'gb')]


def check_dim(d):
    """(in_channel, freq_dim, out_dim) as VGGExtractor.check_dim, asr.py:522-531 (no device needed)."""
    if d % 13 == 0:
        return d // 13, 13, (13 // 4) * 128
    if d % 40 == 0:
        return d // 40, 40, (40 // 4) * 128
    raise ValueError('Acoustic feature dimension for VGG should be 13/26/39(MFCC) or 40/80/120(Fbank) but got ' + str(d))


def get_dims(B, T, D):
    d = VggDims()
    check(_lib.lib().las_vgg_get_dims(I(B), I(T), I(D), ctypes.byref(d)), 'las_vgg_get_dims')
    return d


def _flops(d):
    cout = (64, 64, 128, 128)
    return sum(2.0 * (d.R1 if i < 2 else d.R2) * cout[i] * d.Kp[i] for i in range(4))


class VGGFn(torch.autograd.Function):
    """out = VGGExtractor(x) with x [B,T,D] batch-major; out is [T//4, B, out_dim] (time_major) or [B, T//4, out_dim]."""

    @staticmethod
    def forward(ctx, x, time_major, w1, b1, w2, b2, w3, b3, w4, b4):
        L_ = _lib.lib()
        x = x.contiguous()
        B, T, D = x.shape
        d = get_dims(B, T, D)
        dev = x.device
        f32 = dict(dtype=torch.float32, device=dev)
        u8 = dict(dtype=torch.uint8, device=dev)
        ws = [w.contiguous() for w in (w1, w2, w3, w4)]
        bs = [b.contiguous() for b in (b1, b2, b3, b4)]
        S = dict(y1=torch.empty(d.R1, 64, **f32), y2=torch.empty(d.R1, 64, **f32), p1=torch.empty(d.R2, 64, **f32),
                 y3=torch.empty(d.R2, 128, **f32), y4=torch.empty(d.R2, 128, **f32),
                 idx1=torch.empty(d.R2, 64, **u8), idx2=torch.empty(d.R3, 128, **u8),
                 col=torch.empty(d.col_floats, **f32), wr=torch.empty(d.wr_floats, **f32))
        st = VggState()
        for k, v in S.items():
            setattr(st, k, v.data_ptr())
        pr = VggParams()
        for i in range(4):
            pr.w[i], pr.b[i] = ws[i].data_ptr(), bs[i].data_ptr()
        out = torch.empty((d.T4, B, d.out_dim) if time_major else (B, d.T4, d.out_dim), **f32)
        with ops._Timed('vgg_fwd (4 conv + 2 pool)', _flops(d), 'flop'):
            check(L_.las_vgg_fwd(I(ops._prec), ptr(x), I(B), I(T), I(D), ctypes.byref(pr), ctypes.byref(st), ptr(out),
                                 I(int(time_major)), cur_stream()), 'las_vgg_fwd')
        del S['col']                                   # scratch: not kept between forward and backward
        ctx.S, ctx.dims, ctx.time_major, ctx.x = S, d, bool(time_major), x
        ctx.params = (w1, b1, w2, b2, w3, b3, w4, b4)
        return out

    @staticmethod
    def backward(ctx, gout):
        L_ = _lib.lib()
        S, d, x = ctx.S, ctx.dims, ctx.x
        B, T, D = x.shape
        dev = x.device
        f32 = dict(dtype=torch.float32, device=dev)
        gout = gout.contiguous()
        S = dict(S, col=torch.empty(d.col_floats, **f32), dwr=torch.empty(d.wr_floats, **f32),
                 ga=torch.empty(d.R1, 64, **f32), gb=torch.empty(d.R1, 64, **f32))
        st = VggState()
        for k, v in S.items():
            setattr(st, k, v.data_ptr())
        tg = [ops.wgrad_target(p) for p in ctx.params]
        direct = all(t is not None for t in tg)
        if not direct:
            tg = [torch.zeros_like(p) for p in ctx.params]
        gr = VggGrads()
        for i in range(4):
            gr.dw[i], gr.db[i] = tg[2 * i].data_ptr(), tg[2 * i + 1].data_ptr()
        dx = torch.empty_like(x) if ctx.needs_input_grad[0] else None
        fl = 2.0 * _flops(d) - (0.0 if dx is not None else 2.0 * d.R1 * 64 * d.Kp[0])
        with ops._Timed('vgg_bwd (4 conv + 2 pool)', fl, 'flop'):
            check(L_.las_vgg_bwd(I(ops._prec), ptr(x), ptr(gout), I(B), I(T), I(D), I(int(ctx.time_major)),
                                 ctypes.byref(st), ctypes.byref(gr), ptr(dx), cur_stream()), 'las_vgg_bwd')
        ctx.S = None
        return (dx, None) + (tuple([None] * 8) if direct else tuple(tg))


def vgg_extractor(x, W, time_major=False, prefix='encoder.vgg_extractor.'):
    """W: mapping of the reference's parameter names (conv{1..4}.weight/bias under `prefix`) to HIP tensors."""
    return VGGFn.apply(x, time_major, *[W[prefix + n] for n in NAMES])
