"""torch.autograd.Function wrappers over the C ABI (PyTorch here = device memory + streams only)."""
import torch

from . import _lib
from ._lib import P, I, Z, F, ptr, check, cur_stream


def _ws(nbytes, device):
    return torch.empty(max(int(nbytes), 16), dtype=torch.uint8, device=device)


def _i32(t, device):
    if not torch.is_tensor(t):
        t = torch.as_tensor(t)
    return t.to(device=device, dtype=torch.int32).contiguous()


# ----------------------------------------------------------------------------- CTC
class CTCLossFn(torch.autograd.Function):
    """nll[b], log_alpha = CTC(log_softmax(logits[b]), label[b]); reference src/solver.py:93,160."""

    @staticmethod
    def forward(ctx, logits, label, enc_len, tgt_len, blank):
        L_ = _lib.lib()
        logits = logits.contiguous()
        B, T, V = logits.shape
        L = label.shape[1]
        dev = logits.device
        label, enc_len, tgt_len = _i32(label, dev), _i32(enc_len, dev), _i32(tgt_len, dev)
        nbytes = L_.las_ctc_workspace_bytes(I(B), I(T), I(V), I(L))
        ws = _ws(nbytes, dev)
        nll = torch.empty(B, dtype=torch.float32, device=dev)
        la = torch.empty(B, T, 2 * L + 1, dtype=torch.float32, device=dev)
        check(L_.las_ctc_loss_fwd(ptr(logits), ptr(label), ptr(enc_len), ptr(tgt_len), I(B), I(T), I(V), I(L),
                                  I(blank), ptr(nll), ptr(la), ptr(ws), Z(ws.numel()), cur_stream()),
              'las_ctc_loss_fwd')
        ctx.save_for_backward(logits, label, enc_len, tgt_len, nll, la, ws)
        ctx.blank = blank
        ctx.mark_non_differentiable(la)
        return nll, la

    @staticmethod
    def backward(ctx, gnll, _gla):
        L_ = _lib.lib()
        logits, label, enc_len, tgt_len, nll, la, ws = ctx.saved_tensors
        B, T, V = logits.shape
        L = label.shape[1]
        grad = torch.empty_like(logits)
        gs = gnll.contiguous().float()
        check(L_.las_ctc_loss_bwd(ptr(logits), ptr(label), ptr(enc_len), ptr(tgt_len), I(B), I(T), I(V), I(L),
                                  I(ctx.blank), ptr(nll), ptr(la), ptr(gs), ptr(grad), ptr(ws), Z(ws.numel()),
                                  cur_stream()), 'las_ctc_loss_bwd')
        return grad, None, None, None, None


def ctc_nll(logits, label, enc_len, tgt_len, blank=0):
    """Per-utterance CTC negative log-likelihood and the alpha lattice."""
    return CTCLossFn.apply(logits, label, enc_len, tgt_len, blank)


# ----------------------------------------------------------------------------- precision switch
_PREC = {'bf16': 0, 'f32': 1}
_prec = 0


def set_precision(name):
    """MFMA operand format of all GEMM-shaped work: 'bf16' (default) or 'f32' (exact f32 MFMA)."""
    global _prec
    _prec = _PREC[name]


def get_precision():
    return 'f32' if _prec else 'bf16'


LL = __import__('ctypes').c_longlong


# ----------------------------------------------------------------------------- GEMM
def gemm(A, B, C=None, transA=False, transB=False, alpha=1.0, beta=0.0, bias=None, act=0, M=None, N=None,
         K=None, lda=None, ldb=None, ldc=None, batch=1, sA=0, sB=0, sC=0):
    """Raw las_gemm call on 2-D (or strided-batched) row-major fp32 HIP tensors.  Returns C."""
    L_ = _lib.lib()
    if M is None:
        M = A.shape[-1] if transA else A.shape[-2]
    if K is None:
        K = A.shape[-2] if transA else A.shape[-1]
    if N is None:
        N = B.shape[-2] if transB else B.shape[-1]
    lda = lda if lda is not None else A.stride(-2)
    ldb = ldb if ldb is not None else B.stride(-2)
    if C is None:
        assert beta == 0.0
        C = torch.empty((M, N) if batch == 1 else (batch, M, N), dtype=torch.float32, device=A.device)
        if batch > 1:
            sC = M * N
    ldc = ldc if ldc is not None else C.stride(-2)
    check(L_.las_gemm(I(_prec), I(int(transA)), I(int(transB)), I(M), I(N), I(K), F(alpha), P(A.data_ptr()), LL(lda),
                      LL(sA), P(B.data_ptr()), LL(ldb), LL(sB), F(beta), P(C.data_ptr()), LL(ldc), LL(sC),
                      P(bias.data_ptr()) if bias is not None else None, I(act), I(batch), cur_stream()), 'las_gemm')
    return C


def colsum(X, out, beta=0.0, M=None, N=None, ld=None):
    L_ = _lib.lib()
    M = X.shape[0] if M is None else M
    N = X.shape[1] if N is None else N
    ld = X.stride(0) if ld is None else ld
    check(L_.las_colsum(P(X.data_ptr()), LL(ld), I(M), I(N), F(beta), P(out.data_ptr()), cur_stream()), 'las_colsum')
    return out


# ----------------------------------------------------------------------------- misc kernels
def transpose01(x):
    """[D0,D1,F] -> [D1,D0,F] (batch-major <-> time-major)."""
    L_ = _lib.lib()
    x = x.contiguous()
    D0, D1, Fd = x.shape
    out = torch.empty(D1, D0, Fd, dtype=torch.float32, device=x.device)
    check(L_.las_transpose01(ptr(x), ptr(out), I(D0), I(D1), I(Fd), cur_stream()), 'las_transpose01')
    return out


class Transpose01Fn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x):
        return transpose01(x)

    @staticmethod
    def backward(ctx, g):
        return transpose01(g)


def tanh_bwd(dy, y):
    L_ = _lib.lib()
    dy, y = dy.contiguous(), y.contiguous()
    out = torch.empty_like(y)
    check(L_.las_tanh_bwd(ptr(dy), ptr(y), ptr(out), LL(y.numel()), cur_stream()), 'las_tanh_bwd')
    return out


def infer_lengths(x):
    """int32 [B]: frames whose feature sum != 0 (reference solver.py:134), computed on the device."""
    L_ = _lib.lib()
    x = x.contiguous()
    B, T, D = x.shape
    lens = torch.empty(B, dtype=torch.int32, device=x.device)
    check(L_.las_infer_lengths(ptr(x), I(B), I(T), I(D), ptr(lens), cur_stream()), 'las_infer_lengths')
    return lens


def count_nonzero(y):
    L_ = _lib.lib()
    y = y.contiguous()
    assert y.dtype == torch.int64
    B, Lx = y.shape
    out = torch.empty(B, dtype=torch.int32, device=y.device)
    check(L_.las_count_nonzero_i64(ptr(y), I(B), I(Lx), ptr(out), cur_stream()), 'las_count_nonzero_i64')
    return out


# ----------------------------------------------------------------------------- Linear (+tanh)
class LinearFn(torch.autograd.Function):
    """y = act(x W^T + b) on the last dim; act in {0: none, 1: tanh}.  reference asr.py:307,316 / :46,69 / :384,419."""

    @staticmethod
    def forward(ctx, x, w, b, act):
        x = x.contiguous()
        x2 = x.view(-1, x.shape[-1])
        y = gemm(x2, w, transB=True, bias=b, act=act)
        ctx.save_for_backward(x2, w, y if act else None)
        ctx.act, ctx.has_b, ctx.xshape = act, b is not None, x.shape
        return y.view(*x.shape[:-1], w.shape[0])

    @staticmethod
    def backward(ctx, gy):
        x2, w, y = ctx.saved_tensors
        gy = gy.contiguous().view(-1, w.shape[0])
        if ctx.act:
            gy = tanh_bwd(gy, y)
        gx = gw = gb = None
        if ctx.needs_input_grad[0]:
            gx = gemm(gy, w).view(ctx.xshape)                       # [M,N]x[N,K]
        if ctx.needs_input_grad[1]:
            gw = gemm(gy, x2, transA=True)                          # [N,M]x[M,K]
        if ctx.has_b and ctx.needs_input_grad[2]:
            gb = colsum(gy, torch.empty(w.shape[0], dtype=torch.float32, device=w.device))
        return gx, gw, gb, None


def linear(x, w, b=None, act=0):
    return LinearFn.apply(x, w, b, act)


# ----------------------------------------------------------------------------- persistent (Bi)LSTM layer
def lstm_out_shape(T, H, ND, sr, concat):
    if sr == 1:
        return T, ND * H
    if concat:
        return T // sr, sr * ND * H
    return (T + sr - 1) // sr, ND * H


class LstmLayerFn(torch.autograd.Function):
    """Time-major packed (Bi)LSTM layer with fused down-sampling; reference asr.py:476-501.
    x [T,B,I], lens int32 [B], w_ih [ND*4H,I], w_hh [ND,4H,H], b_ih/b_hh [ND*4H] -> y [T_out,B,F_out]."""

    @staticmethod
    def forward(ctx, x, lens, w_ih, w_hh, b_ih, b_hh, sr, concat, status):
        L_ = _lib.lib()
        x = x.contiguous()
        T, B, Iin = x.shape
        ND, H4, H = w_hh.shape
        dev = x.device
        xproj = gemm(x.view(T * B, Iin), w_ih, transB=True)
        T_out, F_out = lstm_out_shape(T, H, ND, sr, concat)
        hf = torch.empty(T, B, ND * H, dtype=torch.float32, device=dev)
        y = hf if sr == 1 else torch.empty(T_out, B, F_out, dtype=torch.float32, device=dev)
        esz = 2 if _prec == 0 else 4
        hx = torch.empty(ND * T * B * H * esz, dtype=torch.uint8, device=dev)
        gates = torch.empty(T, B, ND * H4, dtype=torch.float32, device=dev)
        cs = torch.empty(T, B, ND * H, dtype=torch.float32, device=dev)
        sync = torch.empty(L_.las_lstm_sync_bytes(), dtype=torch.uint8, device=dev)
        check(L_.las_lstm_rec_fwd(I(_prec), ptr(xproj), ptr(b_ih), ptr(b_hh), ptr(w_hh), ptr(lens), I(T), I(B), I(H),
                                  I(ND), I(sr), I(int(concat)), ptr(y), ptr(hf), ptr(hx), ptr(gates), ptr(cs),
                                  ptr(sync), ptr(status), cur_stream()), 'las_lstm_rec_fwd')
        ctx.save_for_backward(x, lens, w_ih, w_hh, hf, gates, cs, status)
        ctx.cfg = (sr, int(concat), _prec)
        return y

    @staticmethod
    def backward(ctx, gy):
        L_ = _lib.lib()
        x, lens, w_ih, w_hh, hf, gates, cs, status = ctx.saved_tensors
        sr, concat, prec = ctx.cfg
        T, B, Iin = x.shape
        ND, H4, H = w_hh.shape
        dev = x.device
        gy = gy.contiguous()
        esz = 2 if prec == 0 else 4
        dgx = torch.empty(ND * T * B * H4 * esz, dtype=torch.uint8, device=dev)
        dgf = torch.empty(T * B, ND * H4, dtype=torch.float32, device=dev)
        sync = torch.empty(L_.las_lstm_sync_bytes(), dtype=torch.uint8, device=dev)
        check(L_.las_lstm_rec_bwd(I(prec), ptr(gy), ptr(gates), ptr(cs), ptr(w_hh), ptr(lens), I(T), I(B), I(H), I(ND),
                                  I(sr), I(concat), ptr(dgx), ptr(dgf), ptr(sync), ptr(status), cur_stream()),
              'las_lstm_rec_bwd')
        x2 = x.view(T * B, Iin)
        gx = gemm(dgf, w_ih).view(T, B, Iin) if ctx.needs_input_grad[0] else None
        gw_ih = gemm(dgf, x2, transA=True)                                         # [ND*4H, I]
        gb = colsum(dgf, torch.empty(ND * H4, dtype=torch.float32, device=dev))
        gw_hh = torch.empty_like(w_hh)
        hf2 = hf.view(T * B, ND * H)
        for d in range(ND):
            if T > 1:
                if d == 0:      # sum_{t>=1} dg[t]^T h[t-1]
                    A, Bm = dgf[B:, d * H4:(d + 1) * H4], hf2[:(T - 1) * B, d * H:(d + 1) * H]
                else:           # sum_{t<=T-2} dg[t]^T h[t+1]
                    A, Bm = dgf[:(T - 1) * B, d * H4:(d + 1) * H4], hf2[B:, d * H:(d + 1) * H]
                gemm(A, Bm, gw_hh[d], transA=True)
            else:
                gw_hh[d].zero_()
        return gx, None, gw_ih, gw_hh, gb, gb.clone(), None, None, None


def lstm_layer(x_tm, lens, w_ih, w_hh, b_ih, b_hh, sr, concat, status):
    return LstmLayerFn.apply(x_tm, lens, w_ih, w_hh, b_ih, b_hh, sr, concat, status)
