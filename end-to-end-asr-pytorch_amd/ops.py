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
