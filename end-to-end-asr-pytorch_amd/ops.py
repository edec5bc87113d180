"""torch.autograd.Function wrappers over the C ABI (PyTorch here = device memory + streams only)."""
import os

import torch

from . import _lib
from . import dist as _ldist
from ._lib import P, I, Z, F, ptr, check, cur_stream


# ----------------------------------------------------------------------------- live kernel timing (bench.py)
_TIMING = None          # list of (name, start_event, end_event, work, unit) while enabled
PEAK = {'mfma_bf16': 2500.0, 'mfma_f32': 157.3, 'hbm': 8000.0}     # TFLOP/s, TFLOP/s, GB/s (MI355X_MICROARCH.md)


def enable_kernel_timing():
    global _TIMING
    _TIMING = []
    return _TIMING


def disable_kernel_timing():
    global _TIMING
    _TIMING = None


class _Timed:
    """Brackets a C-ABI call with HIP events on the stream the kernels are launched on (torch's current stream).
    name: the rocprofv3 kernel name when the call is ONE kernel launch (single=True), a descriptive group name otherwise;
    work: algorithmic flops / bytes of the call (SURVEY.md 8d formulas); dep_steps: dependent timesteps inside the launch
    (persistent recurrences: the figure that matters there is us per dependent step)."""

    def __init__(self, name, work, unit, single=False, dep_steps=0):
        self.args = (name, work, unit, single, dep_steps)

    def __enter__(self):
        if _TIMING is not None:
            self.e0 = torch.cuda.Event(enable_timing=True)
            self.e1 = torch.cuda.Event(enable_timing=True)
            self.e0.record()
        return self

    def __exit__(self, *exc):
        if _TIMING is not None:
            self.e1.record()
            beside = _BRANCH['stream'] is not None and torch.cuda.current_stream() == _BRANCH['stream']
            _TIMING.append((self.args[0], self.e0, self.e1, self.args[1], self.args[2], self.args[3], self.args[4], beside))
        return False


def kernel_timing_summary(records):
    """Group the timed launches.  The `roofline` object is the dominant SINGLE kernel (largest total time among the
    groups that are one kernel launch per call, named as rocprofv3 names it); every group is in `breakdown`."""
    torch.cuda.synchronize()
    groups = {}
    for name, e0, e1, work, unit, single, dep, beside in records:
        g = groups.setdefault(name, dict(ms=0.0, n=0, work=0.0, unit=unit, single=single, dep=0, beside=beside))
        g['ms'] += e0.elapsed_time(e1)
        g['n'] += 1
        g['work'] += work
        g['dep'] += dep
    rows = []
    for name, g in groups.items():
        sec = g['ms'] * 1e-3
        if g['unit'] == 'flop':
            peak = PEAK['mfma_f32'] if _prec else PEAK['mfma_bf16']
            ach = g['work'] / sec / 1e12 if sec > 0 else 0.0
            row = dict(kernel=name, bound='mfma', achieved=ach, peak=peak, unit='TFLOP/s', frac=ach / peak)
        else:
            ach = g['work'] / sec / 1e9 if sec > 0 else 0.0
            row = dict(kernel=name, bound='hbm', achieved=ach, peak=PEAK['hbm'], unit='GB/s', frac=ach / PEAK['hbm'])
        row.update(launches=g['n'], avg_launch_ms=g['ms'] / g['n'], total_ms=g['ms'], single_kernel=bool(g['single']),
                   algorithmic_work_per_launch=g['work'] / g['n'], traffic=None)
        if g['beside']:             # bracket on the CTC branch stream: its kernels share the GPU with the persistent decoder loop
            row['beside_main_stream'] = True        # (wall time of the bracket, mostly spent waiting for CUs; not on the step's path)
        if g['dep']:
            row['us_per_dependent_step'] = g['ms'] * 1e3 / g['dep']
        rows.append(row)
    rows.sort(key=lambda r: -r['total_ms'])
    singles = [r for r in rows if r['single_kernel']]
    top = dict((singles or rows)[0]) if rows else {}
    top['breakdown'] = rows
    return top


# ----------------------------------------------------------------------------- side stream for weight gradients
# Weight-gradient GEMMs are off the backward dependency chain, and the persistent LSTM kernels that follow them
# occupy only ND*ceil(H/16)*slices CUs: run the wgrads on a second HIP stream, accumulating directly into the
# flat gradient buffer (param.grad views), and join before the optimiser.
_SIDE = {'enabled': True, 'stream': None, 'stream1': None, 'dirty': False, 'dirty1': False, 'inline': False}
_BRANCH = {'stream': None, 'enabled': True}           # (scheduling forms are switched by the set_* functions below: tests, tools/)
_GRAD_READY = None      # dist.backward_with_overlap: called with an encoder layer's first gradient view once that layer's
                        # (and therefore every later parameter's) gradients have all been enqueued


def set_wgrad_overlap(flag):
    _SIDE['enabled'] = bool(flag)


def set_wgrad_inline(flag):
    """True: the weight-gradient work that normally goes to the side streams runs on the current stream instead (same
    kernels, same accumulation into the flat gradient buffer).  For shapes whose persistent LSTM kernels fill every CU
    (H = 1024: 256 workgroups) there is nothing for a side stream to overlap with, only contention."""
    _SIDE['inline'] = bool(flag)


_ONE_SIDE = False      # (tools/: all weight-gradient work on one side stream)


def set_ctc_branch(flag):
    """False: the CTC head + loss run on the main stream instead of beside the attend-and-spell loops."""
    _BRANCH['enabled'] = bool(flag)


def _side_stream(which=0):
    key = 'stream' if which == 0 or _ONE_SIDE else 'stream1'
    if _SIDE[key] is None:
        _SIDE[key] = torch.cuda.Stream()
    return _SIDE[key]


def side_streams():
    """Every side stream that has been created (dist.py makes the collectives wait for their tails)."""
    return [s for s in (_SIDE['stream'], _SIDE['stream1'], _BRANCH['stream']) if s is not None]


def main_stream():
    """A high-priority stream for the step's dependency chain (Trainer.exec, bench.py): the side / branch streams keep the
    default priority, so where one of their kernels and the chain's next kernel are both ready the chain's is dispatched
    first (c3: 15.87 -> 15.78 ms; the range here is (0, -1), i.e. there is no priority below the default to give the side
    streams instead)."""
    if _BRANCH.get('main') is None:
        _BRANCH['main'] = torch.cuda.Stream(priority=-1)
    return _BRANCH['main']


def branch_stream():
    """The stream of the CTC branch (head GEMM, CTC loss and their backward): it has no consumer before the joint loss /
    the sum of the d enc contributions, so it runs beside the attend-and-spell loops (Seq2Seq.forward, JointLossFn)."""
    if _BRANCH['stream'] is None:
        _BRANCH['stream'] = torch.cuda.Stream()
    return _BRANCH['stream']


def length_stream():
    """Stream of the per-step length inference + its small D2H (Trainer.train_step(inputs_ready=...))."""
    if _BRANCH.get('len') is None:
        _BRANCH['len'] = torch.cuda.Stream()
    return _BRANCH['len']


def copy_stream():
    """Stream of the per-step H2D copy of the batch (Trainer.exec, bench.py's H2D-inclusive loop)."""
    if _BRANCH.get('copy') is None:
        _BRANCH['copy'] = torch.cuda.Stream()
    return _BRANCH['copy']


def fork_to(stream):
    """`stream` waits for everything enqueued so far on the current stream."""
    ev = torch.cuda.Event()
    ev.record(torch.cuda.current_stream())
    stream.wait_event(ev)


def join_from(stream):
    """The current stream waits for everything enqueued so far on `stream`."""
    ev = torch.cuda.Event()
    ev.record(stream)
    torch.cuda.current_stream().wait_event(ev)


def wgrad_target(p):
    """The buffer a weight gradient may be accumulated into directly (leaf with a preallocated .grad), else None."""
    if _SIDE['enabled'] and p is not None and p.is_leaf and p.grad is not None and p.grad.is_contiguous():
        return p.grad
    return None


def on_side_stream(fn, inputs, which=0, after=None):
    """Run fn() on side stream `which` (0 | 1) after everything enqueued so far on the current stream (and after the
    event `after`, if given); `inputs` are tensors fn reads (kept alive for the side stream through record_stream).
    Returns an event recorded on the side stream behind fn()."""
    main = torch.cuda.current_stream()
    if _SIDE['inline']:                     # (set_wgrad_inline: the same work in the same order on the current stream)
        if after is not None:
            main.wait_event(after)
        fn()
        done = torch.cuda.Event()
        done.record(main)
        return done
    side = _side_stream(which)
    ev = torch.cuda.Event()
    ev.record(main)
    with torch.cuda.stream(side):
        side.wait_event(ev)
        if after is not None:
            side.wait_event(after)
        fn()
        done = torch.cuda.Event()
        done.record(side)
    for t in inputs:
        if t is not None:
            t.record_stream(side)
    _SIDE['dirty' if which == 0 or _ONE_SIDE else 'dirty1'] = True
    return done


def join_side_stream():
    """Make the current stream wait for all side-stream work (call before reading gradients)."""
    for key, skey in (('dirty', 'stream'), ('dirty1', 'stream1')):
        if _SIDE[key]:
            ev = torch.cuda.Event()
            ev.record(_SIDE[skey])
            torch.cuda.current_stream().wait_event(ev)
            _SIDE[key] = False


def _ws(nbytes, device):
    return torch.empty(max(int(nbytes), 16), dtype=torch.uint8, device=device)


def _i32(t, device):
    if not torch.is_tensor(t):
        t = torch.as_tensor(t)
    return t.to(device=device, dtype=torch.int32).contiguous()


# ----------------------------------------------------------------------------- CTC
def _ctc_fwd(logits, label, enc_len, tgt_len, blank):
    L_ = _lib.lib()
    B, T, V = logits.shape
    L = label.shape[1]
    dev = logits.device
    label, enc_len, tgt_len = _i32(label, dev), _i32(enc_len, dev), _i32(tgt_len, dev)
    nbytes = L_.las_ctc_workspace_bytes(I(B), I(T), I(V), I(L))
    ws = _ws(nbytes, dev)
    nll = torch.empty(B, dtype=torch.float32, device=dev)
    la = torch.empty(B, T, 2 * L + 1, dtype=torch.float32, device=dev)
    with _Timed('ctc_loss_fwd (row pass + alpha scan)', 4.0 * B * T * (V + 2 * L + 1), 'byte'):
        check(L_.las_ctc_loss_fwd(ptr(logits), ptr(label), ptr(enc_len), ptr(tgt_len), I(B), I(T), I(V), I(L),
                                  I(blank), ptr(nll), ptr(la), ptr(ws), Z(ws.numel()), cur_stream()),
              'las_ctc_loss_fwd')
    return nll, la, (logits, label, enc_len, tgt_len, nll, la, ws, blank)


def _ctc_bwd(saved, gscale):
    L_ = _lib.lib()
    logits, label, enc_len, tgt_len, nll, la, ws, blank = saved
    B, T, V = logits.shape
    L = label.shape[1]
    grad = torch.empty_like(logits)
    gs = gscale.contiguous().float()
    with _Timed('ctc_loss_bwd (beta scan + gradient row pass)', 4.0 * B * T * (2 * V + 2 * L + 1), 'byte'):
        check(L_.las_ctc_loss_bwd(ptr(logits), ptr(label), ptr(enc_len), ptr(tgt_len), I(B), I(T), I(V), I(L),
                                  I(blank), ptr(nll), ptr(la), ptr(gs), ptr(grad), ptr(ws), Z(ws.numel()),
                                  cur_stream()), 'las_ctc_loss_bwd')
    return grad


class CTCLossFn(torch.autograd.Function):
    """nll[b], log_alpha = CTC(log_softmax(logits[b]), label[b]); reference src/solver.py:93,160."""

    @staticmethod
    def forward(ctx, logits, label, enc_len, tgt_len, blank):
        nll, la, ctx.saved = _ctc_fwd(logits.contiguous(), label, enc_len, tgt_len, blank)
        ctx.mark_non_differentiable(la)
        ctx.set_materialize_grads(False)         # (no zero tensor of log_alpha's size for its unused gradient)
        return nll, la

    @staticmethod
    def backward(ctx, gnll, _gla):
        if gnll is None:
            ctx.saved = None
            return None, None, None, None, None
        grad = _ctc_bwd(ctx.saved, gnll)
        ctx.saved = None
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
         K=None, lda=None, ldb=None, ldc=None, batch=1, sA=0, sB=0, sC=0, A16=None, B16=None, C16=None):
    """Raw las_gemm call on 2-D (or strided-batched) row-major fp32 HIP tensors.  Returns C.
    A16 / B16: bf16 twins of A / B (same shape and strides); in bf16 mode the kernel then reads those (half the bytes, no
    conversion while staging).  C16: a bf16 tensor shaped like C that also receives the result (a twin for the next
    consumer).  In f32 mode the twins are ignored (C16 is then filled by a cast)."""
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
    use16 = _prec == 0 and USE_BF16_TWINS and (A16 is not None or B16 is not None or C16 is not None)
    with _Timed('gemm_kernel (all layouts / epilogues)', 2.0 * M * N * K * batch, 'flop'):
        rc = -2
        if use16:
            a16, b16 = A16 is not None, B16 is not None
            rc = L_.las_gemm_ex(I(_prec), I(int(transA)), I(int(transB)), I(M), I(N), I(K), F(alpha),
                                P((A16 if a16 else A).data_ptr()), I(int(a16)), LL(lda), LL(sA),
                                P((B16 if b16 else B).data_ptr()), I(int(b16)), LL(ldb), LL(sB), F(beta), P(C.data_ptr()), LL(ldc),
                                LL(sC), P(bias.data_ptr()) if bias is not None else None, I(act), I(batch),
                                P(C16.data_ptr()) if C16 is not None else None, LL(C16.stride(-2) if C16 is not None else 0),
                                cur_stream())
            if rc not in (0, -2):
                check(rc, 'las_gemm_ex')
        if rc == -2:                                    # no twins / operands not aligned for them: the fp32 sources
            check(L_.las_gemm(I(_prec), I(int(transA)), I(int(transB)), I(M), I(N), I(K), F(alpha), P(A.data_ptr()), LL(lda),
                              LL(sA), P(B.data_ptr()), LL(ldb), LL(sB), F(beta), P(C.data_ptr()), LL(ldc), LL(sC),
                              P(bias.data_ptr()) if bias is not None else None, I(act), I(batch), cur_stream()), 'las_gemm')
            if C16 is not None:
                C16.copy_(C)
    return C


USE_BF16_TWINS = True      # (tests / tools switch the bf16 operand twins off by assigning this)
TWIN_MIN_ELEMS = 1 << 18    # below this an operand is not worth a cast pass: the GEMM converts it while staging


def twins_on():
    return _prec == 0 and USE_BF16_TWINS


def cast_bf16(x, out=None):
    """bf16 copy of a contiguous fp32 HIP tensor (las_cast_bf16)."""
    L_ = _lib.lib()
    x = x.contiguous()
    if out is None:
        out = torch.empty(x.shape, dtype=torch.bfloat16, device=x.device)
    check(L_.las_cast_bf16(ptr(x), P(out.data_ptr()), LL(x.numel()), cur_stream()), 'las_cast_bf16')
    return out


def twin(x, make=False):
    """The bf16 twin a producer attached to `x` (tensor attribute `_bf16`), or -- make=True, bf16 mode, big enough -- a
    fresh cast; None otherwise.  A twin is only trusted if it has x's shape and strides."""
    if not twins_on() or x is None:
        return None
    t = getattr(x, '_bf16', None)
    if t is not None and t.shape == x.shape and t.stride() == x.stride() and t.dtype == torch.bfloat16:
        return t
    if make and x.numel() >= TWIN_MIN_ELEMS and x.is_contiguous() and x.data_ptr() % 16 == 0:
        t = cast_bf16(x)
        try:
            x._bf16 = t                      # (a second consumer of the same tensor finds it)
        except Exception:
            pass
        return t
    return None


def with_twin(y, y16):
    if y16 is not None:
        y._bf16 = y16
    return y


def narrow_rows(x, n):
    """x[:n] that keeps the bf16 twin."""
    t = getattr(x, '_bf16', None)
    y = x[:n]
    if t is not None:
        y._bf16 = t[:n]
    return y



def colsum(X, out, beta=0.0, M=None, N=None, ld=None, out2=None):
    """out = beta*out + column sums of X; out2 (optional) receives the same (bias_ih / bias_hh share their gradient)."""
    L_ = _lib.lib()
    M = X.shape[0] if M is None else M
    N = X.shape[1] if N is None else N
    ld = X.stride(0) if ld is None else ld
    check(L_.las_colsum2(P(X.data_ptr()), LL(ld), I(M), I(N), F(beta), P(out.data_ptr()),
                         P(out2.data_ptr()) if out2 is not None else None, cur_stream()), 'las_colsum2')
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


def tanh_bwd(dy, y, want_twin=False):
    """dy * (1 - y^2); want_twin (bf16 mode): also its bf16 twin, written by the same kernel -> (out, out16 | None)."""
    L_ = _lib.lib()
    dy, y = dy.contiguous(), y.contiguous()
    out = torch.empty_like(y)
    if want_twin and twins_on() and y.numel() >= TWIN_MIN_ELEMS:
        out16 = torch.empty(y.shape, dtype=torch.bfloat16, device=y.device)
        check(L_.las_tanh_bwd_twin(ptr(dy), ptr(y), ptr(out), P(out16.data_ptr()), LL(y.numel()), cur_stream()), 'las_tanh_bwd_twin')
        return out, out16
    check(L_.las_tanh_bwd(ptr(dy), ptr(y), ptr(out), LL(y.numel()), cur_stream()), 'las_tanh_bwd')
    return (out, None) if want_twin else out


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
    """y = act(x W^T + b) on the last dim; act in {0: none, 1: tanh}.  reference asr.py:307,316 / :46,69 / :384,419.
    bf16 mode: the GEMMs read the bf16 twins of x (attached by its producer) and W (the optimiser's shadow) when there
    are any, and the forward GEMM's epilogue writes y's twin for the next consumer."""
    last_twin = None

    @staticmethod
    def forward(ctx, x, w, b, act):
        x16 = twin(x)
        x = x.contiguous()
        x2 = x.view(-1, x.shape[-1])
        x16 = x16.reshape(-1, x.shape[-1]) if x16 is not None and x16.is_contiguous() else twin(x2, make=True)
        w16 = twin(w)
        M, N = x2.shape[0], w.shape[0]
        y16 = torch.empty(M, N, dtype=torch.bfloat16, device=x.device) if twins_on() and M * N >= TWIN_MIN_ELEMS and N % 8 == 0 else None
        y = gemm(x2, w, transB=True, bias=b, act=act, A16=x16, B16=w16, C16=y16)
        ctx.save_for_backward(x2, w, y if act else None)
        ctx.act, ctx.has_b, ctx.xshape, ctx.bias, ctx.x16, ctx.w16 = act, b is not None, x.shape, b, x16, w16
        LinearFn.last_twin = y16.view(*x.shape[:-1], N) if y16 is not None else None
        return y.view(*x.shape[:-1], N)

    @staticmethod
    def backward(ctx, gy):
        x2, w, y = ctx.saved_tensors
        x16, w16 = ctx.x16, ctx.w16
        gy = gy.contiguous().view(-1, w.shape[0])
        if ctx.act:
            gy, gy16 = tanh_bwd(gy, y, want_twin=True)
        else:
            gy16 = twin(gy, make=True)
        gx = gw = gb = None
        if ctx.needs_input_grad[0]:
            gx = gemm(gy, w, A16=gy16, B16=w16).view(ctx.xshape)    # [M,N]x[N,K]
        b = ctx.bias
        tw, tb = wgrad_target(w), (wgrad_target(b) if b is not None else None)
        if ctx.needs_input_grad[1] and tw is not None and (not ctx.has_b or tb is not None):
            def side():
                gemm(gy, x2, tw, transA=True, beta=1.0, A16=gy16, B16=x16)      # accumulate into the flat gradient buffer
                if ctx.has_b:
                    colsum(gy, tb, beta=1.0)
            on_side_stream(side, [gy, x2, gy16, x16])
            return gx, None, None, None
        if ctx.needs_input_grad[1]:
            gw = gemm(gy, x2, transA=True, A16=gy16, B16=x16)       # [N,M]x[M,K]
        if ctx.has_b and ctx.needs_input_grad[2]:
            gb = colsum(gy, torch.empty(w.shape[0], dtype=torch.float32, device=w.device))
        return gx, gw, gb, None


def linear(x, w, b=None, act=0):
    y = LinearFn.apply(x, w, b, act)
    return with_twin(y, LinearFn.last_twin)


# ----------------------------------------------------------------------------- persistent (Bi)LSTM layer
def lstm_out_shape(T, H, ND, sr, concat):
    if sr == 1:
        return T, ND * H
    if concat:
        return T // sr, sr * ND * H
    return (T + sr - 1) // sr, ND * H


def _lstm_fwd(x, lens, w_ih, w_hh, b_ih, b_hh, sr, concat, status, w_ih16=None):
    L_ = _lib.lib()
    x16 = twin(x)
    x = x.contiguous()
    T, B, Iin = x.shape
    ND, H4, H = w_hh.shape
    dev = x.device
    # a narrow layer input (the bottom layer's 80 fbank dims): the recurrence kernel forms x W_ih^T itself (las_lstm_rec_fwd_fx) --
    # no projection GEMM (K = 80: 295 MB of output at the C2 shape) and no read-back of its rows
    fx = bool(L_.las_lstm_fwd_fx_ok(I(_prec), I(T), I(B), I(H), I(ND), I(Iin)))
    xproj = None
    if not fx:
        x16 = x16.reshape(T * B, Iin) if x16 is not None and x16.is_contiguous() else twin(x.view(T * B, Iin), make=True)
        xproj = gemm(x.view(T * B, Iin), w_ih, transB=True, A16=x16, B16=w_ih16 if twins_on() else None)
    else:
        x16 = x16.reshape(T * B, Iin) if x16 is not None and x16.is_contiguous() else None
    T_out, F_out = lstm_out_shape(T, H, ND, sr, concat)
    hf = torch.empty(T, B, ND * H, dtype=torch.float32, device=dev)
    y = hf if sr == 1 else torch.empty(T_out, B, F_out, dtype=torch.float32, device=dev)
    esz, vec = (2, 8) if _prec == 0 else (4, 4)
    Hx = (H + vec - 1) // vec * vec
    # sync words and the exchange buffer in ONE allocation, the buffer right behind the words: the library then zeroes both with one fill
    nsync, nhx = L_.las_lstm_sync_bytes(), L_.las_lstm_hx_bytes(I(_prec), I(T), I(B), I(H), I(ND))
    sx = (torch.empty if Hx == H else torch.zeros)(nsync + nhx, dtype=torch.uint8, device=dev)
    sync, hx = sx[:nsync], sx[nsync:]
    gates = torch.empty(T, B, ND * H4, dtype=torch.float32, device=dev)
    cs = torch.empty(T, B, ND * H, dtype=torch.float32, device=dev)
    # y's bf16 twin (operand of the projection GEMM behind the layer): written by the recurrence kernel itself, no cast pass
    y16 = torch.empty(y.shape, dtype=torch.bfloat16, device=dev) if twins_on() and y.numel() >= TWIN_MIN_ELEMS else None
    kname = ('lstm_fwd_kernel', 'lstm_fwd_gr_kernel', 'lstm_fwd_x32_kernel')[L_.las_lstm_fwd_variant(I(_prec), I(T), I(B), I(H), I(ND))]
    with _Timed(kname, 2.0 * ND * T * B * H4 * H, 'flop', single=True, dep_steps=T):
        if fx:
            check(L_.las_lstm_rec_fwd_fx(I(_prec), ptr(x), I(Iin), ptr(w_ih.contiguous()), ptr(b_ih), ptr(b_hh), ptr(w_hh), ptr(lens), I(T), I(B),
                                         I(H), I(ND), I(sr), I(int(concat)), ptr(y), ptr(hf), P(y16.data_ptr()) if y16 is not None else None,
                                         ptr(hx), ptr(gates), ptr(cs), ptr(sync), ptr(status), cur_stream()), 'las_lstm_rec_fwd_fx')
        else:
            check(L_.las_lstm_rec_fwd(I(_prec), ptr(xproj), ptr(b_ih), ptr(b_hh), ptr(w_hh), ptr(lens), I(T), I(B), I(H),
                                      I(ND), I(sr), I(int(concat)), ptr(y), ptr(hf), P(y16.data_ptr()) if y16 is not None else None,
                                      ptr(hx), ptr(gates), ptr(cs), ptr(sync), ptr(status), cur_stream()), 'las_lstm_rec_fwd')
    with_twin(y, y16)
    if sr == 1 and y16 is not None:
        hf._bf16_2d = y16.view(T * B, ND * H)      # (y IS hf: the weight-gradient GEMMs of the backward pass read the same copy)
    return y, (x, lens, w_ih, w_hh, hf, gates, cs, status, sr, int(concat), _prec, x16, w_ih16 if twins_on() else None)


def _lstm_bwd(saved, gy, need_gx, targets=None):
    """-> gx [T,B,I]|None, gw_ih [ND*4H,I], gw_hh [ND,4H,H], gb [ND*4H] (= d b_ih = d b_hh).
    targets = (gw_ih, gw_hh, gb_ih, gb_hh) gradient buffers to accumulate into on the side stream (returns None
    for the weight gradients then)."""
    L_ = _lib.lib()
    x, lens, w_ih, w_hh, hf, gates, cs, status, sr, concat, prec, x16, w_ih16 = saved
    T, B, Iin = x.shape
    ND, H4, H = w_hh.shape
    dev = x.device
    gy = gy.contiguous()
    nsync = L_.las_lstm_sync_bytes()
    sx = torch.empty(nsync + L_.las_lstm_bwd_ws_bytes(I(prec), I(T), I(B), I(H), I(ND)), dtype=torch.uint8, device=dev)
    sync, dgx = sx[:nsync], sx[nsync:]              # (one allocation, the workspace behind the sync words: one zero fill)
    dgf = torch.empty(T * B, ND * H4, dtype=torch.float32, device=dev)
    # d gates' bf16 twin (operand of the d x / d W_ih / d W_hh GEMMs): written by the BPTT kernel itself, no cast pass
    dgf16 = torch.empty(dgf.shape, dtype=torch.bfloat16, device=dev) if prec == 0 and USE_BF16_TWINS and dgf.numel() >= TWIN_MIN_ELEMS else None
    ksplit = L_.las_lstm_bwd_is_ksplit(I(prec), I(T), I(B), I(H), I(ND))
    if _ldist._ACTIVE['ex'] is not None:        # gradient buckets in flight on RCCL's stream: do they and this launch both fit?
        _ldist.persistent_launch_guard(L_.las_lstm_resident_wgs(I(prec), I(T), I(B), I(H), I(ND)), dev)
    with _Timed(('lstm_bwd_kernel', 'lstm_bwd_ks_kernel', 'lstm_bwd_gr_kernel', 'lstm_bwd_x32_kernel')[ksplit], 2.0 * ND * T * B * H4 * H, 'flop', single=True, dep_steps=T):
        check(L_.las_lstm_rec_bwd(I(prec), ptr(gy), ptr(gates), ptr(cs), ptr(w_hh), ptr(lens), I(T), I(B), I(H), I(ND),
                                  I(sr), I(concat), ptr(dgx), ptr(dgf), P(dgf16.data_ptr()) if dgf16 is not None else None,
                                  ptr(sync), ptr(status), cur_stream()),
              'las_lstm_rec_bwd')
    with_twin(dgf, dgf16)
    x2 = x.view(T * B, Iin)
    hf2 = hf.view(T * B, ND * H)

    def w_hh_grad(d, gw_hh, beta, d16, h16):
        if T > 1:
            if d == 0:      # sum_{t>=1} dg[t]^T h[t-1]
                sa, sb = (slice(B, None), slice(d * H4, (d + 1) * H4)), (slice(0, (T - 1) * B), slice(d * H, (d + 1) * H))
            else:           # sum_{t<=T-2} dg[t]^T h[t+1]
                sa, sb = (slice(0, (T - 1) * B), slice(d * H4, (d + 1) * H4)), (slice(B, None), slice(d * H, (d + 1) * H))
            two = d16 is not None and h16 is not None
            gemm(dgf[sa], hf2[sb], gw_hh[d], transA=True, beta=beta, A16=d16[sa] if two else None, B16=h16[sb] if two else None)
        elif beta == 0.0:
            gw_hh[d].zero_()

    if targets is not None:
        # The weight gradients are accumulated into the flat gradient buffer on TWO side streams (the main stream goes on
        # with d x and the layer below).  bf16 twins of the GEMM operands: one cast pass each (6 bytes per element,
        # HBM-bound), after which the three GEMMs that read d gates move half the bytes and convert nothing while staging;
        # only the twin the d x GEMM needs is made on the main stream.
        #   side 1: hf's twin | bias sums (ONE pass for bias_ih and bias_hh) | dW_hh of the reverse direction
        #   side 0: d gates' twin (bottom layer: nobody else needs it) | dW_ih | dW_hh of the forward direction
        gw_ih, gw_hh, gb_ih, gb_hh = targets
        dgf16 = twin(dgf, make=True) if need_gx else None
        gx = gemm(dgf, w_ih, A16=dgf16, B16=w_ih16).view(T, B, Iin) if need_gx else None
        box = {}

        def part1():
            box['hf16'] = getattr(hf, '_bf16_2d', None) if twins_on() else None
            if box['hf16'] is None:
                box['hf16'] = twin(hf2, make=True)
            box['e_hf'] = torch.cuda.Event()
            box['e_hf'].record(torch.cuda.current_stream())
            colsum(dgf, gb_ih, beta=1.0, out2=gb_hh)

        def part0a():
            box['d16'] = dgf16 if dgf16 is not None else twin(dgf, make=True)
            box['e_d16'] = torch.cuda.Event()
            box['e_d16'].record(torch.cuda.current_stream())

        def part0b():
            gemm(dgf, x2, gw_ih, transA=True, beta=1.0, A16=box['d16'], B16=x16)        # [ND*4H, I]
            torch.cuda.current_stream().wait_event(box['e_hf'])
            w_hh_grad(0, gw_hh, 1.0, box['d16'], box['hf16'])

        def part1b():
            torch.cuda.current_stream().wait_event(box['e_d16'])
            for d in range(1, ND):
                w_hh_grad(d, gw_hh, 1.0, box['d16'], box['hf16'])

        on_side_stream(part1, [dgf, hf], which=1)
        if need_gx:
            on_side_stream(lambda: (part0a(), part0b()), [dgf, x, hf, dgf16, x16, box['hf16']], which=0)
            if ND > 1:
                on_side_stream(part1b, [dgf, hf, box['d16'], box['hf16']], which=1)
        else:
            # bottom layer: nothing is left for the main stream, and the side streams share ONE hardware queue (their
            # kernels run one after the other): its half of the weight gradients runs on the main stream itself, so
            # the step's tail is two queues wide
            part0a()
            if ND > 1:
                on_side_stream(part1b, [dgf, hf, box['d16'], box['hf16']], which=1)
            part0b()
            if box['hf16'] is not None:
                box['hf16'].record_stream(torch.cuda.current_stream())
        return gx, None, None, None
    dgf16 = twin(dgf, make=True)
    hf16 = twin(hf2, make=True) if dgf16 is not None else None
    gx = gemm(dgf, w_ih, A16=dgf16, B16=w_ih16).view(T, B, Iin) if need_gx else None
    gw_ih = torch.empty(ND * H4, Iin, dtype=torch.float32, device=dev)
    gw_hh = torch.empty_like(w_hh)
    gb = torch.empty(ND * H4, dtype=torch.float32, device=dev)
    gemm(dgf, x2, gw_ih, transA=True, beta=0.0, A16=dgf16, B16=x16)        # [ND*4H, I]
    colsum(dgf, gb, beta=0.0)
    for d in range(ND):
        w_hh_grad(d, gw_hh, 0.0, dgf16, hf16)
    return gx, gw_ih, gw_hh, gb


class LstmLayerFn(torch.autograd.Function):
    """Time-major packed (Bi)LSTM layer with fused down-sampling; reference asr.py:476-501.
    x [T,B,I], lens int32 [B], w_ih [ND*4H,I], w_hh [ND,4H,H], b_ih/b_hh [ND*4H] -> y [T_out,B,F_out]."""

    @staticmethod
    def forward(ctx, x, lens, w_ih, w_hh, b_ih, b_hh, sr, concat, status):
        y, ctx.saved = _lstm_fwd(x, lens, w_ih, w_hh, b_ih, b_hh, sr, concat, status)
        return y

    @staticmethod
    def backward(ctx, gy):
        gx, gw_ih, gw_hh, gb = _lstm_bwd(ctx.saved, gy, ctx.needs_input_grad[0])
        ctx.saved = None
        return gx, None, gw_ih, gw_hh, gb, gb.clone(), None, None, None


def lstm_layer(x_tm, lens, w_ih, w_hh, b_ih, b_hh, sr, concat, status):
    return LstmLayerFn.apply(x_tm, lens, w_ih, w_hh, b_ih, b_hh, sr, concat, status)


class _LstmLeavesFn(torch.autograd.Function):
    """Same op over kernel-facing concatenated views (plain tensors), routing the gradients to the
    per-direction leaf parameters (reference names weight_ih_l0 / weight_ih_l0_reverse, ...).
    leaves order: for kind in (w_ih, w_hh, b_ih, b_hh): for direction."""

    @staticmethod
    def forward(ctx, x, lens, sr, concat, status, cats, cat_grads, *leaves):
        w_ih, w_hh, b_ih, b_hh = cats[:4]
        y, ctx.saved = _lstm_fwd(x, lens, w_ih, w_hh, b_ih, b_hh, sr, concat, status, w_ih16=cats[4] if len(cats) > 4 else None)
        ctx.shapes = [tuple(l.shape) for l in leaves]
        ctx.cat_grads = cat_grads if _SIDE['enabled'] else None
        return y

    @staticmethod
    def backward(ctx, gy):
        gx, gw_ih, gw_hh, gb = _lstm_bwd(ctx.saved, gy, ctx.needs_input_grad[0], ctx.cat_grads)
        if _GRAD_READY is not None and ctx.cat_grads is not None:
            _GRAD_READY(ctx.cat_grads[0])
        ND = ctx.saved[3].shape[0]
        n_leaves = len(ctx.shapes)
        ctx.saved = None
        if gw_ih is None:                   # accumulated into the flat gradient buffer on the side stream
            return (gx, None, None, None, None, None, None, *([None] * n_leaves))
        out = []
        k = 0
        for g in (gw_ih, gw_hh, gb, gb):
            flat = g.reshape(ND, -1)
            for d in range(ND):
                out.append(flat[d].view(ctx.shapes[k]))
                k += 1
        return (gx, None, None, None, None, None, None, *out)


def lstm_layer_leaves(x, lens, cats, sr, concat, status, ND, leaves, cat_grads=None):
    """cats = (w_ih, w_hh, b_ih, b_hh[, w_ih bf16 shadow]) kernel-facing concatenated views."""
    y = _LstmLeavesFn.apply(x, lens, sr, concat, status, cats, cat_grads, *leaves)
    return with_twin(y, twin(y, make=True))          # operand of the layer's projection GEMM


# ----------------------------------------------------------------------------- joint loss
class JointLossFn(torch.autograd.Function):
    """asr_loss = (1-w)*att_loss + w*ctc_loss, reference solver.py:144-164.
    forward(att_pred [B,L,V]|None, ctc_pred [B,T',V]|None, y i64 [B,Ly], ntok i32 [B], enc_len i32 [B], L, w)
      -> (asr_loss, att_loss, ctc_loss) 0-dim tensors (att/ctc are non-differentiable by-products)."""

    @staticmethod
    def forward(ctx, att_pred, ctc_pred, y, ntok, enc_len, L, w):
        L_ = _lib.lib()
        ctx.set_materialize_grads(False)         # (att_loss / ctc_loss are by-products: no zero fills for their gradients)
        dev = y.device
        f32 = dict(dtype=torch.float32, device=dev)
        att = ctc = datt = None
        ctx.ctc = None
        if att_pred is not None:
            att_pred = att_pred.contiguous()
            B, Lx, V = att_pred.shape
            att = torch.empty(1, **f32)
            datt = torch.empty_like(att_pred)
            rowloss = torch.empty(B * Lx, **f32)
            check(L_.las_ce_loss(ptr(att_pred), ptr(y), I(y.shape[1]), ptr(ntok), I(B), I(Lx), I(V), F(1.0 - w),
                                 ptr(rowloss), ptr(att), ptr(datt), cur_stream()), 'las_ce_loss')
        cs = getattr(ctc_pred, '_branch', None) if ctc_pred is not None else None
        pend, _BRANCH['pending'] = _BRANCH.get('pending'), None
        if cs is None and pend is not None:
            # the head GEMM was launched on the branch stream but the tensor lost its `_branch` tag on the way here (a slice,
            # a cast, a user wrapper): no overlap then, but never a race -- the main stream waits for the branch first
            join_from(pend)
        ctx.branch = cs
        if ctc_pred is not None:
            def ctc_part():
                label = y[:, 1:L + 1].contiguous()
                nll, la, ctc_ctx = _ctc_fwd(ctc_pred.contiguous(), label, enc_len, ntok, 0)
                c = torch.empty(1, **f32)
                check(L_.las_norm_mean_fwd(ptr(nll), ptr(ntok), I(nll.shape[0]), ptr(c), cur_stream()), 'las_norm_mean_fwd')
                if ctx.needs_input_grad[1]:
                    # The CTC gradient needs nothing of the attention branch: it is formed HERE, behind the loss's own forward
                    # on the same (branch) stream, for an upstream gradient of 1 -- beside the attend-and-spell loops' FORWARD
                    # instead of behind their end; backward() only scales it by the upstream gradient.  (With V = 5 000 and
                    # 60 labels the BPTT loop is shorter than the beta scan + gradient row pass, and the encoder's backward
                    # waited for the branch; at c3 the row pass no longer shares the chip with the BPTT loop: 15.65 -> 15.54 ms.)
                    B = ntok.shape[0]
                    gs = torch.empty(B, **f32)
                    one = torch.ones(1, **f32)
                    check(L_.las_norm_mean_bwd(ptr(one), F(w), ptr(ntok), I(B), ptr(gs), cur_stream()), 'las_norm_mean_bwd')
                    return c, la, _ctc_bwd(ctc_ctx, gs)
                return c, la, None
            if cs is None:
                ctc, ctx.log_alpha, ctx.ctc = ctc_part()
            else:
                # The CTC branch (Seq2Seq.forward): its stream forked from the main one after enc / the label counts were
                # enqueued and before the attend-and-spell loop, so the loss runs beside that loop; joined here.
                with torch.cuda.stream(cs):
                    ctc, ctx.log_alpha, ctx.ctc = ctc_part()
                for t_ in (y, ntok, enc_len):
                    t_.record_stream(cs)
                join_from(cs)
                ctc.record_stream(torch.cuda.current_stream())
        total = torch.empty(1, **f32)
        check(L_.las_combine2(ptr(att), F(1.0 - w), ptr(ctc), F(w), ptr(total), cur_stream()), 'las_combine2')
        ctx.datt, ctx.ntok, ctx.w = datt, ntok, w
        z = torch.zeros((), **f32)
        outs = (total.view(()), att.view(()) if att is not None else z, ctc.view(()) if ctc is not None else z)
        ctx.mark_non_differentiable(outs[1], outs[2])
        return outs

    @staticmethod
    def backward(ctx, g, _ga, _gc):
        L_ = _lib.lib()
        if g is None:
            return None, None, None, None, None, None, None
        g = g.contiguous().view(1).float()
        gatt = gctc = None
        if ctx.ctc is not None:
            def ctc_part():                 # ctx.ctc: the gradient for an upstream gradient of 1 (formed in forward)
                gctc = ctx.ctc
                check(L_.las_scale_dev(ptr(gctc), LL(gctc.numel()), ptr(g), cur_stream()), 'las_scale_dev')
                return gctc
            if ctx.branch is None:
                gctc = ctc_part()
            else:                       # beside the decoder's BPTT: its consumer (the head's backward) runs on the same stream
                fork_to(ctx.branch)
                with torch.cuda.stream(ctx.branch):
                    gctc = ctc_part()
                g.record_stream(ctx.branch)
        if ctx.datt is not None:
            gatt = ctx.datt
            check(L_.las_scale_dev(ptr(gatt), LL(gatt.numel()), ptr(g), cur_stream()), 'las_scale_dev')
        ctx.datt = ctx.ctc = None
        return gatt, gctc, None, None, None, None, None


def joint_loss(att_pred, ctc_pred, y, ntok, enc_len, L, w):
    return JointLossFn.apply(att_pred, ctc_pred, y, ntok, enc_len, L, float(w))


# ----------------------------------------------------------------------------- metrics
def argmax_rows(logits):
    """int32 argmax over the last dim of a contiguous fp32 tensor [..., V]."""
    L_ = _lib.lib()
    logits = logits.detach().contiguous()
    V = logits.shape[-1]
    rows = logits.numel() // V
    pred = torch.empty(logits.shape[:-1], dtype=torch.int32, device=logits.device)
    check(L_.las_argmax_rows(ptr(logits), I(rows), I(V), ptr(pred), cur_stream()), 'las_argmax_rows')
    return pred


def token_acc(pred, y):
    """Device-side cal_acc (reference postprocess.py:121-133): pred i32 [B,L], y i64 [B,Ly] -> 0-dim tensor."""
    L_ = _lib.lib()
    B, L = pred.shape
    out = torch.empty(1, dtype=torch.float32, device=pred.device)
    check(L_.las_token_acc(ptr(pred), ptr(y), I(y.shape[1]), I(B), I(L), ptr(out), cur_stream()), 'las_token_acc')
    return out.view(())
