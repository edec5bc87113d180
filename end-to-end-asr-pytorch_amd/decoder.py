"""Host side of the attend-and-spell decoder loop: ctypes mirrors of las_dec_* (include/las_hip.h) and the
autograd Function that runs all L steps (and their BPTT) behind two C-ABI calls."""
import ctypes
import os
import torch

from . import _lib, ops
from ._lib import P, I, ptr, check, cur_stream

LOC_C, LOC_W = 10, 201


class DecDims(ctypes.Structure):
    _fields_ = ([(n, ctypes.c_int) for n in ('B', 'Tp', 'E', 'A', 'C', 'NL', 'V', 'L', 'loc', 'prec')] +
                [('dropout', ctypes.c_float), ('drop_seed', ctypes.c_uint)])


class DecParams(ctypes.Structure):
    _fields_ = [('emb', P), ('w_phi', P), ('conv_w', P), ('w_lp', P), ('w_e', P), ('b_e', P),
                ('w_ih', P * 4), ('w_hh', P * 4), ('b_ih', P * 4), ('b_hh', P * 4), ('w_char', P), ('b_char', P),
                ('w_ihT', P * 4), ('w_hhT', P * 4), ('w_phiT', P),
                ('pk_phi', P), ('pk_cell', P * 4), ('pk_dx', P * 4), ('pk_dh', P * 4)]


class DecState(ctypes.Structure):
    _fields_ = [('tok', P), ('xin', P), ('q', P), ('att', P), ('hs', P), ('cs', P), ('gates', P), ('f', P), ('s', P),
                ('ebuf', P), ('logits_step', P), ('xdrop', P), ('hdrop', P), ('pk_ws', P), ('pk_status', P)]


def _p(t):
    return t.data_ptr() if t is not None else None


USE_PACKED = True     # bf16 mode: weight operands of the per-step products pre-packed in MFMA fragment order


def pack_weights(ws, N, cell_C=0):
    """ws: up to three [N, K_i] fp32 HIP matrices (the k-segments of one skinny product) -> packed bf16 operand."""
    L_ = _lib.lib()
    ws = [w.contiguous() for w in ws] + [None] * (3 - len(ws))
    K = [int(w.shape[1]) if w is not None else 0 for w in ws]
    nbytes = L_.las_skinny_pack_bytes(I(N), I(K[0]), I(K[1]), I(K[2]), I(1 if cell_C else 0), I(cell_C))
    out = torch.empty(nbytes, dtype=torch.uint8, device=ws[0].device)
    LL = ctypes.c_int64
    check(L_.las_skinny_pack_weights(ptr(ws[0]), LL(K[0]), I(K[0]), ptr(ws[1]), LL(K[1]), I(K[1]), ptr(ws[2]), LL(K[2]), I(K[2]),
                                     I(N), I(1 if cell_C else 0), I(cell_C), P(out.data_ptr()), cur_stream()), 'las_skinny_pack_weights')
    return out


def make_params(W, NL, loc, transposed=None, packed=None):
    """W: dict of contiguous fp32 HIP tensors with the reference's parameter names.  packed: dict of pack_weights()
    results under 'phi', 'cell{l}', 'dx{l}', 'dh{l}' (optional)."""
    p = DecParams()
    if packed:
        p.pk_phi = _p(packed.get('phi'))
        for l in range(NL):
            p.pk_cell[l], p.pk_dx[l], p.pk_dh[l] = _p(packed.get(f'cell{l}')), _p(packed.get(f'dx{l}')), _p(packed.get(f'dh{l}'))
    p.emb, p.w_phi = _p(W['embed.weight']), _p(W['attention.phi.weight'])
    if loc:
        p.conv_w, p.w_lp = _p(W['attention.loc_conv.weight']), _p(W['attention.loc_proj.weight'])
        p.w_e, p.b_e = _p(W['attention.gen_energy.weight']), _p(W['attention.gen_energy.bias'])
    for l in range(NL):
        p.w_ih[l], p.w_hh[l] = _p(W[f'decoder.layer{l}.weight_ih']), _p(W[f'decoder.layer{l}.weight_hh'])
        p.b_ih[l], p.b_hh[l] = _p(W[f'decoder.layer{l}.bias_ih']), _p(W[f'decoder.layer{l}.bias_hh'])
    p.w_char, p.b_char = _p(W['char_trans.weight']), _p(W['char_trans.bias'])
    if transposed is not None:
        for l in range(NL):
            p.w_ihT[l], p.w_hhT[l] = _p(transposed[f'ih{l}']), _p(transposed[f'hh{l}'])
        p.w_phiT = _p(transposed['phi'])
    return p


def s_dtype(prec):
    """Element type of the saved s = tanh(psi + q + u): fp32, or the library's 16-bit code in bf16 mode (las_decoder_s_elem_bytes)."""
    return {4: torch.float32, 2: torch.int16}[_lib.lib().las_decoder_s_elem_bytes(I(prec))]


def alloc_state(dims, dev, status=None, persistent=True):
    """Saved-state tensors of one decode loop.  persistent: also the workspace of the one-launch loop (decoder_pk.hip) when
    the shape / mode is eligible (las_decoder_pk_workspace_bytes > 0); status: int32 [1] device tensor that a hand-off
    timeout of that launch is reported in (the model's, so that the Trainer's check sees it)."""
    d = dims
    f32 = dict(dtype=torch.float32, device=dev)
    S = dict(
        tok=torch.empty(d.L, d.B, dtype=torch.int32, device=dev),
        xin=torch.empty(d.L, d.B, d.C + d.E, **f32),
        q=torch.empty(d.L, d.B, d.A, **f32),
        att=torch.empty(d.L + 1, d.B, d.Tp, **f32),
        hs=torch.empty(d.NL, d.L + 1, d.B, d.C, **f32),
        cs=torch.empty(d.NL, d.L + 1, d.B, d.C, **f32),
        gates=torch.empty(d.NL, d.L, d.B, 4 * d.C, **f32),
        ebuf=torch.empty(d.B, d.Tp, **f32),
        logits_step=torch.empty(d.B, d.V, **f32),
    )
    if d.loc:
        S['f'] = torch.empty(d.L, d.B, LOC_C, d.Tp, **f32)
        S['s'] = torch.empty(d.L, d.B, d.Tp, d.A, dtype=s_dtype(d.prec), device=dev)
    if d.dropout > 0:
        S['xdrop'] = torch.empty(d.L, d.B, d.C + d.E, **f32)
        if d.NL > 1:
            S['hdrop'] = torch.empty(d.NL, d.L, d.B, d.C, **f32)
    if persistent:
        nbytes = _lib.lib().las_decoder_pk_workspace_bytes(ctypes.byref(d))
        if nbytes:
            S['pk_ws'] = torch.empty(nbytes, dtype=torch.uint8, device=dev)
            S['pk_status'] = status if status is not None else torch.zeros(1, dtype=torch.int32, device=dev)
    st = DecState()
    for k, v in S.items():
        setattr(st, k, v.data_ptr())
    return S, st


def decoder_forward_raw(W, enc, psi, enc_len, y, L, NL, loc, step_mode=None, seed=0, dropout=0.0, drop_seed=0, status=None,
                        persistent=True):
    """Runs las_decoder_fwd; returns the dict of saved-state tensors (used by tests and by DecoderFn)."""
    L_ = _lib.lib()
    B, Tp, E = enc.shape
    A = psi.shape[-1]
    V, C = W['embed.weight'].shape
    dims = DecDims(B, Tp, E, A, C, NL, V, L, int(loc), ops._prec, float(dropout), int(drop_seed) & 0xffffffff)
    S, st = alloc_state(dims, enc.device, status, persistent and step_mode is None and y is not None)
    packed = None
    if ops._prec == 0 and USE_PACKED and 'pk_ws' not in S:      # (the persistent loop reads the plain weights: nothing to pack)
        packed = {'phi': pack_weights([W['attention.phi.weight']], A)}
        for l in range(NL):
            packed[f'cell{l}'] = pack_weights([W[f'decoder.layer{l}.weight_ih'], W[f'decoder.layer{l}.weight_hh']], 4 * C, cell_C=C)
    params = make_params(W, NL, loc, packed=packed)
    S['_packed'] = packed
    sm = None
    if step_mode is not None:
        sm = (ctypes.c_uint8 * L)(*[int(v) for v in step_mode])
    # algorithmic HBM bytes of the attention steps: psi + enc re-read per step
    with ops._Timed('decoder_fwd (L attend+spell steps)', (2.0 if ops._prec == 0 else 4.0) * L * B * Tp * (A + E), 'byte'):
        check(L_.las_decoder_fwd(ctypes.byref(dims), ctypes.byref(params), ptr(enc), ptr(psi), ptr(enc_len),
                                 ptr(y) if y is not None else None, I(y.shape[1] if y is not None else 0), sm,
                                 ctypes.c_uint(seed & 0xffffffff), ctypes.byref(st), cur_stream()), 'las_decoder_fwd')
    S['_dims'], S['_keep'] = dims, (params, st)
    return S


class DecBwdState(ctypes.Structure):
    _fields_ = [(n, P) for n in ('dgates', 'dxin', 'dq_pre', 'de', 'dh_carry', 'dc_carry', 'd_below', 'da', 'df',
                                 'dpsi', 'acc', 'demb', 'pk_ws', 'pk_status', 'enc_bf16')]


def transpose2d(w):
    L_ = _lib.lib()
    w = w.contiguous()
    R, Cc = w.shape
    out = torch.empty(Cc, R, dtype=torch.float32, device=w.device)
    check(L_.las_transpose2d(ptr(w), ptr(out), I(R), I(Cc), cur_stream()), 'las_transpose2d')
    return out


WNAMES_BASE = ['embed.weight', 'attention.phi.weight']
WNAMES_LOC = ['attention.loc_conv.weight', 'attention.loc_proj.weight', 'attention.gen_energy.weight',
              'attention.gen_energy.bias']


def weight_names(NL, loc):
    n = list(WNAMES_BASE) + (list(WNAMES_LOC) if loc else [])
    for l in range(NL):
        n += [f'decoder.layer{l}.weight_ih', f'decoder.layer{l}.weight_hh', f'decoder.layer{l}.bias_ih',
              f'decoder.layer{l}.bias_hh']
    return n + ['char_trans.weight', 'char_trans.bias']


class DecoderFn(torch.autograd.Function):
    """All L attend-and-spell steps (reference asr.py:77-107) and their BPTT.
    forward(enc [B,T',E], psi [B,T',A], enc_len i32 [B], y i64 [B,Ly] | None, L, NL, loc, step_mode, seed | (seed,
    dropout, drop_seed), *weights)
      -> h_top [L,B,C] (time-major top-layer states), att [L,B,T'] (non-differentiable)."""

    persistent_bwd = True       # (tests switch it off to compare against the per-step BPTT kernels)
    last_tok = None

    @staticmethod
    def forward(ctx, enc, psi, enc_len, y, L, NL, loc, step_mode, seed, *weights):
        status = None
        if isinstance(seed, dict):                       # (seed, status tensor[, dropout, drop_seed]) from Seq2Seq.forward
            status, seed = seed.get('status'), seed['seed']
        names = weight_names(NL, loc)
        W = dict(zip(names, weights))
        enc, psi = enc.contiguous(), psi.contiguous()
        dropout, drop_seed = 0.0, 0
        if isinstance(seed, tuple):
            seed, dropout, drop_seed = seed
        S = decoder_forward_raw(W, enc, psi, enc_len, y, L, NL, loc, step_mode, seed, dropout, drop_seed, status=status)
        ctx.S, ctx.W, ctx.cfg = S, W, (L, NL, loc, names)
        ctx.enc16 = ops.twin(enc) if 'pk_ws' not in S else None      # (the per-step attention backward reads enc as bf16 in bf16 mode)
        DecoderFn.last_tok = S['tok']                     # int32 [L][B]: the token fed at every step (tests read the sampler's draws)
        ctx.save_for_backward(enc, psi, enc_len)
        h_top = S['hs'][NL - 1, 1:]
        att = S['att'][1:]
        ctx.mark_non_differentiable(att)
        ctx.set_materialize_grads(False)         # (no [L, B, T'] zero tensor for the attention maps' unused gradient)
        return h_top, att

    @staticmethod
    def backward(ctx, g_htop, _g_att):
        L_ = _lib.lib()
        if g_htop is None:                       # (h_top unused by the loss: nothing flows back)
            return (None,) * (9 + len(ctx.cfg[3]))
        enc, psi, enc_len = ctx.saved_tensors
        S, W = ctx.S, ctx.W
        L, NL, loc, names = ctx.cfg
        d = S['_dims']
        dev = enc.device
        B, Tp, E, A, C, V = d.B, d.Tp, d.E, d.A, d.C, d.V
        f32 = dict(dtype=torch.float32, device=dev)
        # the persistent BPTT loop reads the plain weights: the transposed / packed copies are the per-step chain's
        pk_bwd_bytes = L_.las_decoder_pk_bwd_workspace_bytes(ctypes.byref(d)) if ('pk_ws' in S and DecoderFn.persistent_bwd) else 0
        tr = None
        if not pk_bwd_bytes:
            tr = {'phi': transpose2d(W['attention.phi.weight'])}
            for l in range(NL):
                tr[f'ih{l}'] = transpose2d(W[f'decoder.layer{l}.weight_ih'])
                tr[f'hh{l}'] = transpose2d(W[f'decoder.layer{l}.weight_hh'])
        packed = None
        if tr is not None and (S.get('_packed') is not None or (ops._prec == 0 and USE_PACKED)):
            packed = {}
            for l in range(NL):
                packed[f'dx{l}'] = pack_weights([tr[f'ih{l}']], int(tr[f'ih{l}'].shape[0]))
                packed[f'dh{l}'] = pack_weights([tr[f'hh{l}']] + ([tr['phi']] if l == 0 else []), C)
        params = make_params(W, NL, loc, tr, packed=packed)
        nch = L_.las_decoder_att_chunks(I(Tp))
        accf = L_.las_decoder_loc_acc_floats(I(A))
        Bw = dict(dgates=torch.empty(NL, L, B, 4 * C, **f32), dxin=torch.empty(L, B, C + E, **f32),
                  dq_pre=torch.empty(L, B, A, **f32), dh_carry=torch.empty(NL, B, C, **f32),
                  dc_carry=torch.empty(NL, B, C, **f32), d_below=torch.empty(B, C, **f32), da=torch.empty(B, Tp, **f32),
                  demb=torch.empty(V, C, **f32), de=torch.empty(L, B, Tp, **f32))
        if loc:
            Bw.update(df=torch.empty(L, B, LOC_C, Tp, **f32), dpsi=torch.empty(B, Tp, A, **f32),
                      acc=torch.empty(B, accf, **f32))
        if pk_bwd_bytes:                                 # the forward ran as one persistent launch: so does the BPTT chain
            nb = pk_bwd_bytes
            if nb:
                Bw['pk_ws'] = torch.empty(nb, dtype=torch.uint8, device=dev)
                Bw['pk_status'] = S['pk_status']
        DecoderFn.last_pk_bwd_ws = Bw.get('pk_ws')      # (tools/pk_stamps.py reads the diagnostic build's cycle stamps from it)
        bw = DecBwdState()
        for k, v in Bw.items():
            setattr(bw, k, v.data_ptr())
        if ctx.enc16 is not None and ctx.enc16.is_contiguous() and tuple(ctx.enc16.shape) == (B, Tp, E):
            bw.enc_bf16 = ctx.enc16.data_ptr()
        st = S['_keep'][1]
        g_htop = g_htop.contiguous()
        tg = {n: ops.wgrad_target(W[n]) for n in names}
        direct = all(tg[n] is not None for n in names if n not in ('embed.weight', 'char_trans.weight', 'char_trans.bias'))
        # every parameter offers its gradient buffer: the parameter-only sums over the steps (d conv_w, embedding rows, the
        # reduction of the per-utterance accumulators) leave the main stream, which goes on with d enc / d psi
        split = direct and tg['embed.weight'] is not None
        with ops._Timed('decoder_bwd (L steps BPTT)', (2.0 if ops._prec == 0 else 4.0) * L * B * Tp * (A + E), 'byte'):      # enc + saved s (loc) / psi (dot)
            check(L_.las_decoder_bwd_parts(ctypes.byref(d), ctypes.byref(params), ptr(enc), ptr(psi), ptr(enc_len),
                                           ctypes.byref(st), ptr(g_htop), ctypes.byref(bw), I(1 if split else 3), cur_stream()),
                  'las_decoder_bwd_parts')
        # ---- contractions over the L steps: one GEMM each; the weight gradients are off the dependency chain and go
        # to the side stream, accumulated into the flat gradient buffer when every parameter offers one
        LB = L * B
        XI = C + E
        g = {}
        hs0_prev = S['hs'][0, :L].reshape(LB, C)
        off = ((A * 10 + A + 1 + 3) // 4) * 4

        def loc_grads(red):
            return {'attention.loc_proj.weight': red[:A * 10].view(10, A).t(),
                    'attention.gen_energy.weight': red[A * 10:A * 10 + A].view(1, A),
                    'attention.gen_energy.bias': red[A * 10 + A:A * 10 + A + 1],
                    'attention.loc_conv.weight': red[off:off + 10 * 201].view(10, 1, 201)}

        if split:
            def param_sums():
                check(L_.las_decoder_bwd_parts(ctypes.byref(d), ctypes.byref(params), ptr(enc), ptr(psi), ptr(enc_len),
                                               ctypes.byref(st), ptr(g_htop), ctypes.byref(bw), I(2), cur_stream()),
                      'las_decoder_bwd_parts')
                tg['embed.weight'].add_(Bw['demb'])
                if loc:
                    red = ops.colsum(Bw['acc'], torch.empty(accf, **f32))
                    for n, v in loc_grads(red).items():
                        tg[n].add_(v)
            ops.on_side_stream(param_sums, [Bw['demb'], Bw['dxin'], Bw.get('df'), Bw.get('acc'), Bw['de'], S['att'], S['tok'], S.get('s'), S.get('f'), enc_len, g_htop],
                               which=1)
        else:
            g['embed.weight'] = Bw['demb']

        def dec_wgrads(out, beta):
            ops.gemm(Bw['dq_pre'].view(LB, A), hs0_prev, out['attention.phi.weight'], transA=True, beta=beta)
            for l in range(NL):
                dg = Bw['dgates'][l].view(LB, 4 * C)
                # with dropout the cells saw the dropped copies (asr.py:353,355)
                x_l = (S['xdrop'] if 'xdrop' in S else S['xin']).view(LB, XI) if l == 0 else S['hs'][l - 1, 1:].reshape(LB, C)
                h_l = S['hdrop'][l] if (l > 0 and 'hdrop' in S) else S['hs'][l, :L]
                ops.gemm(dg, x_l, out[f'decoder.layer{l}.weight_ih'], transA=True, beta=beta)
                ops.gemm(dg, h_l.reshape(LB, C), out[f'decoder.layer{l}.weight_hh'], transA=True, beta=beta)
                ops.colsum(dg, out[f'decoder.layer{l}.bias_ih'], beta=beta, out2=out[f'decoder.layer{l}.bias_hh'])

        if direct:
            ops.on_side_stream(lambda: dec_wgrads(tg, 1.0), [Bw['dq_pre'], Bw['dgates'], S['xin'], S['hs']])
        else:
            g['attention.phi.weight'] = torch.empty(A, C, **f32)
            for l in range(NL):
                g[f'decoder.layer{l}.weight_ih'] = torch.empty(4 * C, XI if l == 0 else C, **f32)
                g[f'decoder.layer{l}.weight_hh'] = torch.empty(4 * C, C, **f32)
                g[f'decoder.layer{l}.bias_ih'] = torch.empty(4 * C, **f32)
                g[f'decoder.layer{l}.bias_hh'] = torch.empty(4 * C, **f32)
            dec_wgrads(g, 0.0)
        # d enc[b] [T',E] = att[:,b]^T [T' x L] * dctx[:,b] [L x E]
        att = S['att'][1:]
        d_enc = torch.empty(B, Tp, E, **f32)
        ops.gemm(att, Bw['dxin'][:, :, C:], d_enc, transA=True, M=Tp, N=E, K=L, lda=B * Tp, ldb=B * XI, ldc=E,
                 batch=B, sA=Tp, sB=XI, sC=Tp * E)
        if loc:
            d_psi = Bw['dpsi']
            if not split:
                red = ops.colsum(Bw['acc'], torch.empty(accf, **f32))
                g.update({n: v.contiguous() for n, v in loc_grads(red).items()})
        else:
            d_psi = torch.empty(B, Tp, A, **f32)
            ops.gemm(Bw['de'], S['q'], d_psi, transA=True, M=Tp, N=A, K=L, lda=B * Tp, ldb=B * A, ldc=A, batch=B,
                     sA=Tp, sB=A, sC=Tp * A)
        ctx.S = None
        wg = [g.get(n) for n in names]
        return (d_enc, d_psi, None, None, None, None, None, None, None, *wg)
