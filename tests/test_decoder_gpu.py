"""GPU parity: attend-and-spell decoder loop (HIP, through the C ABI) vs the CPU oracle's step functions.
f32 mode: atol 3e-5 (attention uses a one-v_exp tanh, abs err ~1e-7 per call); bf16 mode: atol 3e-2."""
import ctypes
import importlib
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = 'cuda:0'


@pytest.fixture(scope='module')
def mods():
    importlib.import_module('end-to-end-asr-pytorch_amd')
    return (importlib.import_module('end-to-end-asr-pytorch_amd.ops'),
            importlib.import_module('end-to-end-asr-pytorch_amd.decoder'))


def rand_weights(rng, V, C, E, A, NL, loc):
    W = {'embed.weight': rng.randn(V, C), 'attention.phi.weight': rng.randn(A, C) / np.sqrt(C),
         'attention.psi.weight': rng.randn(A, E) / np.sqrt(E), 'attention.psi.bias': 0.1 * rng.randn(A),
         'char_trans.weight': rng.randn(V, C) / np.sqrt(C), 'char_trans.bias': 0.1 * rng.randn(V)}
    if loc:
        W['attention.loc_conv.weight'] = rng.randn(10, 1, 201) / np.sqrt(201)
        W['attention.loc_proj.weight'] = rng.randn(A, 10) / np.sqrt(10)
        W['attention.gen_energy.weight'] = rng.randn(1, A) / np.sqrt(A)
        W['attention.gen_energy.bias'] = 0.1 * rng.randn(1)
    for l in range(NL):
        K = C + E if l == 0 else C
        W[f'decoder.layer{l}.weight_ih'] = rng.randn(4 * C, K) / np.sqrt(K)
        W[f'decoder.layer{l}.weight_hh'] = rng.randn(4 * C, C) / np.sqrt(C)
        W[f'decoder.layer{l}.bias_ih'] = 0.1 * rng.randn(4 * C)
        W[f'decoder.layer{l}.bias_hh'] = 0.1 * rng.randn(4 * C)
    return {k: v.astype(np.float32) for k, v in W.items()}


def oracle_decode(W, enc, lens, y, L, NL, mode):
    from oracle import las_ref as R
    Wt = {k: torch.tensor(v) for k, v in W.items()}
    enc = torch.tensor(enc)
    B = enc.shape[0]
    C = W['embed.weight'].shape[1]
    hs = [torch.zeros(B, C) for _ in range(NL)]
    cs = [torch.zeros(B, C) for _ in range(NL)]
    st = R.attention_init(enc, lens, Wt)
    atts, tops, ctxs = [], [], []
    for t in range(L):
        a, ctx = R.attention_step(hs[0], enc, st, Wt, mode)
        top = R.speller_step(torch.cat([Wt['embed.weight'][torch.tensor(y[:, t])], ctx], -1), hs, cs, Wt, NL)
        atts.append(a); tops.append(top); ctxs.append(ctx)
    return torch.stack(atts).numpy(), torch.stack(tops).numpy(), torch.stack(ctxs).numpy(), st['psi'].numpy()


@pytest.mark.parametrize('prec', ['f32', 'bf16'])
@pytest.mark.parametrize('mode,B,Tp,E,A,C,NL,V,L', [('dot', 3, 9, 10, 7, 6, 1, 9, 4), ('loc', 3, 9, 10, 7, 6, 2, 9, 4),
                                                    ('loc', 5, 150, 48, 40, 32, 1, 31, 6), ('dot', 20, 75, 64, 32, 64, 1, 63, 5),
                                                    ('loc', 24, 300, 640, 300, 320, 1, 31, 3)])
def test_decoder_forward(mods, prec, mode, B, Tp, E, A, C, NL, V, L):
    ops, dec = mods
    rng = np.random.RandomState(B * 100 + Tp)
    W = rand_weights(rng, V, C, E, A, NL, mode == 'loc')
    lens = sorted(rng.randint(max(2, Tp // 2), Tp + 1, size=B).tolist(), reverse=True); lens[0] = Tp
    enc = np.zeros((B, Tp, E), np.float32)
    for b, l in enumerate(lens):
        enc[b, :l] = np.tanh(rng.randn(l, E))
    y = rng.randint(2, V, size=(B, L + 2)); y[:, 0] = 0
    att_r, top_r, ctx_r, psi_r = oracle_decode(W, enc, lens, y, L, NL, mode)
    Wg = {k: torch.tensor(v, device=DEV) for k, v in W.items()}
    ops.set_precision(prec)
    try:
        S = dec.decoder_forward_raw(Wg, torch.tensor(enc, device=DEV), torch.tensor(psi_r, device=DEV),
                                    torch.tensor(lens, dtype=torch.int32, device=DEV), torch.tensor(y, device=DEV), L, NL,
                                    mode == 'loc')
        torch.cuda.synchronize()
    finally:
        ops.set_precision('bf16')
    tol = dict(atol=3e-5, rtol=1e-4) if prec == 'f32' else dict(atol=3e-2, rtol=3e-2)
    np.testing.assert_allclose(S['att'][1:].cpu().numpy(), att_r, **tol)
    np.testing.assert_allclose(S['xin'][:, :, C:].cpu().numpy(), ctx_r, **tol)
    np.testing.assert_allclose(S['hs'][NL - 1, 1:].cpu().numpy(), top_r, **tol)
    assert np.array_equal(S['tok'].cpu().numpy(), y[:, :L].T)


@pytest.mark.parametrize('mode,B,Tp,E,A,C,NL,V,L', [('loc', 5, 150, 48, 40, 32, 1, 31, 6), ('dot', 20, 75, 64, 32, 64, 2, 63, 5),
                                                    ('loc', 24, 300, 640, 300, 320, 1, 31, 4), ('loc', 4, 45, 16, 70, 8, 2, 9, 3),
                                                    ('dot', 6, 75, 64, 48, 32, 1, 63, 7), ('dot', 8, 75, 128, 256, 64, 1, 63, 5),
                                                    # E > 1024 on the per-step chain (two Speller layers keep it off the persistent loops):
                                                    # att_bwd_step's 8-piece row path (the 6 x 1024 pBLSTM has E = 2048)
                                                    ('loc', 3, 45, 1280, 70, 8, 2, 9, 3), ('dot', 3, 40, 2048, 32, 8, 2, 9, 3)])
def test_decoder_backward(mods, mode, B, Tp, E, A, C, NL, V, L):
    """BPTT of the whole loop (las_decoder_bwd + the post-loop contractions) vs autograd through the oracle's step
    functions, f32 mode, at sizes with several attention chunks / lanes per row (incl. the C2 shape).  Every gradient
    within 2e-4 of its largest entry + 2e-5."""
    from oracle import las_ref as R
    ops, dec = mods
    rng = np.random.RandomState(B * 100 + Tp + 7)
    loc = mode == 'loc'
    W = rand_weights(rng, V, C, E, A, NL, loc)
    lens = sorted(rng.randint(max(2, Tp // 2), Tp + 1, size=B).tolist(), reverse=True); lens[0] = Tp
    enc = np.zeros((B, Tp, E), np.float32)
    for b, l in enumerate(lens):
        enc[b, :l] = np.tanh(rng.randn(l, E))
    psi = np.tanh(rng.randn(B, Tp, A)).astype(np.float32)
    y = rng.randint(2, V, size=(B, L + 2)); y[:, 0] = 0
    G = rng.randn(L, B, C).astype(np.float32)
    # ---- oracle
    Wt = {k: torch.tensor(v, requires_grad=True) for k, v in W.items()}
    enc_t, psi_t = torch.tensor(enc, requires_grad=True), torch.tensor(psi, requires_grad=True)
    hs = [torch.zeros(B, C) for _ in range(NL)]
    cs = [torch.zeros(B, C) for _ in range(NL)]
    st = R.attention_init(enc_t, lens, Wt)
    st['psi'] = psi_t
    tops = []
    for t in range(L):
        a, ctx = R.attention_step(hs[0], enc_t, st, Wt, mode)
        tops.append(R.speller_step(torch.cat([Wt['embed.weight'][torch.tensor(y[:, t])], ctx], -1), hs, cs, Wt, NL))
    (torch.stack(tops) * torch.tensor(G)).sum().backward()
    # ---- HIP
    names = dec.weight_names(NL, loc)
    Wg = {k: torch.tensor(W[k], device=DEV, requires_grad=True) for k in names}
    enc_g = torch.tensor(enc, device=DEV, requires_grad=True)
    psi_g = torch.tensor(psi, device=DEV, requires_grad=True)
    ops.set_precision('f32')
    try:
        h_top, att = dec.DecoderFn.apply(enc_g, psi_g, torch.tensor(lens, dtype=torch.int32, device=DEV),
                                         torch.tensor(y, device=DEV), L, NL, loc, None, 0, *[Wg[k] for k in names])
        (h_top * torch.tensor(G, device=DEV)).sum().backward()
        ops.join_side_stream()
        torch.cuda.synchronize()
    finally:
        ops.set_precision('bf16')

    def near(got, ref, what):
        ref = ref.detach().numpy()
        got = got.detach().cpu().numpy()
        err, lim = np.abs(got - ref).max(), 2e-5 + 2e-4 * np.abs(ref).max()
        assert err <= lim, (what, float(err), float(lim))
    if mode == 'dot' and NL == 1:
        assert dec.DecoderFn.last_pk_bwd_ws is not None, 'dot attention with one Speller layer was meant to take the persistent loops'
    near(h_top, torch.stack(tops), 'h_top')
    near(enc_g.grad, enc_t.grad, 'd enc')
    near(psi_g.grad, psi_t.grad, 'd psi')
    for k in names:
        if k.startswith('char_trans'):
            continue                                      # not used inside the loop
        near(Wg[k].grad, Wt[k].grad, k)


@pytest.mark.parametrize('NL', [1, 2])
def test_decoder_dropout(mods, NL):
    """Speller dropout (asr.py:327,353,355): the masks the kernels draw (counter hash, exported through
    las_dropout_rows / las_decoder_drop_seed) are replayed in the oracle; forward states and every gradient must then
    agree as in the dropout-free test.  Also: keep rate and scaling of the mask itself."""
    from oracle import las_ref as R
    ops, dec = mods
    lib = importlib.import_module('end-to-end-asr-pytorch_amd._lib')
    L_ = lib.lib()
    L_.las_decoder_drop_seed.restype = ctypes.c_uint
    mode, B, Tp, E, A, C, V, L, p, dseed = 'loc', 5, 40, 24, 20, 16, 13, 5, 0.3, 1234567
    rng = np.random.RandomState(77 + NL)
    W = rand_weights(rng, V, C, E, A, NL, True)
    lens = sorted(rng.randint(Tp // 2, Tp + 1, size=B).tolist(), reverse=True); lens[0] = Tp
    enc = np.zeros((B, Tp, E), np.float32)
    for b, l in enumerate(lens):
        enc[b, :l] = np.tanh(rng.randn(l, E))
    psi = np.tanh(rng.randn(B, Tp, A)).astype(np.float32)
    y = rng.randint(2, V, size=(B, L + 2)); y[:, 0] = 0
    G = rng.randn(L, B, C).astype(np.float32)

    def mask(t, l, n):
        ones = torch.ones(B, n, device=DEV)
        out = torch.empty_like(ones)
        seed = L_.las_decoder_drop_seed(ctypes.c_uint(dseed), lib.I(t), lib.I(l))
        lib.check(L_.las_dropout_rows(lib.P(ones.data_ptr()), ctypes.c_int64(n), lib.P(out.data_ptr()), ctypes.c_int64(n), lib.I(B), lib.I(n),
                                      ctypes.c_float(p), ctypes.c_uint(seed), lib.cur_stream()), 'dropout')
        return out.cpu()
    m0 = mask(0, 0, 4096 // B * B // B)          # statistics on a larger draw
    big = torch.ones(64, 4096, device=DEV); bo = torch.empty_like(big)
    lib.check(L_.las_dropout_rows(lib.P(big.data_ptr()), ctypes.c_int64(4096), lib.P(bo.data_ptr()), ctypes.c_int64(4096), lib.I(64), lib.I(4096),
                                  ctypes.c_float(p), ctypes.c_uint(99), lib.cur_stream()), 'dropout')
    keep = (bo > 0).float().mean().item()
    assert abs(keep - (1 - p)) < 5e-3 and torch.allclose(bo[bo > 0], torch.tensor(1 / (1 - p), device=DEV))
    # ---- oracle with the replayed masks
    Wt = {k: torch.tensor(v, requires_grad=True) for k, v in W.items()}
    enc_t, psi_t = torch.tensor(enc, requires_grad=True), torch.tensor(psi, requires_grad=True)
    hs = [torch.zeros(B, C) for _ in range(NL)]
    cs = [torch.zeros(B, C) for _ in range(NL)]
    st = R.attention_init(enc_t, lens, Wt)
    st['psi'] = psi_t
    tops = []
    for t in range(L):
        a, ctx = R.attention_step(hs[0], enc_t, st, Wt, mode)
        masks = [mask(t, 0, C + E)] + [mask(t, l, C) for l in range(1, NL)]
        tops.append(R.speller_step(torch.cat([Wt['embed.weight'][torch.tensor(y[:, t])], ctx], -1), hs, cs, Wt, NL, masks))
    (torch.stack(tops) * torch.tensor(G)).sum().backward()
    # ---- HIP
    names = dec.weight_names(NL, True)
    Wg = {k: torch.tensor(W[k], device=DEV, requires_grad=True) for k in names}
    enc_g = torch.tensor(enc, device=DEV, requires_grad=True)
    psi_g = torch.tensor(psi, device=DEV, requires_grad=True)
    ops.set_precision('f32')
    try:
        h_top, att = dec.DecoderFn.apply(enc_g, psi_g, torch.tensor(lens, dtype=torch.int32, device=DEV), torch.tensor(y, device=DEV),
                                         L, NL, True, None, (0, p, dseed), *[Wg[k] for k in names])
        (h_top * torch.tensor(G, device=DEV)).sum().backward()
        ops.join_side_stream()
        torch.cuda.synchronize()
    finally:
        ops.set_precision('bf16')

    def near(got, ref, what):
        ref = ref.detach().numpy(); got = got.detach().cpu().numpy()
        err, lim = np.abs(got - ref).max(), 2e-5 + 2e-4 * np.abs(ref).max()
        assert err <= lim, (what, float(err), float(lim))
    if mode == 'dot' and NL == 1:
        assert dec.DecoderFn.last_pk_bwd_ws is not None, 'dot attention with one Speller layer was meant to take the persistent loops'
    near(h_top, torch.stack(tops), 'h_top')
    near(enc_g.grad, enc_t.grad, 'd enc')
    near(psi_g.grad, psi_t.grad, 'd psi')
    for k in names:
        if not k.startswith('char_trans'):
            near(Wg[k].grad, Wt[k].grad, k)


def decode_s16(x):
    """The saved s of bf16 mode (csrc/las_common.h): bf16 bits of copysign(1 - |s|, s) -> s."""
    u = x.view(np.uint16).astype(np.uint32)
    t = ((u & 0x7fff) << 16).view(np.float32)
    return np.where(u & 0x8000, -(1.0 - t), 1.0 - t).astype(np.float32)


@pytest.mark.parametrize('prec,B,Tp,E,A,C,V,L', [('f32', 5, 150, 48, 40, 32, 31, 6), ('bf16', 5, 150, 48, 40, 32, 31, 6),
                                                 ('f32', 12, 77, 96, 130, 64, 17, 5), ('f32', 3, 9, 10, 7, 6, 9, 4),
                                                 ('bf16', 24, 300, 640, 300, 320, 31, 9), ('bf16', 12, 300, 640, 300, 320, 31, 5),
                                                 ('bf16', 30, 201, 256, 512, 128, 40, 4)])
def test_persistent_loop_matches_per_step_path(mods, prec, B, Tp, E, A, C, V, L):
    """The one-launch loop (decoder_pk.hip: cell / attention roles handing off through the L2) against the four
    launches per step of decoder.hip on the same inputs: every saved tensor the backward pass reads.  f32: 2e-5 (only
    summation orders differ); bf16: 2e-2 (the persistent loop additionally keeps enc and the exchanged h / ctx in bf16,
    the MFMA operand format of this mode).  Shapes: several T'-chunks and E-slices per utterance incl. empty ones, the
    C2 / C3 decoder (B = 24 as two batch slices, and the half batch 12), B = 30 with A = 512 (two batch tiles per slice)."""
    ops, dec = mods
    rng = np.random.RandomState(B * 1000 + Tp + L)
    W = rand_weights(rng, V, C, E, A, 1, True)
    lens = sorted(rng.randint(max(2, Tp // 2), Tp + 1, size=B).tolist(), reverse=True); lens[0] = Tp
    enc = np.zeros((B, Tp, E), np.float32)
    for b, l in enumerate(lens):
        enc[b, :l] = np.tanh(rng.randn(l, E))
    psi = np.tanh(rng.randn(B, Tp, A)).astype(np.float32)
    y = rng.randint(2, V, size=(B, L + 2)); y[:, 0] = 0
    Wg = {k: torch.tensor(v, device=DEV) for k, v in W.items()}
    args = (Wg, torch.tensor(enc, device=DEV), torch.tensor(psi, device=DEV), torch.tensor(lens, dtype=torch.int32, device=DEV),
            torch.tensor(y, device=DEV), L, 1, True)
    ops.set_precision(prec)
    try:
        S1 = dec.decoder_forward_raw(*args, persistent=True)
        S0 = dec.decoder_forward_raw(*args, persistent=False)
        torch.cuda.synchronize()
    finally:
        ops.set_precision('bf16')
    assert 'pk_ws' in S1 and 'pk_ws' not in S0, 'the persistent launch must be the path taken for this shape'
    assert int(S1['pk_status'].item()) == 0
    tol = dict(atol=2e-5, rtol=1e-4) if prec == 'f32' else dict(atol=2e-2, rtol=2e-2)
    for k in ['q', 'att', 'xin', 'hs', 'cs', 'gates', 'f', 's']:
        a1, a0 = S1[k].cpu().numpy(), S0[k].cpu().numpy()
        if k == 's' and a1.dtype == np.int16:          # bf16 mode saves s as a 16-bit code (las_common.h): decode both
            a1, a0 = decode_s16(a1), decode_s16(a0)
        if k == 's':                                   # only frames inside the utterance are defined
            for b, l in enumerate(lens):
                np.testing.assert_allclose(a1[:, b, :l], a0[:, b, :l], err_msg=k, **tol)
        else:
            np.testing.assert_allclose(a1, a0, err_msg=k, **tol)
    assert np.array_equal(S1['tok'].cpu().numpy(), S0['tok'].cpu().numpy())


@pytest.mark.parametrize('prec,B,Tp,E,A,C,V,L', [('f32', 5, 150, 48, 40, 32, 31, 6), ('bf16', 5, 150, 48, 40, 32, 31, 6),
                                                 ('f32', 12, 77, 96, 130, 64, 17, 5), ('f32', 3, 9, 16, 7, 6, 9, 4),
                                                 ('bf16', 24, 300, 640, 300, 320, 31, 7), ('bf16', 12, 300, 640, 300, 320, 31, 4)])
def test_persistent_bptt_matches_per_step_path(mods, prec, B, Tp, E, A, C, V, L):
    """The one-launch BPTT chain (decoder_pk_bwd.hip) against the three launches per step of decoder_bwd.hip, both fed by
    the SAME persistent forward: every gradient DecoderFn returns.  f32: 2e-5 + 2e-4 of the largest entry (summation
    orders differ); bf16: 3e-2 of the largest entry (the persistent chain also exchanges the K-split pieces, d q_pre and
    d u in bf16, the MFMA operand format of this mode)."""
    ops, dec = mods
    rng = np.random.RandomState(B * 1000 + Tp + L + 3)
    W = rand_weights(rng, V, C, E, A, 1, True)
    lens = sorted(rng.randint(max(2, Tp // 2), Tp + 1, size=B).tolist(), reverse=True); lens[0] = Tp
    enc = np.zeros((B, Tp, E), np.float32)
    for b, l in enumerate(lens):
        enc[b, :l] = np.tanh(rng.randn(l, E))
    psi = np.tanh(rng.randn(B, Tp, A)).astype(np.float32)
    y = rng.randint(2, V, size=(B, L + 2)); y[:, 0] = 0
    G = rng.randn(L, B, C).astype(np.float32)
    names = dec.weight_names(1, True)
    res = []
    ops.set_precision(prec)
    try:
        for persistent in (True, False):
            dec.DecoderFn.persistent_bwd = persistent
            Wg = {k: torch.tensor(W[k], device=DEV, requires_grad=True) for k in names}
            enc_g = torch.tensor(enc, device=DEV, requires_grad=True)
            psi_g = torch.tensor(psi, device=DEV, requires_grad=True)
            status = torch.zeros(1, dtype=torch.int32, device=DEV)
            h_top, att = dec.DecoderFn.apply(enc_g, psi_g, torch.tensor(lens, dtype=torch.int32, device=DEV),
                                             torch.tensor(y, device=DEV), L, 1, True, None, dict(seed=0, status=status),
                                             *[Wg[k] for k in names])
            (h_top * torch.tensor(G, device=DEV)).sum().backward()
            ops.join_side_stream()
            torch.cuda.synchronize()
            assert int(status.item()) == 0
            res.append(dict({'d enc': enc_g.grad.cpu().numpy(), 'd psi': psi_g.grad.cpu().numpy()},
                            **{k: Wg[k].grad.cpu().numpy() for k in names if not k.startswith('char_trans')}))
    finally:
        dec.DecoderFn.persistent_bwd = True
        ops.set_precision('bf16')
    for k in res[0]:
        ref, got = res[1][k], res[0][k]
        lim = (2e-5 + 2e-4 * np.abs(ref).max()) if prec == 'f32' else 3e-2 * np.abs(ref).max() + 1e-6
        if k == 'attention.gen_energy.bias' and prec == 'bf16':
            continue            # d b_e = sum of d e, which the softmax makes cancel to 0: what is left is rounding, not signal
                                # (the per-step path's `dot` from the saved context leaves 2e-2 at the C2 shape, this one 2e-6)
        assert np.abs(got - ref).max() <= lim, (k, float(np.abs(got - ref).max()), float(lim))


@pytest.mark.parametrize('mode,E', [('loc', 2048), ('loc', 640), ('dot', 1280)])
def test_per_step_backward_reads_enc_as_bf16(mods, mode, E):
    """bf16 mode, per-step BPTT chain (two Speller layers): with enc's bf16 twin attached (as Seq2Seq.forward attaches it) att_bwd_step
    takes d a = enc . d ctx from the bf16 rows (its EB variants; E = 2 048 is the 6 x 1024 pBLSTM's).  Against the same chain on the
    fp32 rows: every gradient within 1e-2 of its largest entry (enc rounded to bf16 in ONE product of the chain)."""
    ops, dec = mods
    B, Tp, A, C, V, L, NL = 4, 50, 64, 32, 31, 5, 2
    rng = np.random.RandomState(E + len(mode))
    loc = mode == 'loc'
    W = rand_weights(rng, V, C, E, A, NL, loc)
    lens = sorted(rng.randint(Tp // 2, Tp + 1, size=B).tolist(), reverse=True); lens[0] = Tp
    enc = np.zeros((B, Tp, E), np.float32)
    for b, l in enumerate(lens):
        enc[b, :l] = np.tanh(rng.randn(l, E))
    psi = np.tanh(rng.randn(B, Tp, A)).astype(np.float32)
    y = rng.randint(2, V, size=(B, L + 2)); y[:, 0] = 0
    G = rng.randn(L, B, C).astype(np.float32)
    names = dec.weight_names(NL, loc)
    res = []
    ops.set_precision('bf16')
    for with_twin in (False, True):
        Wg = {k: torch.tensor(W[k], device=DEV, requires_grad=True) for k in names}
        enc_g = torch.tensor(enc, device=DEV, requires_grad=True)
        if with_twin:
            enc_g._bf16 = enc_g.detach().to(torch.bfloat16)
        psi_g = torch.tensor(psi, device=DEV, requires_grad=True)
        h_top, att = dec.DecoderFn.apply(enc_g, psi_g, torch.tensor(lens, dtype=torch.int32, device=DEV),
                                         torch.tensor(y, device=DEV), L, NL, loc, None, 0, *[Wg[k] for k in names])
        (h_top * torch.tensor(G, device=DEV)).sum().backward()
        ops.join_side_stream()
        torch.cuda.synchronize()
        assert dec.DecoderFn.last_pk_bwd_ws is None
        res.append(dict({'d enc': enc_g.grad.cpu().numpy(), 'd psi': psi_g.grad.cpu().numpy()},
                        **{k: Wg[k].grad.cpu().numpy() for k in names if not k.startswith('char_trans')}))
    differs = False
    for k, ref in res[0].items():
        if k == 'attention.gen_energy.bias':
            continue
        err = np.abs(res[1][k] - ref).max()
        differs |= err > 0
        assert err <= 1e-2 * np.abs(ref).max() + 1e-7, (k, float(err), float(np.abs(ref).max()))
    assert differs, 'the bf16 rows were meant to be read (identical results: the fp32 path ran twice)'


def test_odd_attention_dim_bf16_keeps_the_per_step_kernels(mods):
    """bf16 mode saves s = tanh(psi + q + u) as a 16-bit code that the persistent loops and the MFMA post-loop sums access as
    column PAIRS: an odd attention dim keeps the per-step kernels (2-byte accesses) and the VALU post-loop sums.  Their
    gradients against the same loop in fp32 mode: 3e-2 of the largest entry (bf16 operands)."""
    ops, dec = mods
    B, Tp, E, A, C, V, L = 5, 60, 48, 37, 32, 31, 6
    rng = np.random.RandomState(77)
    W = rand_weights(rng, V, C, E, A, 1, True)
    lens = sorted(rng.randint(Tp // 2, Tp + 1, size=B).tolist(), reverse=True); lens[0] = Tp
    enc = np.zeros((B, Tp, E), np.float32)
    for b, l in enumerate(lens):
        enc[b, :l] = np.tanh(rng.randn(l, E))
    psi = np.tanh(rng.randn(B, Tp, A)).astype(np.float32)
    y = rng.randint(2, V, size=(B, L + 2)); y[:, 0] = 0
    G = rng.randn(L, B, C).astype(np.float32)
    names = dec.weight_names(1, True)
    res = {}
    try:
        for prec in ('f32', 'bf16'):
            ops.set_precision(prec)
            Wg = {k: torch.tensor(W[k], device=DEV, requires_grad=True) for k in names}
            enc_g = torch.tensor(enc, device=DEV, requires_grad=True)
            psi_g = torch.tensor(psi, device=DEV, requires_grad=True)
            h_top, att = dec.DecoderFn.apply(enc_g, psi_g, torch.tensor(lens, dtype=torch.int32, device=DEV),
                                             torch.tensor(y, device=DEV), L, 1, True, None, 0, *[Wg[k] for k in names])
            (h_top * torch.tensor(G, device=DEV)).sum().backward()
            ops.join_side_stream()
            torch.cuda.synchronize()
            if prec == 'bf16':
                assert dec.DecoderFn.last_pk_bwd_ws is None, 'an odd attention dim was meant to keep the per-step kernels in bf16 mode'
            res[prec] = dict({'h_top': h_top.detach().cpu().numpy(), 'd enc': enc_g.grad.cpu().numpy(), 'd psi': psi_g.grad.cpu().numpy()},
                             **{k: Wg[k].grad.cpu().numpy() for k in names if not k.startswith('char_trans')})
    finally:
        ops.set_precision('bf16')
    for k, ref in res['f32'].items():
        if k == 'attention.gen_energy.bias':
            continue                    # (cancels to rounding noise)
        err = np.abs(res['bf16'][k] - ref).max()
        assert err <= 3e-2 * np.abs(ref).max() + 1e-6, (k, float(err), float(np.abs(ref).max()))


@pytest.mark.parametrize('B,Tp,E,A,C,V,L', [(5, 150, 48, 40, 32, 31, 6),        # one tile per wave, no step split
                                            (4, 40, 32, 512, 32, 17, 5),        # A = 512: eight waves, four tiles each
                                            (6, 77, 64, 130, 64, 17, 50),       # steps shared by three workgroups per tile
                                            (24, 300, 640, 300, 320, 31, 60)])  # the C2 / C3 decoder shape
def test_loc_post_mfma_matches_valu_kernel(mods, B, Tp, E, A, C, V, L):
    """The post-loop sums d psi, d w_e, d b_e, d W_lp (att_loc_post): bf16 mode runs them on the matrix cores
    (att_loc_post_mma), LAS_LOC_POST_VALU=1 selects the f32-mode kernel.  Same inputs, same BPTT chain before them, ragged
    utterance AND label lengths (the MFMA kernel stops at an utterance's last step with a gradient).  d psi / d w_e are the
    same f32 sums in another order: 1e-5 of the largest entry; d W_lp takes du and f as bf16 MFMA operands: 4e-3."""
    import os
    ops, dec = mods
    rng = np.random.RandomState(B * 77 + A + L)
    W = rand_weights(rng, V, C, E, A, 1, True)
    lens = sorted(rng.randint(max(2, Tp // 2), Tp + 1, size=B).tolist(), reverse=True); lens[0] = Tp
    enc = np.zeros((B, Tp, E), np.float32)
    for b, l in enumerate(lens):
        enc[b, :l] = np.tanh(rng.randn(l, E))
    psi = np.tanh(rng.randn(B, Tp, A)).astype(np.float32)
    y = rng.randint(2, V, size=(B, L + 2)); y[:, 0] = 0
    G = rng.randn(L, B, C).astype(np.float32)
    nlab = rng.randint(max(1, L // 2), L + 1, size=B); nlab[0] = L
    for b in range(B):
        G[nlab[b]:, b] = 0.0                       # no loss, hence no gradient, behind an utterance's last label
    names = dec.weight_names(1, True)
    res = []
    ops.set_precision('bf16')
    old = os.environ.pop('LAS_LOC_POST_VALU', None)
    try:
        for valu in (False, True):
            if valu:
                os.environ['LAS_LOC_POST_VALU'] = '1'
            Wg = {k: torch.tensor(W[k], device=DEV, requires_grad=True) for k in names}
            enc_g = torch.tensor(enc, device=DEV, requires_grad=True)
            psi_g = torch.tensor(psi, device=DEV, requires_grad=True)
            status = torch.zeros(1, dtype=torch.int32, device=DEV)
            h_top, att = dec.DecoderFn.apply(enc_g, psi_g, torch.tensor(lens, dtype=torch.int32, device=DEV),
                                             torch.tensor(y, device=DEV), L, 1, True, None, dict(seed=0, status=status),
                                             *[Wg[k] for k in names])
            (h_top * torch.tensor(G, device=DEV)).sum().backward()
            ops.join_side_stream()
            torch.cuda.synchronize()
            assert int(status.item()) == 0
            res.append({'d psi': psi_g.grad.cpu().numpy(), 'd w_e': Wg['attention.gen_energy.weight'].grad.cpu().numpy(),
                        'd W_lp': Wg['attention.loc_proj.weight'].grad.cpu().numpy()})
    finally:
        os.environ.pop('LAS_LOC_POST_VALU', None)
        if old is not None:
            os.environ['LAS_LOC_POST_VALU'] = old
    for k, rel in (('d psi', 1e-5), ('d w_e', 1e-5), ('d W_lp', 4e-3)):
        got, ref = res[0][k], res[1][k]
        assert np.isfinite(got).all() and np.abs(ref).max() > 0
        assert np.abs(got - ref).max() <= rel * np.abs(ref).max() + 1e-7, (k, float(np.abs(got - ref).max()), float(np.abs(ref).max()))


@pytest.mark.parametrize('B,Tp,E,A,C,V,L', [(24, 300, 640, 300, 320, 31, 6), (16, 77, 96, 130, 64, 17, 5)])
def test_persistent_bptt_exchange_forms_agree(mods, B, Tp, E, A, C, V, L):
    """The persistent loops' exchanges among an utterance's parts (forward: the energies all-gather; BPTT: d a, d q partials, d f):
    L2-local form (XCD-grouped block ids, plain stores (+ progress words), taken when the run-time XCC-id check passes)
    against the sc1 form (LAS_DEC_NO_XL=1).  Same arithmetic in
    the same order: every gradient equal to 1e-6 of its largest entry."""
    import os
    ops, dec = mods
    rng = np.random.RandomState(B * 31 + Tp + L)
    W = rand_weights(rng, V, C, E, A, 1, True)
    lens = sorted(rng.randint(max(2, Tp // 2), Tp + 1, size=B).tolist(), reverse=True); lens[0] = Tp
    enc = np.zeros((B, Tp, E), np.float32)
    for b, l in enumerate(lens):
        enc[b, :l] = np.tanh(rng.randn(l, E))
    psi = np.tanh(rng.randn(B, Tp, A)).astype(np.float32)
    y = rng.randint(2, V, size=(B, L + 2)); y[:, 0] = 0
    G = rng.randn(L, B, C).astype(np.float32)
    names = dec.weight_names(1, True)
    res = []
    ops.set_precision('bf16')
    old = os.environ.pop('LAS_DEC_NO_XL', None)
    try:
        for no_xl in (False, True):
            if no_xl:
                os.environ['LAS_DEC_NO_XL'] = '1'
            Wg = {k: torch.tensor(W[k], device=DEV, requires_grad=True) for k in names}
            enc_g = torch.tensor(enc, device=DEV, requires_grad=True)
            psi_g = torch.tensor(psi, device=DEV, requires_grad=True)
            status = torch.zeros(1, dtype=torch.int32, device=DEV)
            h_top, att = dec.DecoderFn.apply(enc_g, psi_g, torch.tensor(lens, dtype=torch.int32, device=DEV),
                                             torch.tensor(y, device=DEV), L, 1, True, None, dict(seed=0, status=status),
                                             *[Wg[k] for k in names])
            (h_top * torch.tensor(G, device=DEV)).sum().backward()
            ops.join_side_stream()
            torch.cuda.synchronize()
            assert int(status.item()) == 0
            res.append(dict({'h_top': h_top.detach().cpu().numpy(), 'att': att.detach().cpu().numpy(),        # (the forward loop's energies exchange too)
                             'd enc': enc_g.grad.cpu().numpy(), 'd psi': psi_g.grad.cpu().numpy()},
                            **{k: Wg[k].grad.cpu().numpy() for k in names if not k.startswith('char_trans')}))
    finally:
        os.environ.pop('LAS_DEC_NO_XL', None)
        if old is not None:
            os.environ['LAS_DEC_NO_XL'] = old
    for k in res[0]:
        a0, a1 = res[0][k], res[1][k]
        assert np.isfinite(a0).all()
        if k == 'attention.gen_energy.bias':
            continue                     # (a sum that cancels to rounding noise; float atomics in another order)
        assert np.abs(a0 - a1).max() <= 1e-6 * np.abs(a1).max() + 1e-9, (k, float(np.abs(a0 - a1).max()), float(np.abs(a1).max()))


@pytest.mark.parametrize('mode,B,Tp,E,A,C,V,L', [('loc', 12, 150, 128, 96, 64, 31, 12), ('loc', 24, 300, 640, 300, 320, 31, 5),
                                                 ('dot', 8, 75, 512, 256, 256, 63, 8)])      # the last: the TIMIT config's decoder (c1)
def test_persistent_loops_bf16_vs_oracle(mods, mode, B, Tp, E, A, C, V, L):
    """The benchmark's decoder path -- dec_pk_fwd_kernel / dec_pk_bwd_kernel in bf16 mode -- against the ORACLE (autograd
    through attention_step / speller_step), not against another HIP kernel: a mid shape and the C2 / C3 decoder shape, both
    of which take the persistent launches (asserted).  The oracle runs with the operands of the matrix products rounded to
    bf16 as the kernels round them (phi h, W_lp f, the context's enc, the cell's two products), so what is left is the
    rounding of the BACKWARD operands (the exchanged pieces, d q_pre, d u as bf16): forward states to 3e-3, every gradient
    within 6e-3 of its largest entry (measured: 4.3e-3 at worst; the per-step-vs-persistent comparison above allows 3e-2, as does bf16 mode against pure fp32).
    Dot attention (e = psi . q sums A = 256 products and is scaled by 2 before the softmax) has a far larger gain from h to the
    scores than the location-aware form, and the bf16 roundings of the backward operands come back through that gain: measured
    1.07e-2 at worst (d phi), bound 1.5e-2; the same kernels in fp32 mode meet test_decoder_backward's fp32 bound on dot shapes."""
    from oracle import las_ref as R
    ops, dec = mods
    rng = np.random.RandomState(B * 31 + Tp + L)
    W = rand_weights(rng, V, C, E, A, 1, mode == 'loc')
    lens = sorted(rng.randint(max(2, Tp // 2), Tp + 1, size=B).tolist(), reverse=True); lens[0] = Tp
    enc = np.zeros((B, Tp, E), np.float32)
    for b, l in enumerate(lens):
        enc[b, :l] = np.tanh(rng.randn(l, E))
    psi = np.tanh(rng.randn(B, Tp, A)).astype(np.float32)
    y = rng.randint(2, V, size=(B, L + 2)); y[:, 0] = 0
    G = rng.randn(L, B, C).astype(np.float32)
    # ---- oracle, bf16-rounded operands
    Wt = {k: torch.tensor(v, requires_grad=True) for k, v in W.items()}
    enc_t, psi_t = torch.tensor(enc, requires_grad=True), torch.tensor(psi, requires_grad=True)
    hs, cs = [torch.zeros(B, C)], [torch.zeros(B, C)]
    st = R.attention_init(enc_t, lens, Wt)
    st['psi'] = psi_t
    tops, atts = [], []
    for t in range(L):
        a, ctx = R.attention_step(hs[0], enc_t, st, Wt, mode, bf16_operands=True)
        atts.append(a)
        tops.append(R.speller_step(torch.cat([Wt['embed.weight'][torch.tensor(y[:, t])], ctx], -1), hs, cs, Wt, 1, bf16_operands=True))
    (torch.stack(tops) * torch.tensor(G)).sum().backward()
    # ---- HIP, bf16 mode
    names = dec.weight_names(1, mode == 'loc')
    Wg = {k: torch.tensor(W[k], device=DEV, requires_grad=True) for k in names}
    enc_g = torch.tensor(enc, device=DEV, requires_grad=True)
    psi_g = torch.tensor(psi, device=DEV, requires_grad=True)
    status = torch.zeros(1, dtype=torch.int32, device=DEV)
    ops.set_precision('bf16')
    h_top, att = dec.DecoderFn.apply(enc_g, psi_g, torch.tensor(lens, dtype=torch.int32, device=DEV), torch.tensor(y, device=DEV),
                                     L, 1, mode == 'loc', None, dict(seed=0, status=status), *[Wg[k] for k in names])
    (h_top * torch.tensor(G, device=DEV)).sum().backward()
    ops.join_side_stream()
    torch.cuda.synchronize()
    assert int(status.item()) == 0
    assert dec.DecoderFn.last_pk_bwd_ws is not None, 'this shape was meant to take the persistent loops'
    np.testing.assert_allclose(h_top.detach().cpu().numpy(), torch.stack(tops).detach().numpy(), atol=3e-3, rtol=3e-3)
    np.testing.assert_allclose(att.cpu().numpy(), torch.stack(atts).detach().numpy(), atol=3e-3, rtol=3e-3)
    worst = {}
    for k, got, ref in [('d enc', enc_g.grad, enc_t.grad), ('d psi', psi_g.grad, psi_t.grad)] + \
                       [(k, Wg[k].grad, Wt[k].grad) for k in names if not k.startswith('char_trans') and k != 'attention.gen_energy.bias']:
        ref = ref.numpy()
        worst[k] = float(np.abs(got.cpu().numpy() - ref).max() / max(np.abs(ref).max(), 1e-12))
    print('persistent decoder bf16 vs oracle, worst / largest entry:', (B, Tp, E, A, C, L), worst)
    assert max(worst.values()) <= (6e-3 if mode == 'loc' else 1.5e-2), worst
