"""Pin the CPU oracle (oracle/) against golden vectors produced by the imported reference
(tools/gen_golden.py).  CPU only.  Tolerances: fp32, 2e-5 abs / 1e-4 rel unless noted."""
import os
import json
import numpy as np
import torch
import pytest

from oracle import las_ref as R
from oracle.ctc_c import ctc_ref

from conftest import GOLDEN


def G(name):
    return np.load(os.path.join(GOLDEN, name))


def W_of(d, prefix='w.'):
    return {k[len(prefix):]: torch.tensor(d[k]) for k in d.files if k.startswith(prefix)}


def close(a, b, atol=2e-5, rtol=1e-4):
    a = np.asarray(a.detach() if torch.is_tensor(a) else a, dtype=np.float64)
    b = np.asarray(b, dtype=np.float64)
    np.testing.assert_allclose(a, b, atol=atol, rtol=rtol)


@pytest.mark.parametrize('fast', [False, True])
@pytest.mark.parametrize('name,style', [('concat_odd', 'concat'), ('drop_odd', 'drop'), ('sr1', 'concat'),
                                        ('uni_concat3', 'concat')])
def test_rnnlayer(name, style, fast):
    d = G(f'g1_rnnlayer_{name}.npz')
    W = {('L.' + k): v.requires_grad_(True) for k, v in W_of(d).items()}
    x = torch.tensor(d['x'], requires_grad=True)
    y, ol = R.rnn_layer(x, list(d['lens']), W, 'L', int(d['sr']), style, bool(d['bidir']), fast=fast)
    close(y, d['y'])
    assert ol == list(d['out_lens'])
    (y * torch.tensor(d['gy'])).sum().backward()
    close(x.grad, d['gx'])
    for k, v in W.items():
        close(v.grad, d['grad.' + k[2:]])


def test_listener():
    d = G('g1_listener.npz')
    W = {('encoder.' + k): v.requires_grad_(True) for k, v in W_of(d).items()}
    cfg = dict(srs=[2, 2, 1], dims=[8, 8, 8], style='concat', bidir=True)
    x = torch.tensor(d['x'], requires_grad=True)
    y, ol = R.listener(x, list(d['lens']), W, cfg)
    close(y, d['y'])
    assert ol == list(d['out_lens'])
    (y * torch.tensor(d['gy'])).sum().backward()
    close(x.grad, d['gx'])
    for k, v in W.items():
        close(v.grad, d['grad.' + k[len('encoder.'):]])


@pytest.mark.parametrize('name', ['mfcc26', 'fbank40'])
def test_vgg(name):
    """VGGExtractor (asr.py:507-558), incl. the dropped T%4 tail and floor pooling of odd freq dims."""
    d = G(f'g1_vgg_{name}.npz')
    W = {('V.' + k): v.requires_grad_(True) for k, v in W_of(d).items()}
    x = torch.tensor(d['x'], requires_grad=True)
    y, ol = R.vgg_extractor(x, list(d['lens']), W, prefix='V')
    close(y, d['y'])
    assert ol == list(d['out_lens'])
    (y * torch.tensor(d['gy'])).sum().backward()
    close(x.grad, d['gx'])
    for k, v in W.items():
        close(v.grad, d['grad.' + k[2:]], atol=5e-5)


@pytest.mark.parametrize('mode', ['dot', 'loc'])
def test_attention(mode):
    d = G(f'g1_attention_{mode}.npz')
    W = {('attention.' + k): v.requires_grad_(True) for k, v in W_of(d).items()}
    enc = torch.tensor(d['enc'], requires_grad=True)
    st = R.attention_init(enc, list(d['lens']), W)
    hs = [torch.tensor(d[f'h{i}'], requires_grad=True) for i in range(3)]
    loss = 0
    for i in range(3):
        a, c = R.attention_step(hs[i], enc, st, W, mode)
        close(a, d[f'score{i}'])
        close(c, d[f'ctx{i}'])
        loss = loss + (c * torch.tensor(d[f'gc{i}'])).sum() + (a * torch.tensor(d[f'gs{i}'])).sum()
    loss.backward()
    close(enc.grad, d['genc'])
    for i in range(3):
        close(hs[i].grad, d[f'gh{i}'])
    for k, v in W.items():
        close(v.grad, d['grad.' + k[len('attention.'):]])


@pytest.mark.parametrize('nl', [1, 2])
def test_speller(nl):
    d = G(f'g1_speller_l{nl}.npz')
    W = {('decoder.' + k): v.requires_grad_(True) for k, v in W_of(d).items()}
    B, C = d['out0'].shape
    hs = [torch.zeros(B, C) for _ in range(nl)]
    cs = [torch.zeros(B, C) for _ in range(nl)]
    xs = [torch.tensor(d[f'x{i}'], requires_grad=True) for i in range(3)]
    loss = 0
    for i in range(3):
        o = R.speller_step(xs[i], hs, cs, W, nl)
        close(o, d[f'out{i}'])
        loss = loss + (o * torch.tensor(d[f'go{i}'])).sum()
    loss.backward()
    for i in range(3):
        close(xs[i].grad, d[f'gx{i}'])
    for k, v in W.items():
        close(v.grad, d['grad.' + k[len('decoder.'):]])


CTC_CASES = ['basic', 'repeat', 'minimal', 'infeasible', 'wide']


@pytest.mark.parametrize('impl', ['numpy', 'c'])
@pytest.mark.parametrize('name', CTC_CASES)
def test_ctc_lattice(name, impl):
    """nll, log_alpha ('CTC alignments') and d(mean loss)/dlogits vs ATen via the reference call pattern."""
    d = G(f'g2_ctc_{name}.npz')
    if impl == 'numpy' and name == 'wide':
        pytest.skip('pure-python loops: small cases only')
    f = R.ctc_numpy if impl == 'numpy' else ctc_ref
    nll, la, grad = f(d['logits'], d['label'], d['enc_len'], d['tgt_len'])
    fin = np.isfinite(d['nll'])
    close(nll[fin], d['nll'][fin])
    assert np.all(np.isinf(nll[~fin]))
    ref_la = d['log_alpha']
    for b in range(len(nll)):               # compare inside each utterance's own lattice
        T, S = int(d['enc_len'][b]), 2 * int(d['tgt_len'][b]) + 1
        a, r = la[b, :T, :S], ref_la[b, :T, :S]
        assert np.array_equal(np.isinf(a), np.isinf(r))
        close(a[np.isfinite(r)], r[np.isfinite(r)], atol=5e-5)
    B = len(nll)
    scale = 1.0 / (np.maximum(d['tgt_len'], 1) * B)        # reduction='mean'
    g = grad * scale[:, None, None]
    for b in range(B):
        T = int(d['enc_len'][b])
        if fin[b]:
            close(g[b, :T], d['glogits'][b, :T], atol=2e-6)
            close(g[b, T:], 0 * g[b, T:])
        else:
            assert np.all(np.isnan(g[b, :T])) and np.all(np.isnan(d['glogits'][b, :T]))
    loss = np.mean(nll / np.maximum(d['tgt_len'], 1))
    if np.isfinite(d['loss']):
        close(loss, d['loss'])
    else:
        assert np.isinf(loss)


@pytest.mark.parametrize('fast', [False, True])
@pytest.mark.parametrize('name', ['dot_att', 'loc_ctc', 'ctc_only', 'vgg_loc_ctc'])
def test_train_step(name, fast):
    """Whole step: forward, joint loss, backward, clip, 3 optimiser steps (Adam / Adadelta)."""
    import sys
    sys.path.insert(0, os.path.join(os.path.dirname(__file__), '..', 'tools'))
    from gen_golden import TINY
    d = G(f'g3_step_{name}.npz')
    st = R.RefTrainStep(W_of(d), TINY[name], fast=fast)
    unused = [k for k in st.W if ('grad.' + k) not in d.files]
    for it in range(3):
        out = st.step(d['x'], d['y'])
        close(out['loss'], d[f'loss_it{it}'], atol=1e-5)
        if it == 0:
            close(out['att_loss'], d['att_loss'], atol=1e-5)
            close(out['ctc_loss'], d['ctc_loss'], atol=1e-5)
            close(out['grad_norm'], d['grad_norm'], atol=1e-5)
            assert out['aux']['enc_len'] == list(d['enc_len'])
            if 'att_pred' in d.files:
                close(out['aux']['att_pred'], d['att_pred'])
                close(out['aux']['att_map'], d['att_map'])
            if 'ctc_pred' in d.files:
                close(out['aux']['ctc_pred'], d['ctc_pred'])
            for k, g in out['grads'].items():
                close(g, d['grad.' + k], atol=1e-6)
    for k, v in st.W.items():
        if k not in unused:
            close(v, d['w_after.' + k], atol=1e-5)


def test_trainer_trace_losses():
    """loss/train_att of the reference's 3-step Trainer.exec() trace, replayed through the oracle.
    Batches: TimitDataset sorts by length desc and buckets (dataset.py:36-49); DataLoader(shuffle=True)
    order is RNG dependent, so each golden step is matched to the bucket that reproduces it."""
    d = G('g4_trainer_trace.npz')
    cfg = json.load(open(os.path.join(GOLDEN, 'g4_config.json')))
    xs = np.split(d['train_x'], np.cumsum(d['train_xlen'])[:-1])
    ys = np.split(d['train_y'], np.cumsum(d['train_ylen'])[:-1])
    order = list(reversed(np.argsort([len(v) for v in xs])))
    bs = cfg['solver']['batch_size']
    buckets = []
    for b in range(0, len(order), bs):
        idx = order[b:b + bs]
        T = len(xs[idx[0]])
        L = max(len(ys[i]) for i in idx)
        X = np.zeros((len(idx), T, xs[0].shape[1]), np.float32)
        Y = np.zeros((len(idx), L), np.int64)
        for j, i in enumerate(idx):
            X[j, :len(xs[i])] = xs[i]
            Y[j, :len(ys[i])] = ys[i]
        buckets.append((X, Y))
    st = R.RefTrainStep(W_of(d), cfg['asr_model'])
    want = [v for s, n, v in zip(d['trace_step'], d['trace_name'], d['trace_val']) if n == 'loss/train_att']
    for step, w in enumerate(want):
        best = None
        snap = {k: v.detach().clone() for k, v in st.W.items()}
        for X, Y in buckets:
            loss = float(st.forward_loss(X, Y)[0])
            if best is None or abs(loss - w) < abs(best[0] - w):
                best = (loss, X, Y)
        assert abs(best[0] - w) < 2e-5, (step, best[0], w)
        st.step(best[1], best[2])
    for k, v in st.W.items():
        close(v, d['w_after.' + k], atol=2e-5)


def test_ctc_prefix_scores():
    """CTCPrefixScore.init_state / cheap_compute (ctc.py:19-27,65-101) incl. the repeated-token case."""
    from oracle import beam_ref as Bm
    d = G('g5_ctc_prefix.npz')
    lp = d['lp'][0]
    r0 = Bm.ctc_prefix_init(lp)
    close(r0, d['r0'], atol=1e-5)
    cand = [1, 2, 3, 4, 5]
    psi1, r1 = Bm.ctc_prefix_cheap(lp, [], r0, cand)
    close(psi1, d['psi1'], atol=1e-5); close(r1, d['r1'], atol=1e-5)
    psi2, r2 = Bm.ctc_prefix_cheap(lp, [3], r1[cand.index(3)], cand)
    close(psi2, d['psi2'], atol=1e-5); close(r2, d['r2'], atol=1e-5)
    psi3, r3 = Bm.ctc_prefix_cheap(lp, [3, 3], r2[cand.index(3)], [2, 3, 5])
    close(psi3, d['psi3'], atol=1e-5); close(r3, d['r3'], atol=1e-5)


@pytest.mark.parametrize('name', ['loc_ctc_b1', 'loc_ctc_b3', 'dot_att_b1', 'dot_att_b3'])
def test_beam_decode(name):
    """Seq2Seq.beam_decode (asr.py:155-258): same hypotheses in the same order, per-token scores within 2e-5."""
    import sys
    sys.path.insert(0, os.path.join(os.path.dirname(__file__), '..', 'tools'))
    from gen_golden import TINY
    from oracle import beam_ref as Bm
    d = G(f'g6_beam_{name}.npz')
    cfg = R.parse_cfg(TINY[name.rsplit('_', 1)[0]])
    hyps = Bm.beam_decode(W_of(d), cfg, torch.tensor(d['x']), int(d['steps']), int(d['beam']))
    assert len(hyps) == int(d['n_hyps'])
    for i, (seq, scores) in enumerate(hyps):
        assert seq == d[f'hyp{i}.seq'].tolist(), (i, seq)
        close(np.array(scores), d[f'hyp{i}.scores'], atol=2e-5)


@pytest.mark.parametrize('name', ['dot_att', 'loc_ctc'])
def test_scheduled_sampling_replay(name):
    """Scheduled sampling (reference asr.py:95-100) at tf_rate 0.5: g9_sched_* hold the values random.random() returned at
    asr.py:96 and the tokens Categorical.sample() drew at :99 in the reference's own run.  Replaying both through the
    oracle (teacher input where the flip said so, the recorded draw otherwise) reproduces its logits, attention maps,
    losses and every (clipped) gradient."""
    import sys
    sys.path.insert(0, os.path.join(os.path.dirname(__file__), '..', 'tools'))
    from gen_golden import TINY
    d = G(f'g9_sched_{name}.npz')
    cfg = R.parse_cfg(TINY[name])
    W = {k: v.requires_grad_(True) for k, v in W_of(d).items()}
    x, y = torch.tensor(d['x']), torch.tensor(d['y'])
    flips = [bool(v <= 0.5) for v in d['flip_values']]
    assert not all(flips) and any(flips)
    sampled = {int(t): torch.tensor(tok) for t, tok in zip(d['draw_step'], d['draw_tokens'])}
    assert sorted(sampled) == [t for t, f in enumerate(flips) if not f]
    L = int((y != 0).sum(-1).max())
    ctc_pred, enc_len, att_pred, att_map = R.seq2seq_forward(W, cfg, x, L, teacher=y, lens=R.infer_lengths(x), use_teacher=flips,
                                                             sampled=sampled)
    loss, att, ctc = R.joint_loss(ctc_pred, att_pred, y, L, enc_len, cfg['ctc_w'])
    close(att_pred, d['att_pred'])
    close(att_map, d['att_map'])
    close(loss, d['loss'], atol=1e-5)
    loss.backward()
    used = [k for k in W if W[k].grad is not None]
    R.clip_grad_norm([W[k].grad for k in used])
    for k in used:
        close(W[k].grad, d['grad.' + k], atol=1e-6)
