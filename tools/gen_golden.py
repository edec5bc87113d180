#!/usr/bin/env python
"""Generate golden vectors from the *imported* reference (this container only).

Runs /root/reference's own src/asr.py + src/solver.py (PyTorch CPU, fp32) on
small seeded inputs and dumps inputs / weights / outputs / grads as .npz
fixtures under tests/golden/.  The reference never travels to the GPU box;
only these data files do.  Nothing is written under /root/reference
(sys.dont_write_bytecode).

Harness-level shims (reference files untouched), per SURVEY.md §8(c):
  1. sys.modules['editdistance']   - Levenshtein `eval` (postprocess.py:5)
  2. sys.modules['librosa']        - empty stub (preprocess.py:3, unused here)
  3. sys.modules['tensorboardX']   - SummaryWriter recording add_scalars
  4. Tensor.masked_fill_ wrapper   - uint8 mask -> bool (asr.py:418,429,454)

Usage:  python tools/gen_golden.py [--out tests/golden]
"""
import sys
sys.dont_write_bytecode = True
import os
import types
import argparse
import pickle
import random
import tempfile

import numpy as np
import torch

REF = '/root/reference'


# ----------------------------------------------------------------------------- shims
def _install_shims():
    ed = types.ModuleType('editdistance')

    def _eval(a, b):
        a, b = list(a), list(b)
        prev = list(range(len(b) + 1))
        for i, ca in enumerate(a, 1):
            cur = [i]
            for j, cb in enumerate(b, 1):
                cur.append(min(prev[j] + 1, cur[j - 1] + 1, prev[j - 1] + (ca != cb)))
            prev = cur
        return prev[-1]
    ed.eval = _eval
    sys.modules['editdistance'] = ed
    sys.modules['librosa'] = types.ModuleType('librosa')

    tbx = types.ModuleType('tensorboardX')

    class SummaryWriter:
        trace = []

        def __init__(self, *a, **k):
            pass

        def add_scalars(self, name, d, step):
            SummaryWriter.trace.append((int(step), name, {k: float(v) for k, v in d.items()}))

        texts, images = [], []

        def add_image(self, name, img, step):
            SummaryWriter.images.append((int(step), name, np.asarray(img).copy()))

        def add_text(self, name, txt, step):
            SummaryWriter.texts.append((int(step), name, str(txt)))
    tbx.SummaryWriter = SummaryWriter
    sys.modules['tensorboardX'] = tbx

    _orig = torch.Tensor.masked_fill_

    def _mf(self, mask, value):
        if mask.dtype == torch.uint8:
            mask = mask.bool()
        return _orig(self, mask, value)
    torch.Tensor.masked_fill_ = _mf
    return SummaryWriter


def _np(t):
    return t.detach().cpu().numpy().copy()


def _state(module, prefix=''):
    return {prefix + k: _np(v) for k, v in module.state_dict().items()}


def _grads(module, prefix='grad.'):
    return {prefix + k: _np(p.grad) for k, p in module.named_parameters() if p.grad is not None}


def _seed(s):
    random.seed(s)
    np.random.seed(s)
    torch.manual_seed(s)


def _ragged_x(B, T, D, lens, gen):
    x = torch.zeros(B, T, D)
    for b, l in enumerate(lens):
        x[b, :l] = torch.randn(l, D, generator=gen)
    return x


# ----------------------------------------------------------------------------- G1 modules
def g1_rnnlayer(asr, out):
    for name, style, sr, T, lens, bidir in [
        ('concat_odd', 'concat', 2, 11, [11, 9, 6, 3], True),
        ('drop_odd', 'drop', 2, 11, [11, 8, 7, 2], True),
        ('sr1', 'concat', 1, 7, [7, 7, 4], True),
        ('uni_concat3', 'concat', 3, 10, [10, 5], False),
    ]:
        _seed(11)
        gen = torch.Generator().manual_seed(5)
        B, D, H = len(lens), 6, 8
        layer = asr.RNNLayer(D, H, sr, sample_style=style, rnn_cell='LSTM', dropout_rate=0.0, bidir=bidir)
        x = _ragged_x(B, T, D, lens, gen).requires_grad_(True)
        y, _, olen = layer(x, state_len=lens, pack_input=True)
        gy = torch.randn(y.shape, generator=gen)
        (y * gy).sum().backward()
        d = {'x': _np(x), 'lens': np.array(lens), 'y': _np(y), 'out_lens': np.array(olen),
             'gy': _np(gy), 'gx': _np(x.grad), 'sr': np.array(sr), 'bidir': np.array(int(bidir))}
        d.update(_state(layer, 'w.'))
        d.update(_grads(layer))
        np.savez(os.path.join(out, f'g1_rnnlayer_{name}.npz'), **d)


def g1_listener(asr, out):
    _seed(12)
    gen = torch.Generator().manual_seed(6)
    lens = [13, 12, 9, 5]
    B, T, D = 4, 13, 5
    x = _ragged_x(B, T, D, lens, gen).requires_grad_(True)
    enc = asr.Listener(x, enc_type='BiRNN', sample_rate='2_2_1', sample_style='concat',
                       dim='8_8_8', dropout='0_0_0', rnn_cell='LSTM')
    y, olen = enc(x, lens)
    gy = torch.randn(y.shape, generator=gen)
    (y * gy).sum().backward()
    d = {'x': _np(x), 'lens': np.array(lens), 'y': _np(y), 'out_lens': np.array(olen),
         'gy': _np(gy), 'gx': _np(x.grad)}
    d.update(_state(enc, 'w.'))
    d.update(_grads(enc))
    np.savez(os.path.join(out, 'g1_listener.npz'), **d)


def g1_vgg(asr, out):
    """VGGExtractor (asr.py:507-558): MFCC-style 2x13 and fbank-style 1x40 inputs, T not a multiple of 4."""
    for name, B, T, D in [('mfcc26', 2, 11, 26), ('fbank40', 2, 9, 40)]:
        _seed(15)
        gen = torch.Generator().manual_seed(10)
        lens = [T, T - 3]
        x = _ragged_x(B, T, D, lens, gen).requires_grad_(True)
        vgg = asr.VGGExtractor(x)
        y, olen = vgg(x, lens)
        gy = torch.randn(y.shape, generator=gen)
        (y * gy).sum().backward()
        d = {'x': _np(x), 'lens': np.array(lens), 'y': _np(y), 'out_lens': np.array(olen), 'gy': _np(gy), 'gx': _np(x.grad)}
        d.update(_state(vgg, 'w.'))
        d.update(_grads(vgg))
        np.savez(os.path.join(out, f'g1_vgg_{name}.npz'), **d)


def g1_attention(asr, out):
    for mode in ['dot', 'loc']:
        _seed(13)
        gen = torch.Generator().manual_seed(7)
        B, Tp, E, C, A = 3, 9, 10, 6, 7
        lens = [9, 6, 4]
        att = asr.Attention(E, C, att_mode=mode, dim=A, proj=True, num_head=1)
        enc = _ragged_x(B, Tp, E, lens, gen).requires_grad_(True)
        hs = [torch.randn(B, C, generator=gen).requires_grad_(True) for _ in range(3)]
        att.reset_enc_mem()
        scores, ctxs = [], []
        for h in hs:
            s, c = att(h, enc, lens)
            scores.append(s[0])
            ctxs.append(c)
        gc = [torch.randn(B, E, generator=gen) for _ in range(3)]
        gs = [torch.randn(B, Tp, generator=gen) for _ in range(3)]
        loss = sum((c * g).sum() for c, g in zip(ctxs, gc)) + sum((s * g).sum() for s, g in zip(scores, gs))
        loss.backward()
        d = {'enc': _np(enc), 'lens': np.array(lens), 'genc': _np(enc.grad)}
        for i in range(3):
            d[f'h{i}'] = _np(hs[i]); d[f'gh{i}'] = _np(hs[i].grad)
            d[f'score{i}'] = _np(scores[i]); d[f'ctx{i}'] = _np(ctxs[i])
            d[f'gc{i}'] = _np(gc[i]); d[f'gs{i}'] = _np(gs[i])
        d.update(_state(att, 'w.'))
        d.update(_grads(att))
        np.savez(os.path.join(out, f'g1_attention_{mode}.npz'), **d)


def g1_speller(asr, out):
    for nl in [1, 2]:
        _seed(14)
        gen = torch.Generator().manual_seed(8)
        B, I, C = 3, 11, 6
        sp = asr.Speller(I, dim=C, layer=nl, rnn_cell='LSTMCell', dropout=0.0)
        ctx0 = torch.zeros(B, 4)
        sp.init_rnn(ctx0)
        xs = [torch.randn(B, I, generator=gen).requires_grad_(True) for _ in range(3)]
        outs = [sp(x) for x in xs]
        go = [torch.randn(B, C, generator=gen) for _ in range(3)]
        sum((o * g).sum() for o, g in zip(outs, go)).backward()
        d = {}
        for i in range(3):
            d[f'x{i}'] = _np(xs[i]); d[f'gx{i}'] = _np(xs[i].grad)
            d[f'out{i}'] = _np(outs[i]); d[f'go{i}'] = _np(go[i])
        d.update(_state(sp, 'w.'))
        d.update(_grads(sp))
        np.savez(os.path.join(out, f'g1_speller_l{nl}.npz'), **d)


# ----------------------------------------------------------------------------- G2 CTC
def g2_ctc(out):
    """solver.py:93,160 call pattern: CTCLoss(blank=0,'mean')(log_softmax(pred^T), label2d, LongTensor(enc_len), tgt_len)."""
    import torch.nn.functional as F
    cases = {
        # name: (T', V, enc_len, labels (without padding; include <eos>=1))
        'basic': (12, 7, [12, 10, 8], [[3, 4, 5, 1], [2, 2, 6, 1], [5, 1]]),
        'repeat': (10, 5, [10, 9], [[2, 2, 2, 1], [3, 3, 4, 4, 1]]),
        'minimal': (7, 6, [7, 5], [[2, 3, 4, 1], [2, 2, 1]]),           # T' == len + repeats
        'infeasible': (6, 6, [6, 4], [[2, 3, 1], [2, 2, 3, 1]]),          # 2nd: needs 5 > 4 -> inf
        'wide': (20, 300, [20, 17, 15, 9], [[17, 250, 3, 3, 99, 1], [5, 1], [299, 298, 297, 1], [2, 1]]),
    }
    for name, (Tp, V, enc_len, labs) in cases.items():
        gen = torch.Generator().manual_seed(21)
        B = len(labs)
        L = max(len(l) for l in labs)
        label = torch.zeros(B, L, dtype=torch.long)
        for b, l in enumerate(labs):
            label[b, :len(l)] = torch.tensor(l)
        logits = (2.0 * torch.randn(B, Tp, V, generator=gen)).requires_grad_(True)
        tgt_len = (label != 0).sum(-1)
        lp = F.log_softmax(logits.transpose(0, 1), dim=-1)
        loss = torch.nn.CTCLoss(blank=0, reduction='mean')(lp, label, torch.LongTensor(enc_len), tgt_len)
        loss.backward()
        nll, log_alpha = torch._ctc_loss(lp.detach(), label, enc_len, tgt_len.tolist(), 0, False)
        np.savez(os.path.join(out, f'g2_ctc_{name}.npz'),
                 logits=_np(logits), label=_np(label), enc_len=np.array(enc_len), tgt_len=_np(tgt_len),
                 loss=_np(loss), nll=_np(nll), log_alpha=_np(log_alpha), glogits=_np(logits.grad))


# ----------------------------------------------------------------------------- G3 step level
TINY = {
    'dot_att': dict(
        optimizer=dict(type='Adam', learning_rate=0.001, joint_ctc=0.0),
        encoder=dict(enc_type='BiRNN', sample_rate='2_2_1', sample_style='concat', dim='8_8_8',
                     dropout='0_0_0', rnn_cell='LSTM'),
        attention=dict(att_mode='dot', dim=8, proj=True, num_head=1),
        decoder=dict(dim=8, layer=1, dropout=0, rnn_cell='LSTMCell')),
    'loc_ctc': dict(
        optimizer=dict(type='Adadelta', learning_rate=1.0, joint_ctc=0.5),
        encoder=dict(enc_type='BiRNN', sample_rate='2_1', sample_style='drop', dim='8_12',
                     dropout='0_0', rnn_cell='LSTM'),
        attention=dict(att_mode='loc', dim=6, proj=True, num_head=1),
        decoder=dict(dim=8, layer=2, dropout=0.0, rnn_cell='LSTMCell')),
    'vgg_loc_ctc': dict(     # the structure of the shipped config/libri_example.yaml, tiny
        optimizer=dict(type='Adadelta', learning_rate=1.0, joint_ctc=0.5),
        encoder=dict(enc_type='VGGBiRNN', sample_rate='1_1', sample_style='drop', dim='8_8', dropout='0_0', rnn_cell='LSTM'),
        attention=dict(att_mode='loc', dim=6, proj=True, num_head=1),
        decoder=dict(dim=8, layer=1, dropout=0.0, rnn_cell='LSTMCell')),
    'ctc_only': dict(
        optimizer=dict(type='Adam', learning_rate=0.001, joint_ctc=1.0),
        encoder=dict(enc_type='BiRNN', sample_rate='2_1', sample_style='concat', dim='8_8',
                     dropout='0_0', rnn_cell='LSTM'),
        attention=dict(att_mode='dot', dim=8, proj=True, num_head=1),
        decoder=dict(dim=8, layer=1, dropout=0, rnn_cell='LSTMCell')),
}


def _synth_batch(B, T, D, V, Lmax, gen, min_frac=0.6):
    lens = sorted([T] + [int(torch.randint(int(np.ceil(min_frac * T)), T + 1, (1,), generator=gen)) for _ in range(B - 1)],
                  reverse=True)
    x = _ragged_x(B, T, D, lens, gen)
    ns = [int(torch.randint(max(1, Lmax // 2), Lmax + 1, (1,), generator=gen)) for _ in range(B)]
    y = torch.zeros(B, max(ns) + 2, dtype=torch.long)
    for b, n in enumerate(ns):
        y[b, 1:n + 1] = torch.randint(2, V, (n,), generator=gen)
        y[b, n + 1] = 1
    return x, y, lens


def ref_step(asr, cfg, x, y, V, n_opt_steps=1, tf_rate=1.0, record=None):
    """The arithmetic of Trainer.exec's step body, solver.py:132-182.  tf_rate < 1 (scheduled sampling, asr.py:95-100):
    `record` receives the values random.random() returned at asr.py:96 and the tokens Categorical.sample() drew at :99."""
    import torch.nn.functional as F
    model = asr.Seq2Seq(x, V, cfg)
    if record is not None:
        real_random, real_cat = asr.random.random, asr.Categorical

        def rec_random():
            v = real_random()
            record['flip_values'].append(v)
            return v

        class RecCategorical(real_cat):
            def sample(self, *a, **k):
                t = super().sample(*a, **k)
                record['draws'].append((len(record['flip_values']) - 1, t.clone()))      # the draw that follows flip number ...
                return t
        asr.random.random, asr.Categorical = rec_random, RecCategorical
    w0 = _state(model, 'w.')
    seq_loss = torch.nn.CrossEntropyLoss(ignore_index=0, reduction='none')
    ctc_loss_f = torch.nn.CTCLoss(blank=0, reduction='mean')
    ctc_w = cfg['optimizer']['joint_ctc']
    opt = getattr(torch.optim, cfg['optimizer']['type'])(model.parameters(), lr=cfg['optimizer']['learning_rate'], eps=1e-8)
    rec = {}
    for it in range(n_opt_steps):
        state_len = np.sum(np.sum(x.numpy(), axis=-1) != 0, axis=-1)
        state_len = [int(s) for s in state_len]
        ans_len = int(torch.max(torch.sum(y != 0, dim=-1)))
        opt.zero_grad()
        ctc_pred, enc_len, att_pred, att_maps = model(x, ans_len, tf_rate=tf_rate, teacher=y, state_len=state_len)
        if record is not None:
            asr.random.random, asr.Categorical = real_random, real_cat
        label = y[:, 1:ans_len + 1].contiguous()
        att_loss, ctc_loss = 0, 0
        if ctc_w < 1:
            b, t, c = att_pred.shape
            att_loss = seq_loss(att_pred.view(b * t, c), label.view(-1))
            att_loss = torch.sum(att_loss.view(b, t), dim=-1) / torch.sum(y != 0, dim=-1).float()
            att_loss = torch.mean(att_loss)
        if ctc_w > 0:
            target_len = torch.sum(y != 0, dim=-1)
            ctc_loss = ctc_loss_f(F.log_softmax(ctc_pred.transpose(0, 1), dim=-1), label, torch.LongTensor(enc_len), target_len)
        loss = (1 - ctc_w) * att_loss + ctc_w * ctc_loss
        loss.backward()
        gn = torch.nn.utils.clip_grad_norm_(model.parameters(), 5)
        if it == 0:
            rec.update({'att_loss': np.array(float(att_loss)), 'ctc_loss': np.array(float(ctc_loss)),
                        'loss': np.array(float(loss)), 'grad_norm': np.array(float(gn)),
                        'enc_len': np.array(enc_len)})
            if att_pred is not None:
                rec['att_pred'] = _np(att_pred)
                rec['att_map'] = _np(att_maps[0])
            if ctc_pred is not None:
                rec['ctc_pred'] = _np(ctc_pred)
            rec.update(_grads(model))          # clipped grads (post clip_grad_norm_)
        opt.step()
        rec[f'loss_it{it}'] = np.array(float(loss))
    rec.update(_state(model, 'w_after.'))
    rec.update(w0)
    return rec


def g3_steps(asr, out):
    for name, cfg in TINY.items():
        _seed(31)
        gen = torch.Generator().manual_seed(9)
        V = 9
        if 'VGG' in cfg['encoder']['enc_type']:
            x, y, lens = _synth_batch(3, 38, 26, V, 3, gen)        # D=26 = 2 x 13 MFCC-style; T'=9 after the 4x reduction
        else:
            x, y, lens = _synth_batch(4, 17, 5, V, 4, gen)
        rec = ref_step(asr, cfg, x, y, V, n_opt_steps=3)
        rec.update({'x': _np(x), 'y': _np(y), 'lens': np.array(lens), 'V': np.array(V)})
        np.savez(os.path.join(out, f'g3_step_{name}.npz'), **rec)


def g9_sched_sampling(asr, out):
    """Scheduled sampling (asr.py:95-100) at tf_rate = 0.5, one optimiser-free step per config: the reference's outputs and
    gradients together with the coin flips and the sampled tokens it drew, so that a replay (the same flips, the same
    tokens fed back) must reproduce them.  The seed is chosen so that both branches of the flip occur."""
    for name in ('dot_att', 'loc_ctc'):
        cfg = TINY[name]
        for seed in range(31, 60):
            _seed(seed)
            gen = torch.Generator().manual_seed(9)
            V = 9
            x, y, lens = _synth_batch(4, 29, 5, V, 6, gen)
            record = dict(flip_values=[], draws=[])
            rec = ref_step(asr, cfg, x, y, V, n_opt_steps=1, tf_rate=0.5, record=record)
            L = len(record['flip_values'])
            flips = [v <= 0.5 for v in record['flip_values']]
            n_s = sum(not f for f in flips[:L - 1])
            if n_s >= 2 and (L - 1 - n_s) >= 2 and np.isfinite(float(rec['loss'])):      # both branches occur before the last step
                break
        rec = {k: v for k, v in rec.items() if not k.startswith('w_after.') and not k.startswith('loss_it')}
        rec.update({'x': _np(x), 'y': _np(y), 'lens': np.array(lens), 'V': np.array(V), 'seed': np.array(seed),
                    'flip_values': np.array(record['flip_values'], np.float64),
                    'draw_step': np.array([t for t, _ in record['draws']], np.int64),
                    'draw_tokens': np.stack([_np(tok) for _, tok in record['draws']]).astype(np.int64)})
        np.savez(os.path.join(out, f'g9_sched_{name}.npz'), **rec)


# ----------------------------------------------------------------------------- G4 trainer trace
def g4_trace(Writer, out):
    """4-step Trainer.exec() on seeded synthetic TIMIT-format pickles (dataset.py:23-52)."""
    import yaml
    from src.solver import Trainer
    tmp = tempfile.mkdtemp(prefix='g4_')
    rng = np.random.RandomState(41)
    V, D = 12, 6
    mapping = {'<sos>': 0, '<eos>': 1}
    for i in range(2, V):
        mapping['p%d#' % i] = i          # '#' -> unit 'phone' would need collapse map; avoid: use char unit
    mapping = {'<sos>': 0, '<eos>': 1}
    for i in range(2, V):
        mapping[chr(ord('a') + i)] = i    # <50 symbols, no '#', no '▁' -> unit 'char'
    with open(os.path.join(tmp, 'mapping.pkl'), 'wb') as f:
        pickle.dump(mapping, f)

    def mk(n):
        xs, ys = [], []
        for _ in range(n):
            T = rng.randint(12, 25)
            xs.append(rng.randn(T, D).astype(np.float32))
            L = rng.randint(2, 5)
            ys.append([0] + list(rng.randint(2, V, size=L)) + [1])
        return xs, ys
    data = {}
    for split, n in [('train', 12), ('test', 4)]:
        xs, ys = mk(n)
        data[split] = (xs, ys)
        with open(os.path.join(tmp, f'{split}_x.pkl'), 'wb') as f:
            pickle.dump(xs, f)
        with open(os.path.join(tmp, f'{split}_y.pkl'), 'wb') as f:
            pickle.dump(ys, f)

    cfg = yaml.safe_load(open(os.path.join(REF, 'config/timit_example.yaml')))
    cfg['asr_model']['encoder'].update(dim='8_8_8')
    cfg['asr_model']['attention'].update(dim=8)
    cfg['asr_model']['decoder'].update(dim=8)
    cfg['asr_model']['optimizer'].update(learning_rate=0.001)
    cfg['solver'].update(data_path=tmp, n_jobs=0, batch_size=4, dev_batch_size=4, apex=False,
                         total_steps=3, tf_start=1.0, tf_end=1.0, dev_step=1000)
    paras = argparse.Namespace(gpu=False, name='g4', config='config/g4.yaml', seed=0,
                               ckpdir=os.path.join(tmp, 'ckpt'), logdir=os.path.join(tmp, 'log'),
                               load=None, verbose=False, njobs=1)
    _seed(0)
    Writer.trace.clear()
    t = Trainer(cfg, paras)
    t.load_data()
    t.set_model()
    w0 = _state(t.asr_model, 'w.')
    t.exec()
    # flatten the scalar trace: rows of (step, name, key, value)
    rows = [(s, n, k, v) for (s, n, d) in Writer.trace for k, v in sorted(d.items())]
    rec = dict(w0)
    rec['trace_step'] = np.array([r[0] for r in rows])
    rec['trace_name'] = np.array([r[1] + '/' + r[2] for r in rows])
    rec['trace_val'] = np.array([r[3] for r in rows], dtype=np.float64)
    for split in data:
        xs, ys = data[split]
        rec[f'{split}_xlen'] = np.array([len(v) for v in xs])
        rec[f'{split}_x'] = np.concatenate(xs, 0)
        rec[f'{split}_ylen'] = np.array([len(v) for v in ys])
        rec[f'{split}_y'] = np.concatenate([np.array(v) for v in ys])
    rec['V'] = np.array(V)
    rec.update(_state(t.asr_model, 'w_after.'))
    np.savez(os.path.join(out, 'g4_trainer_trace.npz'), **rec)
    import json
    json.dump(cfg, open(os.path.join(out, 'g4_config.json'), 'w'), indent=1)


# ----------------------------------------------------------------------------- G5 prefix scorer (next-row N3)
def g5_prefix(out):
    from src.ctc import CTCPrefixScore
    gen = torch.Generator().manual_seed(51)
    Tp, V = 9, 6
    lp = torch.log_softmax(torch.randn(1, Tp, V, generator=gen), -1)
    sc = CTCPrefixScore(lp)
    r0 = sc.init_state()
    cand = [1, 2, 3, 4, 5]
    psi1, r1 = sc.cheap_compute([], r0, cand)
    g = [3]
    psi2, r2 = sc.cheap_compute(g, r1[cand.index(3)], cand)      # includes the repeat-token quirk (3 in cand)
    g = [3, 3]
    psi3, r3 = sc.cheap_compute(g, r2[cand.index(3)], [2, 3, 5])
    np.savez(os.path.join(out, 'g5_ctc_prefix.npz'), lp=_np(lp), r0=r0, psi1=psi1, r1=r1, psi2=psi2, r2=r2,
             psi3=psi3, r3=r3)


# ----------------------------------------------------------------------------- G6 beam search
def g6_beam(asr, out):
    """Seq2Seq.beam_decode (asr.py:155-258) on one utterance: joint CTC/attention (loc, 2 decoder layers) and
    attention-only (dot), beam 1 and 3; every returned hypothesis with its token ids and per-token scores."""
    for name in ['loc_ctc', 'dot_att']:
        cfg = TINY[name]
        for beam in [1, 3]:
            _seed(61)
            gen = torch.Generator().manual_seed(17)
            V, T, D, steps = 9, 23, 5, 7
            x = _ragged_x(1, T, D, [T], gen)
            model = asr.Seq2Seq(x, V, cfg)
            with torch.no_grad():                       # sharpen the random-init output layer: distinct beams
                model.char_trans.weight.mul_(6.0)
                if cfg['optimizer']['joint_ctc'] > 0:
                    model.ctc_layer.weight.mul_(4.0)
            model.eval()
            model.decode_lm_weight = 0
            with torch.no_grad():
                hyps = model.beam_decode(x, steps, [T], beam)
            d = {'x': _np(x), 'V': np.array(V), 'steps': np.array(steps), 'beam': np.array(beam), 'n_hyps': np.array(len(hyps))}
            for i, h in enumerate(hyps):
                d[f'hyp{i}.seq'] = np.array(h.outIndex, dtype=np.int64)
                d[f'hyp{i}.scores'] = np.array([float(v) for v in h.output_scores], dtype=np.float64)
            d.update(_state(model, 'w.'))
            np.savez(os.path.join(out, f'g6_beam_{name}_b{beam}.npz'), **d)



# ----------------------------------------------------------------------------- G7 validation (next-row N2)
def _timit_dir(rng, V, D, sizes, Tlo=12, Thi=25):
    tmp = tempfile.mkdtemp(prefix='g7_')
    mapping = {'<sos>': 0, '<eos>': 1}
    for i in range(2, V):
        mapping[chr(ord('a') + i)] = i    # <50 symbols, no '#', no '▁' -> unit 'char'
    with open(os.path.join(tmp, 'mapping.pkl'), 'wb') as f:
        pickle.dump(mapping, f)
    data = {}
    for split, n in sizes:
        xs, ys = [], []
        for _ in range(n):
            T = rng.randint(Tlo, Thi)
            xs.append(rng.randn(T, D).astype(np.float32))
            L = rng.randint(2, 5)
            ys.append([0] + list(rng.randint(2, V, size=L)) + [1])
        data[split] = (xs, ys)
        with open(os.path.join(tmp, f'{split}_x.pkl'), 'wb') as f:
            pickle.dump(xs, f)
        with open(os.path.join(tmp, f'{split}_y.pkl'), 'wb') as f:
            pickle.dump(ys, f)
    return tmp, data


_G7_CALLS = []


def _g7_hook(m, i, o):
    _G7_CALLS.append(o)


def g7_valid(Writer, out):
    """Trainer.valid() (solver.py:211-291) on a 6-utterance dev set (two buckets: 4 + 2), initial weights: greedy
    no-teacher decoding for ans_len + 30 steps (asr.py:101-102), dev_att / dev_ctc / dev_full, error rate, acc, the
    att_/hyp_/txt_ log entries of the last bucket, best_hyp.txt.  Two models: dot attention-only, and loc + CTC 0.5 with a
    2-layer Speller whose dropout is 0.3 (must be OFF in valid(): the reference brackets it with eval()/train())."""
    import yaml
    import json
    from src.solver import Trainer
    def attempt(name, seed):
        rng = np.random.RandomState(71)
        V, D = 12, 6
        tmp, data = _timit_dir(rng, V, D, [('train', 8), ('test', 6)], *((32, 48) if name == 'loc_ctc_drop' else (12, 25)))
        cfg = yaml.safe_load(open(os.path.join(REF, 'config/timit_example.yaml')))
        cfg['asr_model']['encoder'].update(dim='8_8_8')
        cfg['asr_model']['attention'].update(dim=8)
        cfg['asr_model']['decoder'].update(dim=8)
        cfg['asr_model']['optimizer'].update(learning_rate=0.001)
        if name == 'loc_ctc_drop':
            cfg['asr_model']['optimizer'].update(joint_ctc=0.5)
            cfg['asr_model']['attention'].update(att_mode='loc')
            cfg['asr_model']['decoder'].update(layer=2, dropout=0.3)
        cfg['solver'].update(data_path=tmp, n_jobs=0, batch_size=4, dev_batch_size=4, apex=False,
                             total_steps=3, tf_start=1.0, tf_end=1.0, dev_step=1000)
        paras = argparse.Namespace(gpu=False, name='g7', config='config/g7.yaml', seed=0,
                                   ckpdir=os.path.join(tmp, 'ckpt'), logdir=os.path.join(tmp, 'log'),
                                   load=None, verbose=False, njobs=1)
        _seed(seed)
        Writer.trace.clear(); Writer.texts.clear(); Writer.images.clear()
        t = Trainer(cfg, paras)
        t.load_data()
        t.set_model()
        with torch.no_grad():                       # sharpen the random-init output layer: argmax margins >> 1e-5
            t.asr_model.char_trans.weight.mul_(8.0)
        w0 = _state(t.asr_model, 'w.')
        calls = _G7_CALLS
        calls.clear()
        t.asr_model.register_forward_hook(_g7_hook)      # (module-level: valid() pickles the whole module, solver.py:283)
        t.valid()
        assert t.asr_model.training                 # valid() restores train mode (solver.py:287)
        rec = dict(w0)
        rows = [(s_, n, k, v) for (s_, n, d) in Writer.trace for k, v in sorted(d.items())]
        rec['trace_name'] = np.array([r[1] + '/' + r[2] for r in rows])
        rec['trace_val'] = np.array([r[3] for r in rows], dtype=np.float64)
        rec['text_name'] = np.array([n for _, n, _ in Writer.texts])
        rec['text_val'] = np.array([v for _, _, v in Writer.texts])
        rec['image_name'] = np.array([n for _, n, _ in Writer.images])
        for i, (_, n, img) in enumerate(Writer.images):
            rec[f'image{i}'] = img.astype(np.float32)
        margin = 1e9
        for i, (ctc_pred, enc_len, att_pred, att_maps) in enumerate(calls):
            rec[f'call{i}.att_pred'] = _np(att_pred)
            rec[f'call{i}.att_map'] = _np(att_maps[0])
            rec[f'call{i}.enc_len'] = np.array(enc_len)
            if ctc_pred is not None:
                rec[f'call{i}.ctc_pred'] = _np(ctc_pred)
            top2 = torch.topk(att_pred, 2, dim=-1).values
            margin = min(margin, float((top2[..., 0] - top2[..., 1]).min()))
        rec['n_calls'] = np.array(len(calls))
        rec['argmax_margin'] = np.array(margin)
        if margin <= 1e-3 or not all(np.isfinite(r[3]) for r in rows):
            return None
        rec['model_seed'] = np.array(seed)
        rec['best_hyp'] = np.array(open(os.path.join(tmp, 'ckpt', 'g7', 'best_hyp.txt')).read())
        for split in data:
            xs, ys = data[split]
            rec[f'{split}_xlen'] = np.array([len(v) for v in xs])
            rec[f'{split}_x'] = np.concatenate(xs, 0)
            rec[f'{split}_ylen'] = np.array([len(v) for v in ys])
            rec[f'{split}_y'] = np.concatenate([np.array(v) for v in ys])
        rec['V'] = np.array(V)
        np.savez(os.path.join(out, f'g7_valid_{name}.npz'), **rec)
        json.dump(cfg, open(os.path.join(out, f'g7_config_{name}.json'), 'w'), indent=1)
        return rec

    for name in ['dot_att', 'loc_ctc_drop']:
        # (first model seed whose greedy argmax margins all exceed 1e-3, so that the token sequence is not decided by rounding)
        assert any(attempt(name, seed) is not None for seed in range(7, 40)), name


# ----------------------------------------------------------------------------- G8 LibriSpeech loader (next-row N4)
def g8_libri(out):
    """LoadDataset for the csv + per-utterance .npy format (dataset.py:57-155): bucket membership and order for
    train / dev / test, incl. a bucket that triggers the half-batch rule by length (> 800 frames), one by label length
    (> 150), the drop filters (length >= max_timestep, label length >= max_label_len), a trailing partial bucket, ties in
    length, and the padded tensors of two buckets.  The csv rows are stored so the test rebuilds the directory."""
    from src.dataset import LoadDataset
    rng = np.random.RandomState(81)
    tmp = tempfile.mkdtemp(prefix='g8_')
    D = 4
    rows = {}
    spec = {'train': [900, 850, 820, 805, 640, 640, 640, 500, 410, 400, 399, 380, 300, 300, 250, 1300, 1250, 120, 90, 64, 33, 30],
            'dev': [700, 650, 610, 90, 80, 70, 60],
            'test': [55, 801, 47]}
    for split, lens in spec.items():
        rr = []
        os.makedirs(os.path.join(tmp, split), exist_ok=True)
        order = rng.permutation(len(lens))
        for k in order:
            n = lens[k]
            fp = f'{split}/utt{k:02d}.npy'
            np.save(os.path.join(tmp, fp), rng.randn(n, D).astype(np.float32))
            L = int(rng.randint(3, 9))
            if split == 'train' and n == 410:
                L = 160                     # half-batch by label length
            if split == 'train' and n == 33:
                L = 420                     # dropped: label length >= max_label_len
            lab = [0] + list(rng.randint(2, 30, size=L)) + [1]
            rr.append((fp, n, '_'.join(str(v) for v in lab)))
        rows[split] = rr
        with open(os.path.join(tmp, split + '.csv'), 'w') as f:
            f.write('file_path,length,label\n')
            for r in rr:
                f.write('%s,%d,%s\n' % r)
    solver = dict(data_path=tmp, batch_size=4, max_timestep=1200, max_label_len=400, use_gpu=False, n_jobs=0,
                  dataset='librispeech', train_set=['train'], dev_set=['dev'], test_set=['test'], dev_batch_size=4,
                  decode_beam_size=1, dev_step=10)
    rec = {}
    for split in ['train', 'dev', 'test']:
        dl = LoadDataset(split, text_only=False, **solver)
        ds = dl.dataset
        rec[f'{split}.n_buckets'] = np.array(len(ds))
        for i in range(len(ds)):
            rec[f'{split}.bucket{i}.files'] = np.array(ds.X[i])
            x, y = ds[i]
            rec[f'{split}.bucket{i}.xshape'] = np.array(x.shape)
            rec[f'{split}.bucket{i}.y'] = np.asarray(y)
            if split != 'train' or i in (0, 3):
                rec[f'{split}.bucket{i}.xsum'] = _np(x.sum(-1))
        rec[f'{split}.csv'] = np.array(['%s,%d,%s' % r for r in rows[split]])
    # one more loader with decode_beam_size > 1: test buckets of a single utterance (dataset.py:136)
    solver['decode_beam_size'] = 5
    ds = LoadDataset('test', text_only=False, **solver).dataset
    rec['test_beam.files'] = np.array([v[0] for v in ds.X])
    rec['D'] = np.array(D)
    np.savez(os.path.join(out, 'g8_libri_buckets.npz'), **rec)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--out', default=os.path.join(os.path.dirname(__file__), '..', 'tests', 'golden'))
    ap.add_argument('--only', default='', help='comma list of groups to (re)generate, e.g. g6; default all')
    a = ap.parse_args()
    out = os.path.abspath(a.out)
    os.makedirs(out, exist_ok=True)
    torch.set_num_threads(1)            # bit-stable reductions
    Writer = _install_shims()
    sys.path.insert(0, REF)
    import src.asr as asr
    only = set(v for v in a.only.split(',') if v)
    want = lambda g: not only or g in only
    if want('g1'):
        g1_rnnlayer(asr, out)
        g1_listener(asr, out)
        g1_attention(asr, out)
        g1_vgg(asr, out)
        g1_speller(asr, out)
    if want('g2'):
        g2_ctc(out)
    if want('g3'):
        g3_steps(asr, out)
    if want('g4'):
        g4_trace(Writer, out)
    if want('g5'):
        g5_prefix(out)
    if want('g6'):
        g6_beam(asr, out)
    if want('g7'):
        g7_valid(Writer, out)
    if want('g8'):
        g8_libri(out)
    if want('g9'):
        g9_sched_sampling(asr, out)
    tot = sum(os.path.getsize(os.path.join(out, f)) for f in os.listdir(out))
    print('wrote', len(os.listdir(out)), 'files,', tot // 1024, 'KiB ->', out)


if __name__ == '__main__':
    main()
