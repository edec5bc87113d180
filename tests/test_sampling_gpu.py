"""GPU: scheduled sampling (reference src/asr.py:95-100), the branch the shipped TIMIT config trains with (tf_start 0.9 ->
tf_end 0.7).
  * replay of the reference's own run at tf_rate 0.5 (tests/golden/g9_sched_*.npz: its coin flips and its sampled tokens):
    logits, attention maps, losses and every gradient, f32 mode at the whole-step tolerances of test_step_gpu.py;
  * the device sampler's path (step_mode 0: per-step kernels, logits of the previous step on the device, Gumbel-max draw,
    embedding of the drawn token) against a teacher-forced replay of the tokens it drew: the same numbers;
  * the draw itself: empirical frequencies of las_sample_rows over >= 20 000 draws against softmax(logits), chi-square."""
import ctypes
import importlib
import os
import random
import sys
import numpy as np
import pytest
import torch

from conftest import GOLDEN, ROOT

pytestmark = pytest.mark.gpu
DEV = 'cuda:0'
sys.path.insert(0, os.path.join(ROOT, 'tools'))


@pytest.fixture(scope='module')
def las():
    importlib.import_module('end-to-end-asr-pytorch_amd')
    return (importlib.import_module('end-to-end-asr-pytorch_amd.ops'), importlib.import_module('end-to-end-asr-pytorch_amd.asr'),
            importlib.import_module('end-to-end-asr-pytorch_amd.decoder'))


def run_step(ops, model, x, y, w, **fw):
    lens = ops.infer_lengths(x)
    ntok = ops.count_nonzero(y)
    L = int(ntok.max().item())
    ctc_pred, enc_len, att_pred, att_maps = model(x, L, teacher=y, state_len=lens.cpu().tolist(), **fw)
    loss, att_loss, ctc_loss = ops.joint_loss(att_pred, ctc_pred, y, ntok, model.last_enc_len_dev, L, w)
    model.flat_grads.zero_()
    loss.backward()
    ops.join_side_stream()
    torch.cuda.synchronize()
    assert int(model.status.item()) == 0
    return dict(att_pred=att_pred.detach().cpu().numpy(), att_map=att_maps[0].cpu().numpy(), loss=float(loss.detach()),
                grads={n: p.grad.detach().cpu().numpy().copy() for n, p in model.named_parameters()})


@pytest.mark.parametrize('prec', ['f32', 'bf16'])
@pytest.mark.parametrize('name', ['dot_att', 'loc_ctc'])
def test_scheduled_sampling_replays_reference(las, name, prec):
    ops, asr, _ = las
    from gen_golden import TINY
    d = np.load(os.path.join(GOLDEN, f'g9_sched_{name}.npz'))
    cfg = TINY[name]
    x, y = torch.tensor(d['x'], device=DEV), torch.tensor(d['y'], device=DEV)
    flips = [bool(v <= 0.5) for v in d['flip_values']]
    tokens = {int(t): torch.tensor(tok) for t, tok in zip(d['draw_step'], d['draw_tokens'])}
    ops.set_precision(prec)
    try:
        model = asr.Seq2Seq(x, int(d['V']), cfg, device=DEV)
        model.load_reference_state({k[2:]: d[k] for k in d.files if k.startswith('w.')})
        out = run_step(ops, model, x, y, cfg['optimizer']['joint_ctc'], tf_rate=0.5, replay=dict(flips=flips, tokens=tokens))
    finally:
        ops.set_precision('bf16')
    f32 = prec == 'f32'
    tol = dict(atol=5e-5, rtol=1e-3) if f32 else dict(atol=3e-2, rtol=3e-2)
    np.testing.assert_allclose(out['att_pred'], d['att_pred'], **tol)
    np.testing.assert_allclose(out['att_map'], d['att_map'], **tol)
    assert abs(out['loss'] - float(d['loss'])) <= (2e-5 if f32 else 2e-2) * max(1.0, abs(float(d['loss'])))
    gmax = max(np.abs(d[k]).max() for k in d.files if k.startswith('grad.'))
    coef = min(1.0, 5.0 / (float(d['grad_norm']) + 1e-6))            # the golden gradients are the clipped ones
    bad = []
    for n, g in out['grads'].items():
        ref = d['grad.' + n]
        err = np.abs(g * coef - ref).max()
        lim = (2e-5 + 1e-3 * np.abs(ref).max()) if f32 else 3e-2 * gmax
        if not err <= lim:
            bad.append((n, float(err), float(lim)))
    assert not bad, bad


@pytest.mark.parametrize('name', ['dot_att', 'loc_ctc'])
def test_sampled_steps_equal_teacher_forced_replay(las, name, monkeypatch):
    """The live path: flips from random.random(), draws from the device sampler.  The tokens it fed (DecoderFn.last_tok) are
    then replayed as teacher input through the same model: the per-step kernels with a drawn token must compute what they
    compute with that token given -- logits and attention maps equal to 1e-6, gradients to 1e-5 of the largest entry (float
    atomics in the sums).  Also: the draws differ between two seeds, and every drawn token is a valid class."""
    ops, asr, dec = las
    from gen_golden import TINY
    d = np.load(os.path.join(GOLDEN, f'g9_sched_{name}.npz'))
    cfg = TINY[name]
    x, y = torch.tensor(d['x'], device=DEV), torch.tensor(d['y'], device=DEV)
    V = int(d['V'])
    flips = [False, True, False, False, True, False]
    ops.set_precision('f32')
    try:
        model = asr.Seq2Seq(x, V, cfg, device=DEV)
        model.load_reference_state({k[2:]: d[k] for k in d.files if k.startswith('w.')})
        L0 = int((d['y'] != 0).sum(-1).max())
        vals, state = [0.1 if f else 0.9 for f in flips[:L0]], {'i': 0}      # every forward draws exactly L0 values: the same pattern each time

        def fake_random():
            state['i'] += 1
            return vals[(state['i'] - 1) % L0]
        monkeypatch.setattr(asr.random, 'random', fake_random)
        live = run_step(ops, model, x, y, cfg['optimizer']['joint_ctc'], tf_rate=0.5)
        fed = dec.DecoderFn.last_tok.cpu().numpy().copy()              # [L][B]
        L = fed.shape[0]
        assert ((fed >= 0) & (fed < V)).all()
        yh = d['y']
        for t in range(1, L):
            if flips[t - 1]:
                assert (fed[t] == yh[:, t]).all()                      # teacher-fed steps got the label
        tokens = {t: torch.tensor(fed[t + 1]) for t in range(L - 1) if not flips[t]}
        rep = run_step(ops, model, x, y, cfg['optimizer']['joint_ctc'], tf_rate=0.5, replay=dict(flips=flips[:L], tokens=tokens))
        live2 = run_step(ops, model, x, y, cfg['optimizer']['joint_ctc'], tf_rate=0.5)
        fed2 = dec.DecoderFn.last_tok.cpu().numpy()
    finally:
        ops.set_precision('bf16')
    np.testing.assert_allclose(rep['att_pred'], live['att_pred'], atol=1e-6, rtol=1e-6)
    np.testing.assert_allclose(rep['att_map'], live['att_map'], atol=1e-6, rtol=1e-6)
    assert abs(rep['loss'] - live['loss']) <= 1e-6 * max(1.0, abs(live['loss']))
    gmax = max(np.abs(g).max() for g in live['grads'].values())
    for n in live['grads']:
        assert np.abs(rep['grads'][n] - live['grads'][n]).max() <= 1e-5 * gmax, n
    sampled_rows = [t + 1 for t in range(L - 1) if not flips[t]]
    assert any((fed[t] != fed2[t]).any() for t in sampled_rows), 'two forward passes drew identical tokens: the seed does not advance'


@pytest.mark.parametrize('V,rows', [(31, 40000), (5000, 40000)])
def test_device_sampler_matches_softmax(las, V, rows):
    """las_sample_rows (the draw of asr.py:99 on the device: counter-hash Gumbel-max): `rows` independent draws from ONE logit
    vector; tokens are merged into bins of expected count >= 40 and the chi-square statistic must lie within 5 sigma of its
    mean (df +- 5 sqrt(2 df)); a second seed gives different draws with the same property; greedy = argmax."""
    ops, _, _ = las
    lib = importlib.import_module('end-to-end-asr-pytorch_amd._lib')
    L_ = lib.lib()
    rng = np.random.RandomState(V)
    logits = (2.5 * rng.randn(V)).astype(np.float32)
    p = np.exp(logits.astype(np.float64) - logits.max())
    p /= p.sum()
    X = torch.tensor(np.tile(logits, (rows, 1)), device=DEV)
    draws = []
    for seed in (12345, 999):
        tok = torch.empty(rows, dtype=torch.int32, device=DEV)
        lib.check(L_.las_sample_rows(lib.ptr(X), ctypes.c_int(rows), ctypes.c_int(V), ctypes.c_int(0), ctypes.c_uint(seed), lib.ptr(tok),
                                     lib.cur_stream()), 'las_sample_rows')
        torch.cuda.synchronize()
        t = tok.cpu().numpy()
        assert ((t >= 0) & (t < V)).all()
        draws.append(t)
        cnt = np.bincount(t, minlength=V).astype(np.float64)
        order = np.argsort(-p)
        bins_o, bins_e, o_acc, e_acc = [], [], 0.0, 0.0
        for i in order:
            o_acc += cnt[i]; e_acc += p[i] * rows
            if e_acc >= 40.0:
                bins_o.append(o_acc); bins_e.append(e_acc); o_acc = e_acc = 0.0
        if e_acc > 0:
            bins_o[-1] += o_acc; bins_e[-1] += e_acc
        bo, be = np.array(bins_o), np.array(bins_e)
        df = len(bo) - 1
        chi2 = float(((bo - be) ** 2 / be).sum())
        assert df >= 10 and abs(chi2 - df) <= 5.0 * np.sqrt(2.0 * df), (V, seed, chi2, df)
    assert (draws[0] != draws[1]).mean() > 0.3
    tok = torch.empty(rows, dtype=torch.int32, device=DEV)
    lib.check(L_.las_sample_rows(lib.ptr(X), ctypes.c_int(rows), ctypes.c_int(V), ctypes.c_int(1), ctypes.c_uint(1), lib.ptr(tok),
                                 lib.cur_stream()), 'las_sample_rows')
    assert (tok.cpu().numpy() == int(np.argmax(logits))).all()
