"""GPU parity of the WHOLE train-step arithmetic (Seq2Seq.forward + joint loss + backward) against golden
vectors produced by the imported reference (tests/golden/g3_step_*.npz): logits, attention maps, CTC logits,
losses and every parameter gradient.  f32 mode: atol 5e-5 / rtol 1e-3; bf16 mode: loss rel 2e-2, grads atol 3e-2
relative to the largest gradient entry (the stated tolerance of the bf16 MFMA path)."""
import importlib
import os
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
    return (importlib.import_module('end-to-end-asr-pytorch_amd.ops'),
            importlib.import_module('end-to-end-asr-pytorch_amd.asr'))


@pytest.mark.parametrize('prec', ['f32', 'bf16'])
@pytest.mark.parametrize('name', ['dot_att', 'loc_ctc', 'ctc_only', 'vgg_loc_ctc'])
def test_step_vs_reference(las, name, prec):
    ops, asr = las
    from gen_golden import TINY
    d = np.load(os.path.join(GOLDEN, f'g3_step_{name}.npz'))
    cfg = TINY[name]
    x = torch.tensor(d['x'], device=DEV)
    y = torch.tensor(d['y'], device=DEV)
    V = int(d['V'])
    ops.set_precision(prec)
    try:
        model = asr.Seq2Seq(x, V, cfg, device=DEV)
        model.load_reference_state({k[2:]: d[k] for k in d.files if k.startswith('w.')})
        lens = ops.infer_lengths(x)
        assert lens.cpu().tolist() == list(d['lens'])
        ntok = ops.count_nonzero(y)
        ans_len = int(ntok.max().item())
        ctc_pred, enc_len, att_pred, att_maps = model(x, ans_len, tf_rate=1.0, teacher=y, state_len=lens.cpu().tolist())
        w = cfg['optimizer']['joint_ctc']
        loss, att_loss, ctc_loss = ops.joint_loss(att_pred, ctc_pred, y, ntok, model.last_enc_len_dev, ans_len, w)
        model.flat_grads.zero_()
        loss.backward()
        torch.cuda.synchronize()
    finally:
        ops.set_precision('bf16')
    assert int(model.status.item()) == 0
    assert enc_len == list(d['enc_len'])
    f32 = prec == 'f32'
    tol = dict(atol=5e-5, rtol=1e-3) if f32 else dict(atol=3e-2, rtol=3e-2)
    if 'att_pred' in d.files:
        np.testing.assert_allclose(att_pred.detach().cpu().numpy(), d['att_pred'], **tol)
        np.testing.assert_allclose(att_maps[0].cpu().numpy(), d['att_map'], **tol)
    if 'ctc_pred' in d.files:
        np.testing.assert_allclose(ctc_pred.detach().cpu().numpy(), d['ctc_pred'], **tol)
    lt = 2e-5 if f32 else 2e-2
    assert abs(float(loss.detach()) - float(d['loss'])) <= lt * max(1.0, abs(float(d['loss'])))
    assert abs(float(att_loss) - float(d['att_loss'])) <= lt * max(1.0, abs(float(d['att_loss'])))
    assert abs(float(ctc_loss) - float(d['ctc_loss'])) <= lt * max(1.0, abs(float(d['ctc_loss'])))
    gmax = max(np.abs(d[k]).max() for k in d.files if k.startswith('grad.'))
    bad = []
    for n, p in model.named_parameters():
        ref = d['grad.' + n]
        got = p.grad.detach().cpu().numpy()
        err = np.abs(got - ref).max()
        lim = (2e-5 + 1e-3 * np.abs(ref).max()) if f32 else 3e-2 * gmax
        if not err <= lim:
            bad.append((n, float(err), float(lim)))
    assert not bad, bad
    gn = float(torch.sqrt((model.flat_grads.double() ** 2).sum()))
    assert abs(gn - float(d['grad_norm'])) <= (1e-4 if f32 else 3e-2) * max(1.0, float(d['grad_norm']))


def test_seq2seq_dropout_modes(las):
    """decoder.dropout > 0 (asr.py:327): active in train mode (two passes differ), off in eval mode (equals the
    dropout-0 model); encoder `dropout` is accepted and inert, as nn.LSTM(num_layers=1, dropout=p) is in the reference."""
    ops, asr = las
    from gen_golden import TINY
    import copy
    cfg = copy.deepcopy(TINY['loc_ctc'])
    cfg['decoder']['dropout'] = 0.25
    cfg['encoder']['dropout'] = '0.2_0.2'
    d = np.load(os.path.join(GOLDEN, 'g3_step_loc_ctc.npz'))
    x = torch.tensor(d['x'], device=DEV); y = torch.tensor(d['y'], device=DEV)
    ops.set_precision('f32')
    try:
        m = asr.Seq2Seq(x, int(d['V']), cfg, device=DEV)
        m.load_reference_state({k[2:]: d[k] for k in d.files if k.startswith('w.')})
        L = int((y != 0).sum(-1).max())
        m.train()
        a1 = m(x, L, tf_rate=1.0, teacher=y)[2].detach().cpu().numpy()
        a2 = m(x, L, tf_rate=1.0, teacher=y)[2].detach().cpu().numpy()
        m.eval()
        a3 = m(x, L, tf_rate=1.0, teacher=y)[2].detach().cpu().numpy()
        torch.cuda.synchronize()
    finally:
        ops.set_precision('bf16')
    assert np.abs(a1 - a2).max() > 1e-3
    np.testing.assert_allclose(a3, d['att_pred'], atol=5e-5, rtol=1e-3)
