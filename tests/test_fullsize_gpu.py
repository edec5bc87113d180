"""GPU: size-independent properties of ONE whole train-step forward + backward at the FULL sizes bench.py times (SURVEY.md
8d: B=24, T_max=1200, L_max=200 / 60), so that the long-sequence code paths (T=1200 recurrences, ~200 decode steps, the
persistent decoder loops, V=5000 CTC) are asserted and not just timed.  No oracle at these sizes (it would take minutes):
  * the loss and every gradient are finite; at random init the attention CE is ~ln V;
  * every attention row sums to 1 inside enc_len and is exactly 0 beyond it (asr.py:429-431 mask + softmax);
  * d ctc / d logits sums to 0 over the vocabulary for every frame inside enc_len (log-softmax Jacobian) and is exactly 0
    beyond it (ATen's ctc_loss_backward contract);
  * nothing flows into padding: d loss / d x is exactly 0 for frames >= the utterance's length (packed-sequence semantics,
    asr.py:480-483) and non-zero inside;
  * the persistent kernels report no hand-off timeout."""
import importlib
import math
import os
import sys
import numpy as np
import pytest
import torch

from conftest import ROOT

pytestmark = pytest.mark.gpu
DEV = 'cuda:0'
sys.path.insert(0, ROOT)


@pytest.mark.parametrize('workload', ['c1', 'c2', 'c3', 'c4', 'c5'])
def test_full_size_step_properties(workload):
    import bench
    ops = importlib.import_module('end-to-end-asr-pytorch_amd.ops')
    asr = importlib.import_module('end-to-end-asr-pytorch_amd.asr')
    synth = importlib.import_module('end-to-end-asr-pytorch_amd.synth')
    w = bench.WORKLOADS[workload]
    cfg = bench.model_cfg(w)
    tr = bench.time_reduction(w)
    x, y, lens = synth.make_batch(3, w['B'], w['T_max'], w['D'], w['V'], w['L_max'], tr, ctc=w['ctc'] > 0)
    torch.manual_seed(1)
    ops.set_precision(w['prec'])
    try:
        model = asr.Seq2Seq(x, w['V'], cfg, device=DEV)
        xd = x.to(DEV).requires_grad_(True)
        yd = y.to(DEV)
        ntok = ops.count_nonzero(yd)
        L = int(ntok.max().item())
        ctc_pred, enc_len, att_pred, att_maps = model(xd, L, tf_rate=1.0, teacher=yd, state_len=lens)
        if ctc_pred is not None:
            ctc_pred.retain_grad()
        loss, att_loss, ctc_loss = ops.joint_loss(att_pred, ctc_pred, yd, ntok, model.last_enc_len_dev, L, w['ctc'])
        model.flat_grads.zero_()
        loss.backward()
        ops.join_side_stream()
        torch.cuda.synchronize()
    finally:
        ops.set_precision('bf16')
    assert int(model.status.item()) == 0
    assert enc_len == [l // tr for l in lens]
    assert math.isfinite(float(loss.detach())) and bool(torch.isfinite(model.flat_grads).all())
    assert float(model.flat_grads.abs().max()) > 0
    if w['ctc'] < 1:
        assert 0.5 * math.log(w['V']) < float(att_loss) < 2.5 * math.log(w['V'])
        att = att_maps[0]                                         # (B, L, T')
        Tp = att.shape[-1]
        mask = torch.arange(Tp, device=DEV)[None, :] < torch.tensor(enc_len, device=DEV)[:, None]       # (B, T')
        assert float((att.sum(-1) - 1).abs().max()) < 1e-4
        assert float((att * (~mask)[:, None, :]).abs().max()) == 0.0
        assert float(att.min()) >= 0.0
    if w['ctc'] > 0:
        g = ctc_pred.grad                                          # (B, T', V)
        Tp = g.shape[1]
        mask = torch.arange(Tp, device=DEV)[None, :] < torch.tensor(enc_len, device=DEV)[:, None]
        scale = float(g.abs().max())
        assert scale > 0 and bool(torch.isfinite(g).all())
        assert float(g.sum(-1).abs().max()) <= 2e-4 * max(scale, 1e-6) * math.sqrt(w['V'])
        assert float((g * (~mask)[..., None]).abs().max()) == 0.0
    gx = xd.grad                                                   # (B, T, D)
    T = gx.shape[1]
    inside = torch.arange(T, device=DEV)[None, :] < torch.tensor(lens, device=DEV)[:, None]
    assert float((gx * (~inside)[..., None]).abs().max()) == 0.0
    assert float(((gx.abs().sum(-1) > 0) & inside).sum()) > 0.99 * float(inside.sum())
