"""GPU, one process: the data-parallel recipe on the REAL kernels (VERDICT r1 item 8 / SURVEY.md 8e normalisation caveat).
The gradient of two round-robin shards, each computed through the HIP path and weighted B_local * world / B_global as
Trainer.train_step does, summed and scaled by 1/world (what the sum all-reduce + the clip kernel's gscale do), equals the
full-batch HIP gradient -- for equal shards (4 -> 2 + 2) and unequal ones (3 -> 2 + 1).  f32 mode, atol 2e-6 + 1e-4 rel."""
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


def grad_of(ops, model, x, y, w, weight=1.0):
    model.flat_grads.zero_()
    lens = ops.infer_lengths(x)
    ntok = ops.count_nonzero(y)
    L = int(ntok.max().item())
    ctc_pred, _, att_pred, _ = model(x, L, tf_rate=1.0, teacher=y, state_len=lens.cpu().tolist())
    loss, _, _ = ops.joint_loss(att_pred, ctc_pred, y, ntok, model.last_enc_len_dev, L, w)
    (loss * weight if weight != 1.0 else loss).backward()
    ops.join_side_stream()
    torch.cuda.synchronize()
    return model.flat_grads.clone(), float(loss.detach())


@pytest.mark.parametrize('name', ['loc_ctc', 'dot_att'])
@pytest.mark.parametrize('B', [4, 3])
def test_sharded_gradients_sum_to_full_batch(name, B):
    importlib.import_module('end-to-end-asr-pytorch_amd')
    ops = importlib.import_module('end-to-end-asr-pytorch_amd.ops')
    asr = importlib.import_module('end-to-end-asr-pytorch_amd.asr')
    ldist = importlib.import_module('end-to-end-asr-pytorch_amd.dist')
    from gen_golden import TINY
    d = np.load(os.path.join(GOLDEN, f'g3_step_{name}.npz'))
    cfg = TINY[name]
    x = torch.tensor(d['x'][:B], device=DEV)
    y = torch.tensor(d['y'][:B], device=DEV)
    w = cfg['optimizer']['joint_ctc']
    world = 2
    ops.set_precision('f32')
    try:
        model = asr.Seq2Seq(x, int(d['V']), cfg, device=DEV)
        model.load_reference_state({k[2:]: d[k] for k in d.files if k.startswith('w.')})
        full, loss_full = grad_of(ops, model, x, y, w)
        acc = torch.zeros_like(full)
        loss_acc = 0.0
        lens = ops.infer_lengths(x).cpu().tolist()
        for r in range(world):
            xs, ys, ls = ldist.shard_bucket(x, y, lens, r, world)
            assert ls == sorted(ls, reverse=True)
            wr = xs.shape[0] * world / B
            g, l = grad_of(ops, model, xs.contiguous(), ys.contiguous(), w, weight=wr)
            acc += g                                      # the sum all-reduce
            loss_acc += l * wr
        acc /= world                                      # las_grad_norm's gscale = 1/world
    finally:
        ops.set_precision('bf16')
    assert abs(loss_acc / world - loss_full) < 2e-5 * max(1.0, abs(loss_full))
    np.testing.assert_allclose(acc.cpu().numpy(), full.cpu().numpy(), atol=2e-6, rtol=1e-4)
