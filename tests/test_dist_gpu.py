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


# ------------------------------------------------------------------------------------------------------------------
# The RCCL branch of dist.backward_with_overlap, executed: a 1-rank `nccl` group on the one GPU of the box (RCCL refuses two
# ranks on one device, so this is as many ranks as a one-GPU box can give it) with LAS_DIST_FORCE_OVERLAP=1, one c3-shaped
# train step (bench.py's workload at B = 24, T = 1200).  A 1-rank all-reduce is the identity, so a bucket released too early
# would go unnoticed in the gradient itself; the exchange therefore also copies every bucket aside (`snapshot`) on the comm
# stream at the very point where its collective may start, and that copy must equal the final gradient BIT FOR BIT.
def _c3_step(ops, asr, ldist, model, x, y, snapshot=None):
    model.flat_grads.zero_()
    lens = ops.infer_lengths(x)
    ntok = ops.count_nonzero(y)
    L = int(ntok.max().item())
    model.ctc_branch = True
    try:
        ctc_pred, _, att_pred, _ = model(x, L, tf_rate=1.0, teacher=y, state_len=lens.cpu().tolist(), state_len_dev=lens)
    finally:
        model.ctc_branch = False
    loss, _, _ = ops.joint_loss(att_pred, ctc_pred, y, ntok, model.last_enc_len_dev, L, 0.5)
    ldist.backward_with_overlap(loss, model, snapshot=snapshot)
    torch.cuda.synchronize()
    return model.flat_grads.clone(), float(loss.detach())


def test_nccl_overlap_branch_executes_on_one_rank(monkeypatch):
    import socket
    import torch.distributed as dist
    import bench
    importlib.import_module('end-to-end-asr-pytorch_amd')
    ops = importlib.import_module('end-to-end-asr-pytorch_amd.ops')
    asr = importlib.import_module('end-to-end-asr-pytorch_amd.asr')
    synth = importlib.import_module('end-to-end-asr-pytorch_amd.synth')
    ldist = importlib.import_module('end-to-end-asr-pytorch_amd.dist')
    w = bench.WORKLOADS['c3']
    cfg = bench.model_cfg(w)
    x, y, _ = synth.make_batch(0, w['B'], w['T_max'], w['D'], w['V'], w['L_max'], bench.time_reduction(w), ctc=True)
    x, y = x.to(DEV), y.to(DEV)
    torch.manual_seed(0)
    model = asr.Seq2Seq(x, w['V'], cfg, device=DEV)
    with socket.socket() as s:
        s.bind(('127.0.0.1', 0))
        port = s.getsockname()[1]
    monkeypatch.setenv('MASTER_ADDR', '127.0.0.1')
    monkeypatch.setenv('MASTER_PORT', str(port))
    torch.cuda.set_device(0)
    dist.init_process_group(backend='nccl', world_size=1, rank=0)
    try:
        assert not ldist.overlap_active(model.flat_grads)                  # world 1: the plain path ...
        plain, loss_plain = _c3_step(ops, asr, ldist, model, x, y)
        again, _ = _c3_step(ops, asr, ldist, model, x, y)                  # (run-to-run noise of the split-K / colsum float atomics)
        noise = float((plain - again).abs().max())
        monkeypatch.setenv('LAS_DIST_FORCE_OVERLAP', '1')
        assert ldist.overlap_active(model.flat_grads)                      # ... forced onto the RCCL branch
        fired = []
        real_ready = ldist._Exchange.ready
        monkeypatch.setattr(ldist._Exchange, 'ready', lambda self, lo: (fired.append(int(lo)), real_ready(self, lo))[1])
        snap = torch.full_like(model.flat_grads, float('nan'))
        over, loss_over = _c3_step(ops, asr, ldist, model, x, y, snapshot=snap)
        assert int(model.status.item()) == 0
        assert ldist.STATS['collectives'] >= 2, ldist.STATS                # per-layer buckets went out during backward
        assert fired == model.grad_ready_offsets() + [0], (fired, model.grad_ready_offsets())   # what an empty rank replays
        assert torch.equal(snap, over), 'a bucket was released to RCCL before its gradients were final'
        assert loss_over == loss_plain
        scale = float(plain.abs().max())
        assert float((over - plain).abs().max()) <= max(4 * noise, 1e-6 * scale), (noise, scale)
        if noise == 0.0:
            assert torch.equal(over, plain)                                # deterministic kernels: bit for bit
        # the guard for chip-filling persistent kernels: with only 100 "CUs" the encoder's BPTT launches (80 resident
        # workgroups + the RCCL reserve) no longer fit beside a collective, so the main stream must wait for the buckets first
        monkeypatch.setenv('LAS_DIST_FAKE_CUS', '100')
        guarded, _ = _c3_step(ops, asr, ldist, model, x, y)
        assert ldist.STATS['guard_waits'] >= 1 and ldist.STATS['collectives'] >= 2, ldist.STATS
        assert float((guarded - plain).abs().max()) <= max(4 * noise, 1e-6 * scale)
        assert int(model.status.item()) == 0
    finally:
        dist.destroy_process_group()
