"""CPU, world_size 2 over gloo: the data-parallel recipe of dist.py -- shard the length-sorted bucket round-robin,
compute each shard's gradient of ITS mean loss, all-reduce(sum) the flat gradient, scale by 1/world -- reproduces
the full-batch gradient of the reference step (checked with the CPU oracle as the per-rank worker)."""
import importlib
import os
import sys
import numpy as np
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _worker(rank, world, port, q):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, 'tools'))
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), WORLD_SIZE=str(world), RANK=str(rank), LOCAL_RANK=str(rank))
    ldist = importlib.import_module('end-to-end-asr-pytorch_amd.dist')
    from oracle import las_ref as R
    from gen_golden import TINY
    w, r, _ = ldist.init(backend='gloo')
    assert (w, r) == (world, rank)
    d = np.load(os.path.join(ROOT, 'tests', 'golden', 'g3_step_loc_ctc.npz'))
    W = {k[2:]: d[k] for k in d.files if k.startswith('w.')}
    x, y = torch.tensor(d['x']), torch.tensor(d['y'])
    lens = R.infer_lengths(x)
    xs, ys, ls = ldist.shard_bucket(x, y, lens, rank, world)
    assert ls == sorted(ls, reverse=True) and len(ls) == x.shape[0] // world
    st = R.RefTrainStep(W, TINY['loc_ctc'])
    loss, _, _, _ = st.forward_loss(xs.numpy(), ys.numpy())
    loss.backward()
    names = sorted(st.W)
    flat = torch.cat([st.W[k].grad.reshape(-1) for k in names])
    ldist.allreduce_grads(flat, bucket_elems=1000)          # forces the multi-bucket path
    flat /= world
    lsum = loss.detach().clone()
    dist.all_reduce(lsum)
    if rank == 0:
        q.put((flat.numpy(), float(lsum) / world, names))
    dist.barrier()
    dist.destroy_process_group()


def test_sharded_allreduce_equals_full_batch():
    sys.path.insert(0, os.path.join(ROOT, 'tools'))
    from oracle import las_ref as R
    from gen_golden import TINY
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    port = 29500 + (os.getpid() % 2000)
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    flat, loss, names = q.get(timeout=120)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    d = np.load(os.path.join(ROOT, 'tests', 'golden', 'g3_step_loc_ctc.npz'))
    W = {k[2:]: d[k] for k in d.files if k.startswith('w.')}
    st = R.RefTrainStep(W, TINY['loc_ctc'])
    full, _, _, _ = st.forward_loss(d['x'], d['y'])
    full.backward()
    ref = torch.cat([st.W[k].grad.reshape(-1) for k in names]).numpy()
    assert abs(loss - float(full)) < 1e-5
    np.testing.assert_allclose(flat, ref, atol=2e-6)


def test_bucket_plan_covers_the_flat_gradient_once():
    """dist.backward_with_overlap's bookkeeping: the ranges sent at the successive "layer l and everything after it is
    final" calls (the encoder layers' first-parameter offsets, descending, then 0) are disjoint, cover [0, n) exactly once
    and respect the bucket size; a repeated or out-of-order call sends nothing twice."""
    sys.path.insert(0, ROOT)
    ldist = importlib.import_module('end-to-end-asr-pytorch_amd.dist')
    n = 1000
    plan = ldist.BucketPlan(n, 300)
    sent = []
    for lo in [700, 700, 820, 410, 64, 0, 0]:
        sent += plan.ready(lo)
    assert all(0 < e - o <= 300 for o, e in sent)
    cover = np.zeros(n, int)
    for o, e in sent:
        cover[o:e] += 1
    assert (cover == 1).all()
    assert sent[0] == (700, 1000) and sent[-1][0] == 0
