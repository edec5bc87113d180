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


# ------------------------------------------------------------------------------------------------------------------
# A bucket smaller than the world: the rank with the empty shard has no backward, but must issue the same collectives, in
# the same order and sizes, as the rank that runs one (under nccl a different sequence hangs or corrupts).
class _FakeModel:
    """What dist.py needs of a model: the flat gradient vector, the status word, the offsets at which backward fires."""

    def __init__(self, n, offsets):
        self.flat_grads = torch.zeros(n)
        self.status = torch.zeros(1, dtype=torch.int32)
        self._offs = offsets

    def grad_ready_offsets(self):
        return list(self._offs)


class _FakeLoss:
    """backward() = what the real autograd graph does to dist.py: fills the gradient and fires ops._GRAD_READY per encoder
    layer, top layer first, with that layer's first gradient view."""

    def __init__(self, model):
        self.m = model

    def backward(self):
        ops = importlib.import_module('end-to-end-asr-pytorch_amd.ops')
        self.m.flat_grads.copy_(torch.arange(self.m.flat_grads.numel(), dtype=torch.float32))
        for lo in self.m.grad_ready_offsets():
            if ops._GRAD_READY is not None:
                ops._GRAD_READY(self.m.flat_grads[lo:])


def _seq_worker(rank, world, port, q, force_overlap_logic):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), WORLD_SIZE=str(world), RANK=str(rank), LOCAL_RANK=str(rank))
    ldist = importlib.import_module('end-to-end-asr-pytorch_amd.dist')
    ldist.init(backend='gloo')
    log = []
    real = dist.all_reduce

    def spy(t, op=dist.ReduceOp.SUM, async_op=False, **kw):
        log.append((int(t.numel()), str(op), t.dtype == torch.float32))
        return real(t, op=op, async_op=async_op, **kw)
    dist.all_reduce = spy
    if force_overlap_logic:
        # the nccl branch's bookkeeping (BucketPlan driven by the hooks / by the replay) with CPU tensors: what differs from
        # the real thing is only the stream plumbing, which tests/test_dist_gpu.py executes on the GPU
        ldist.overlap_active = lambda flat: True
        ldist._comm_stream = lambda: None

        class _Ex(ldist._Exchange):
            def ready(self, lo):
                for off, end in self.plan.ready(lo):
                    self.works.append(dist.all_reduce(self.flat[off:end], op=dist.ReduceOp.SUM, async_op=True))
                    ldist.STATS['collectives'] += 1
        ldist._Exchange = _Ex
    m = _FakeModel(1000, [704, 448, 64])
    if rank == 0:
        ldist.backward_with_overlap(_FakeLoss(m), m, bucket_elems=200)
    else:
        ldist.exchange_without_backward(m, bucket_elems=200)       # empty shard: zero gradient, same schedule
    q.put((rank, log, m.flat_grads.clone().numpy()))
    dist.barrier()
    dist.destroy_process_group()


def _run_seq(force):
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    port = 31500 + (os.getpid() % 2000) + (7 if force else 0)
    procs = [ctx.Process(target=_seq_worker, args=(r, 2, port, q, force)) for r in range(2)]
    for p in procs:
        p.start()
    got = dict((r, (log, g)) for r, log, g in (q.get(timeout=120) for _ in range(2)))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    return got


def test_empty_shard_issues_the_same_collectives_as_its_peers():
    for force in (False, True):
        got = _run_seq(force)
        assert got[0][0] == got[1][0], (force, got[0][0], got[1][0])            # same count, sizes, ops, order
        want = np.arange(1000, dtype=np.float32)                                # rank 0's gradient + rank 1's zeros
        np.testing.assert_array_equal(got[0][1], want)
        np.testing.assert_array_equal(got[1][1], want)
        n_grad = [n for n, _, f32 in got[0][0] if f32]
        assert sum(n_grad) == 1000                                              # the vector went out exactly once
        if force:
            assert n_grad == [200, 96, 200, 56, 200, 184, 64], n_grad           # per-layer buckets, top layer first
        assert got[0][0][-1][0] == 1 and 'MIN' in got[0][0][-1][1]              # the status word's agreement comes last
