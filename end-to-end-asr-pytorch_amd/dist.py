"""One process per GPU; utterance batch sharded by rank; a sum all-reduce of the flat fp32 gradient vector over
RCCL/xGMI per step, in per-layer buckets launched DURING backward as soon as a bucket's last weight gradient is
enqueued (new functionality: the reference is single-device).

Both losses are per-utterance normalised then batch-meaned (solver.py:152-154, CTC 'mean'), so with equal shard
sizes mean-over-ranks of the local gradient equals the global-batch gradient: all-reduce(sum) then scale by
1/world inside the fused clip kernel (las_grad_norm gscale), and the clip / NaN-skip decision is made on the
global gradient identically on every rank."""
import os
import torch
import torch.distributed as dist


def env_world():
    return int(os.environ.get('WORLD_SIZE', '1')), int(os.environ.get('RANK', '0')), int(os.environ.get('LOCAL_RANK', '0'))


def init(backend=None):
    """Initialise torch.distributed from the torchrun environment.  Returns (world, rank, local_rank)."""
    world, rank, local = env_world()
    if world > 1 and not dist.is_initialized():
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        os.environ.setdefault('MASTER_PORT', '29500')
        if backend is None:
            backend = os.environ.get('LAS_DIST_BACKEND') or ('nccl' if torch.cuda.is_available() else 'gloo')
        if torch.cuda.is_available():
            torch.cuda.set_device(local % torch.cuda.device_count())
        dist.init_process_group(backend=backend, world_size=world, rank=rank)
    return world, rank, local


def allreduce_grads(flat_grads, bucket_elems=16 * 1024 * 1024):
    """Sum the flat gradient vector across ranks in a few large buckets (xGMI ring all-reduce is per-link bound:
    few, large messages).  No-op for a single process."""
    if not dist.is_initialized() or dist.get_world_size() == 1:
        return
    n = flat_grads.numel()
    if n <= bucket_elems:
        dist.all_reduce(flat_grads, op=dist.ReduceOp.SUM)
        return
    handles = []
    for off in range(0, n, bucket_elems):
        handles.append(dist.all_reduce(flat_grads[off:off + bucket_elems], op=dist.ReduceOp.SUM, async_op=True))
    for h in handles:
        h.wait()


def broadcast_params(flat_params):
    if dist.is_initialized() and dist.get_world_size() > 1:
        dist.broadcast(flat_params, src=0)


def shard_bucket(x, y, lens, rank, world):
    """Deal a length-sorted global bucket round-robin over ranks (balances sum T), keep descending order."""
    idx = list(range(rank, x.shape[0], world))
    return x[idx], y[idx], [lens[i] for i in idx]


# ----------------------------------------------------------------------------- bucketed all-reduce under backward
# The flat gradient vector is laid out in forward order: [VGG | encoder layer 0 (LSTM, proj) | layer 1 | ... | attention |
# decoder | embed | char_trans | ctc_layer].  Backward produces it from the END: when encoder layer l's BPTT has been
# enqueued, every gradient at or beyond that layer's first parameter is final (its weight-gradient GEMMs sit on the side
# stream, the decoder's returned gradients were accumulated on the main stream).  At that moment the range
# [layer l start, previous boundary) goes out as one all-reduce on RCCL's own stream, which is made to wait for exactly
# those two streams' tails, and runs under the BPTT of the layers below (the encoder's layer 0, the longest, lands last).
# xGMI rings are per-link bound, so buckets are whole layers (1.7 - 4.6 MB at C2, 34 - 134 MB at C5), not small chunks.
_COMM = {'stream': None}


def _comm_stream():
    if _COMM['stream'] is None:
        _COMM['stream'] = torch.cuda.Stream()
    return _COMM['stream']


class BucketPlan:
    """Which flat-gradient ranges go out when: ready(lo) = "every gradient at offsets >= lo has been enqueued" returns the
    not-yet-sent ranges [lo, previous boundary) cut into pieces of <= bucket_elems; together the calls cover [0, n) once."""

    def __init__(self, n, bucket_elems):
        self.hi, self.bucket = int(n), int(bucket_elems)

    def ready(self, lo):
        lo = max(0, min(int(lo), self.hi))
        out = [(off, min(off + self.bucket, self.hi)) for off in range(lo, self.hi, self.bucket)]
        self.hi = lo
        return out


def backward_with_overlap(loss, model, bucket_elems=64 * 1024 * 1024):
    """loss.backward() + the gradient exchange.  Single process: just backward and the side-stream join."""
    from . import ops
    flat = model.flat_grads
    # Overlap only over RCCL (backend 'nccl'): its collectives are device-side work on a stream of their own.  gloo moves HIP
    # tensors through host copies issued by worker threads; launched in the middle of backward (two ranks rehearsing on ONE
    # GPU, persistent kernels of both resident) each of its collectives then took 2-3 s (measured: 13 s per step against
    # 0.12 s with the exchange after backward), so gloo keeps the exchange after backward.  LAS_DIST_NO_OVERLAP=1 forces that.
    if (not dist.is_initialized() or dist.get_world_size() == 1 or not flat.is_cuda or os.environ.get('LAS_DIST_NO_OVERLAP')
            or dist.get_backend() != 'nccl'):
        loss.backward()
        ops.join_side_stream()
        allreduce_grads(flat)
        return
    st = {'works': []}
    plan = BucketPlan(flat.numel(), bucket_elems)
    comm = _comm_stream()
    base = flat.data_ptr()

    def ready(lo):
        """everything at flat offsets >= lo has been enqueued (main or side stream)"""
        ranges = plan.ready(lo)
        if not ranges:
            return
        comm.wait_stream(torch.cuda.current_stream())
        for side in ops.side_streams():                       # weight-gradient streams and the CTC branch
            comm.wait_stream(side)
        with torch.cuda.stream(comm):                         # the collective's stream waits for comm's tail only
            for off, end in ranges:
                st['works'].append(dist.all_reduce(flat[off:end], op=dist.ReduceOp.SUM, async_op=True))

    dbg = os.environ.get('LAS_DIST_DEBUG') and dist.get_rank() == 0
    import time
    t0 = time.perf_counter()
    ops._GRAD_READY = lambda first_grad: ready((first_grad.data_ptr() - base) // 4)
    try:
        loss.backward()
    finally:
        ops._GRAD_READY = None
    t1 = time.perf_counter()
    ops.join_side_stream()
    ready(0)
    t2 = time.perf_counter()
    for w in st['works']:
        w.wait()                                              # the current stream waits for the collective
    if dbg:
        print('[dist] backward enqueue %.1f ms, last bucket %.1f ms, waits %.1f ms, %d collectives' %
              ((t1 - t0) * 1e3, (t2 - t1) * 1e3, (time.perf_counter() - t2) * 1e3, len(st['works'])), flush=True)
