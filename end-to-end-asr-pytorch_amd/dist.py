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


STATS = {'collectives': 0, 'guard_waits': 0}     # counters of the most recent exchange (tests, LAS_DIST_DEBUG)
RCCL_CU_RESERVE = 32        # CUs a collective's channels may need while a persistent kernel is resident (guard below)
_ACTIVE = {'ex': None}


def overlap_active(flat):
    """True when the gradient exchange runs DURING backward: RCCL only (backend 'nccl': its collectives are device-side work
    on a stream of their own).  gloo moves HIP tensors through host copies issued by worker threads; launched in the middle
    of backward (two ranks rehearsing on ONE GPU, persistent kernels of both resident) each of its collectives took 2-3 s
    (13 s per step against 0.12 s with the exchange after backward), so gloo keeps the exchange after backward.
    LAS_DIST_NO_OVERLAP=1 forces that path everywhere; LAS_DIST_FORCE_OVERLAP=1 (tests only) takes the overlap branch for a
    1-rank nccl group too, which is how the branch is executed on a one-GPU box."""
    if not dist.is_initialized() or not flat.is_cuda or os.environ.get('LAS_DIST_NO_OVERLAP') or dist.get_backend() != 'nccl':
        return False
    return dist.get_world_size() > 1 or bool(os.environ.get('LAS_DIST_FORCE_OVERLAP'))


class _Exchange:
    """One step's bucketed all-reduce of the flat gradient vector on RCCL's stream.  `ready(lo)` = "every gradient at flat
    offsets >= lo has been enqueued (main or side streams)": the not-yet-sent ranges go out behind the tails of exactly those
    streams.  Every rank MUST make the same sequence of ready() calls (same ranges, same order): RCCL matches collectives by
    issue order.  `snapshot` (tests): right where a bucket's collective may start, the bucket is also copied aside on the comm
    stream; if the copy later differs from the final gradient, the collective was released before its gradients were final."""

    def __init__(self, flat, bucket_elems, snapshot=None):
        from . import ops
        self.ops, self.flat, self.works = ops, flat, []
        self.plan = BucketPlan(flat.numel(), bucket_elems)
        self.comm = _comm_stream()
        self.snapshot = snapshot
        STATS['collectives'] = STATS['guard_waits'] = 0

    def ready(self, lo):
        ranges = self.plan.ready(lo)
        if not ranges:
            return
        comm, flat = self.comm, self.flat
        comm.wait_stream(torch.cuda.current_stream())
        for side in self.ops.side_streams():                  # weight-gradient streams and the CTC branch
            comm.wait_stream(side)
        with torch.cuda.stream(comm):                         # the collective's stream waits for comm's tail only
            for off, end in ranges:
                if self.snapshot is not None:
                    self.snapshot[off:end].copy_(flat[off:end])
                self.works.append(dist.all_reduce(flat[off:end], op=dist.ReduceOp.SUM, async_op=True))
                STATS['collectives'] += 1

    def hold_for_persistent(self, resident_wgs, n_cus):
        """A persistent kernel is about to be launched on the current stream while collectives of this exchange may be in
        flight.  Its workgroups must ALL be co-resident (they spin on each other), one per CU; an RCCL kernel that holds CUs
        the launch needs leaves part of the grid unscheduled until the collective ends, and a collective that finds every CU
        taken waits for the persistent kernel -- on every rank at a different moment.  If the launch does not leave
        RCCL_CU_RESERVE CUs free, the current stream first waits for the outstanding collectives (they then overlap with the
        non-persistent kernels between the recurrences only); otherwise both run side by side."""
        if self.works and resident_wgs + RCCL_CU_RESERVE > n_cus:
            for w in self.works:
                w.wait()                                      # stream-level wait: the host does not block
            self.done = getattr(self, 'done', 0) + len(self.works)
            self.works = []
            STATS['guard_waits'] += 1

    def finish(self):
        for w in self.works:
            w.wait()                                          # the current stream waits for the collective
        self.works = []


def persistent_launch_guard(resident_wgs, device):
    """ops.py calls this before every persistent-kernel launch of backward (LSTM BPTT).  No-op outside an exchange."""
    ex = _ACTIVE['ex']
    if ex is not None:
        ex.hold_for_persistent(int(resident_wgs), torch.cuda.get_device_properties(device).multi_processor_count
                               if not os.environ.get('LAS_DIST_FAKE_CUS') else int(os.environ['LAS_DIST_FAKE_CUS']))


def backward_with_overlap(loss, model, bucket_elems=64 * 1024 * 1024, snapshot=None):
    """loss.backward() + the gradient exchange.  Single process: just backward and the side-stream join."""
    from . import ops
    flat = model.flat_grads
    if not overlap_active(flat):
        loss.backward()
        ops.join_side_stream()
        allreduce_grads(flat)
        _agree_on_status(model)
        return
    ex = _Exchange(flat, bucket_elems, snapshot)
    base = flat.data_ptr()
    dbg = os.environ.get('LAS_DIST_DEBUG') and dist.get_rank() == 0
    import time
    t0 = time.perf_counter()
    ops._GRAD_READY = lambda first_grad: ex.ready((first_grad.data_ptr() - base) // 4)
    _ACTIVE['ex'] = ex
    try:
        loss.backward()
    finally:
        ops._GRAD_READY = None
        _ACTIVE['ex'] = None
    t1 = time.perf_counter()
    ops.join_side_stream()
    ex.ready(0)
    t2 = time.perf_counter()
    ex.finish()
    _agree_on_status(model)
    if dbg:
        print('[dist] backward enqueue %.1f ms, last bucket %.1f ms, waits %.1f ms, %d collectives, %d guard waits' %
              ((t1 - t0) * 1e3, (t2 - t1) * 1e3, (time.perf_counter() - t2) * 1e3, STATS['collectives'], STATS['guard_waits']),
              flush=True)


def exchange_without_backward(model, bucket_elems=64 * 1024 * 1024):
    """A rank whose shard of the bucket is empty (bucket smaller than the world) has no backward to run but must issue the
    SAME collectives, in the same order and sizes, as its peers: it replays the bucket plan with the offsets at which their
    backward reports "this encoder layer and everything behind it is final" (Seq2Seq.grad_ready_offsets), on its zero
    gradient.  Without overlap every rank makes the same allreduce_grads call."""
    flat = model.flat_grads
    if not overlap_active(flat):
        allreduce_grads(flat)
        _agree_on_status(model)
        return
    ex = _Exchange(flat, bucket_elems)
    for lo in model.grad_ready_offsets():
        ex.ready(lo)
    ex.ready(0)
    ex.finish()
    _agree_on_status(model)


def _agree_on_status(model):
    """MIN (LAS_E_* codes are negative) of the persistent kernels' status word over the ranks, in place and on the device: a
    hand-off timeout on ONE rank then raises LasError on EVERY rank at the same step (Trainer._check_status), instead of
    leaving its peers waiting in the next collective for a rank that has aborted.  One 4-byte collective per step."""
    st = getattr(model, 'status', None)
    if st is not None and dist.is_initialized() and dist.get_world_size() > 1:
        dist.all_reduce(st, op=dist.ReduceOp.MIN)
