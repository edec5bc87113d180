"""One process per GPU; utterance batch sharded by rank; ONE sum all-reduce of the flat fp32 gradient vector over
RCCL/xGMI per step (new functionality: the reference is single-device).

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
