"""Batch / camera-view sharding over RCCL (new functionality: the reference has no distributed code).

One process per GPU (``torch.distributed``, backend ``nccl`` = RCCL on ROCm; ``gloo`` on CPU for tests).
Every kernel on the hot path treats batch items / views independently (rasterization_cuda.cu:62,80;
dibr_soft_mask_cuda.cu:51,83; sided_distance_cuda.cu:60,214), so the data path needs NO collective: each
rank renders / measures its own contiguous slice of the batch.  The only exchange is the gradient of
parameters shared by all items (mesh vertices, textures, a global offset ...): ONE all-reduce(SUM) over a
single flat bucket per step.  The payload is small (25k x 3 fp32 = 300 KB for config C4), i.e. latency-bound
on xGMI: one bucket, one collective, no per-tensor calls.
"""
import os

import torch
import torch.distributed as dist

__all__ = ['init_from_env', 'is_distributed', 'rank', 'world_size', 'shard_range', 'shard', 'all_reduce_gradients',
           'all_gather_batch', 'barrier']


def init_from_env(backend=None):
    """Initialises the default process group from RANK / WORLD_SIZE / MASTER_ADDR / MASTER_PORT (as exported by
    ``python -m torch.distributed.run``); binds this process to GPU LOCAL_RANK.  No-op for a single process."""
    world = int(os.environ.get('WORLD_SIZE', '1'))
    if world <= 1 or dist.is_initialized():
        return world > 1
    use_cuda = torch.cuda.is_available()
    if use_cuda:
        torch.cuda.set_device(int(os.environ.get('LOCAL_RANK', '0')) % max(torch.cuda.device_count(), 1))
    os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
    dist.init_process_group(backend or ('nccl' if use_cuda else 'gloo'), init_method='env://')
    return True


def is_distributed():
    return dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1


def rank():
    return dist.get_rank() if is_distributed() else 0


def world_size():
    return dist.get_world_size() if is_distributed() else 1


def shard_range(num_items, rank_=None, world_=None):
    """[begin, end) of the contiguous slice of `num_items` owned by a rank (sizes differ by at most one)."""
    r = rank() if rank_ is None else rank_
    w = world_size() if world_ is None else world_
    base, extra = divmod(num_items, w)
    begin = r * base + min(r, extra)
    return begin, begin + base + (1 if r < extra else 0)


def shard(tensor, dim=0):
    """This rank's contiguous slice of a replicated tensor along `dim`."""
    b, e = shard_range(tensor.shape[dim])
    return tensor.narrow(dim, b, e - b)


def all_reduce_gradients(params, average=False):
    """Sums (or averages) the gradients of shared parameters over all ranks with a SINGLE all-reduce over one
    flat bucket (params whose .grad is None contribute zeros so that every rank posts the same size)."""
    params = [p for p in params if p.requires_grad]
    if not params or not is_distributed():
        return
    if len(params) == 1 and params[0].grad is not None and params[0].grad.is_contiguous():
        # one shared tensor (the usual case: the mesh vertices): reduce its gradient in place, no bucket copy
        dist.all_reduce(params[0].grad, op=dist.ReduceOp.SUM)
        if average:
            params[0].grad /= world_size()
        return
    flat = torch.cat([(p.grad if p.grad is not None else torch.zeros_like(p)).reshape(-1).to(params[0].dtype)
                      for p in params])
    dist.all_reduce(flat, op=dist.ReduceOp.SUM)
    if average:
        flat /= world_size()
    off = 0
    for p in params:
        n = p.numel()
        g = flat[off:off + n].view_as(p).to(p.dtype)
        if p.grad is None:
            p.grad = g.clone()
        else:
            p.grad.copy_(g)
        off += n


def all_gather_batch(tensor):
    """Concatenates equally-sized per-rank batch slices along dim 0 (for callers that want the full batch)."""
    if not is_distributed():
        return tensor
    out = [torch.empty_like(tensor) for _ in range(world_size())]
    dist.all_gather(out, tensor.contiguous())
    return torch.cat(out, dim=0)


def barrier():
    if is_distributed():
        dist.barrier()
