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

__all__ = ['init_from_env', 'is_distributed', 'rank', 'world_size', 'shard_range', 'shard', 'shard_views',
           'all_reduce_gradients', 'SharedGradientReducer', 'all_gather_batch', 'barrier']


def _forced():
    """KAMD_DIST_FORCE=1: keep the whole distributed control flow (process group, gradient hooks, collectives, barriers) for
    a world of ONE rank -- lets a 1-GPU box execute the RCCL path (``python -m torch.distributed.run --nproc-per-node 1``)."""
    return os.environ.get('KAMD_DIST_FORCE') == '1'


def init_from_env(backend=None):
    """Initialises the default process group from RANK / WORLD_SIZE / MASTER_ADDR / MASTER_PORT (as exported by
    ``python -m torch.distributed.run``); binds this process to GPU LOCAL_RANK.  No-op for a single process (unless
    KAMD_DIST_FORCE=1, see :func:`_forced`)."""
    world = int(os.environ.get('WORLD_SIZE', '1'))
    if (world <= 1 and not (_forced() and 'RANK' in os.environ)) or dist.is_initialized():
        return world > 1 or (_forced() and dist.is_initialized())
    use_cuda = torch.cuda.is_available()
    if use_cuda:
        torch.cuda.set_device(int(os.environ.get('LOCAL_RANK', '0')) % max(torch.cuda.device_count(), 1))
    os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
    # (KAMD_DIST_BACKEND=gloo: several ranks sharing one GPU, where RCCL refuses duplicate devices -- tests of the control flow)
    backend = backend or os.environ.get('KAMD_DIST_BACKEND') or ('nccl' if use_cuda else 'gloo')
    dist.init_process_group(backend, init_method='env://')
    return True


def is_distributed():
    return dist.is_available() and dist.is_initialized() and (dist.get_world_size() > 1 or _forced())


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


def shard_views(*tensors, dim=0):
    """This rank's contiguous slice of per-view tensors (camera positions / rotations / translations, target images ...):
    config C4 of SURVEY.md 8(e) -- the views of one mesh are split over the ranks, the mesh itself is replicated.
    Returns one tensor, or a tuple in the order given; all of them must have the same number of views along `dim`."""
    n = tensors[0].shape[dim]
    if any(t.shape[dim] != n for t in tensors):
        raise ValueError('shard_views: tensors disagree on the number of views')
    out = tuple(shard(t, dim) for t in tensors)
    return out[0] if len(out) == 1 else out


def _all_reduce_tensor(t, async_op=False):
    """SUM all-reduce of one tensor in place.  RCCL ("nccl") reduces device memory directly; the gloo backend of the CPU
    tests takes GPU tensors through a host copy."""
    if t.is_cuda and dist.get_backend() == 'gloo':
        host = t.detach().cpu()
        dist.all_reduce(host, op=dist.ReduceOp.SUM)
        t.copy_(host)
        return None
    return dist.all_reduce(t, op=dist.ReduceOp.SUM, async_op=async_op)


class SharedGradientReducer:
    """Posts the all-reduce of every shared parameter's gradient from autograd itself, the moment that gradient has been
    accumulated (``register_post_accumulate_grad_hook``), instead of after ``backward()`` returns: the collective of a
    parameter whose gradient is ready early (a texture, a global offset) overlaps the rest of the backward pass, and the
    last one overlaps the host's return from ``backward()`` and whatever the caller enqueues next.  ``wait()`` (before
    the optimizer step) blocks until every posted collective has completed.  One small collective per parameter: the
    payloads of this path are latency-bound on xGMI (300 KB of vertex gradient at C4), and posting early beats bucketing
    late.  Single process: a no-op.

        reducer = SharedGradientReducer([vertices, texture])
        loss.backward(); reducer.wait(); optimizer.step(); optimizer.zero_grad()

    The hook reduces ``p.grad`` ITSELF, i.e. whatever has accumulated there: clear the gradients (``zero_grad`` /
    ``p.grad = None``) between two ``backward()`` calls.  When gradients are accumulated over several micro-batches, wrap all
    but the last ``backward()`` in :meth:`no_sync` (as with DistributedDataParallel) -- otherwise the sum reduced after the
    first micro-batch is reduced again after the second and counts ``world_size`` times.  The hooks stay registered until
    :meth:`remove`.
    """

    def __init__(self, params, average=False, single_bucket=False):
        """``single_bucket``: ONE collective per ``backward()`` whatever the number of shared parameters ("one
        all_reduce(SUM) of the shared-parameter gradient ... + texture grad if trained") -- the gradients are packed into one flat
        bucket and its all-reduce is posted from the hook of the LAST parameter whose gradient arrives (a parameter that receives
        no gradient in a pass is sent as zeros by :meth:`wait`).  Default: one collective per parameter, each posted the moment
        its gradient is ready."""
        self.params = [p for p in params if p.requires_grad]
        self.average = average
        self.single_bucket = bool(single_bucket) and len(self.params) > 1
        self._pending = []
        self._arrived = set()
        self._bucket = None
        self.posted = 0
        self._sync = True
        self._handles = [p.register_post_accumulate_grad_hook(self._hook) for p in self.params]

    def no_sync(self):
        """Context manager: ``backward()`` calls inside accumulate locally, nothing is posted (gradient accumulation over
        micro-batches; the first ``backward()`` outside reduces the accumulated sum once)."""
        reducer = self

        class _NoSync:
            def __enter__(self):
                self.prev, reducer._sync = reducer._sync, False

            def __exit__(self, *exc):
                reducer._sync = self.prev
                return False
        return _NoSync()

    def _hook(self, p):
        if not self._sync or not is_distributed() or p.grad is None:
            return
        if self.single_bucket:
            self._arrived.add(id(p))
            if len(self._arrived) == len(self.params):
                self._post_bucket()
            return
        g = p.grad if p.grad.is_contiguous() else p.grad.contiguous()
        self._pending.append((p, g, _all_reduce_tensor(g, async_op=True)))
        self.posted += 1

    def _post_bucket(self):
        dtype = self.params[0].dtype
        flat = torch.cat([(p.grad if p.grad is not None else torch.zeros_like(p)).reshape(-1).to(dtype) for p in self.params])
        self._bucket = (flat, _all_reduce_tensor(flat, async_op=True))
        self._arrived = set()
        self.posted += 1

    def wait(self):
        if self.single_bucket and self._sync and is_distributed():
            if self._bucket is None and self._arrived:
                self._post_bucket()     # (some parameter received no gradient in this pass: every rank still posts the same size)
            if self._bucket is not None:
                flat, work = self._bucket
                if work is not None:
                    work.wait()
                if self.average:
                    flat /= world_size()
                off = 0
                for p in self.params:
                    n = p.numel()
                    g = flat[off:off + n].view_as(p).to(p.dtype)
                    if p.grad is None:
                        p.grad = g.clone()
                    else:
                        p.grad.copy_(g)
                    off += n
                self._bucket = None
        for p, g, work in self._pending:
            if work is not None:
                work.wait()
            if self.average:
                g /= world_size()
            if g is not p.grad:
                p.grad.copy_(g)
        self._pending = []

    def remove(self):
        for h in self._handles:
            h.remove()
        self._handles = []


def all_reduce_gradients(params, average=False):
    """Sums (or averages) the gradients of shared parameters over all ranks with a SINGLE all-reduce over one
    flat bucket (params whose .grad is None contribute zeros so that every rank posts the same size)."""
    params = [p for p in params if p.requires_grad]
    if not params or not is_distributed():
        return
    if len(params) == 1 and params[0].grad is not None and params[0].grad.is_contiguous():
        # one shared tensor (the usual case: the mesh vertices): reduce its gradient in place, no bucket copy
        _all_reduce_tensor(params[0].grad)
        if average:
            params[0].grad /= world_size()
        return
    flat = torch.cat([(p.grad if p.grad is not None else torch.zeros_like(p)).reshape(-1).to(params[0].dtype)
                      for p in params])
    _all_reduce_tensor(flat)
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
