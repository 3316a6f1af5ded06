"""Point-cloud metrics: ``sided_distance``, ``chamfer_distance``, ``f_score``.

API mirror of kaolin/metrics/pointcloud.py (reference file:line cited per function).
The nearest-neighbour search and its gradient run in hand-written HIP kernels
(kaolin_amd/csrc/sided_distance.hip) reached through ``kaolin_amd._C.metrics``.
"""
import torch

from .. import _C

__all__ = ['sided_distance', 'chamfer_distance', 'f_score']


class _SidedDistanceFunction(torch.autograd.Function):
    """autograd shim with the contract of the reference's (kaolin/metrics/pointcloud.py:20-49): the nearest index is a
    non-differentiable output, both clouds receive gradients."""

    @staticmethod
    def forward(ctx, p1, p2):
        queries, targets = p1.contiguous(), p2.contiguous()
        dist, nearest = _C.metrics.sided_distance_forward_cuda(queries, targets)
        ctx.mark_non_differentiable(nearest)
        ctx.set_materialize_grads(False)
        ctx.save_for_backward(queries, targets, nearest)
        return dist, nearest

    @staticmethod
    def backward(ctx, grad_dist, _grad_idx):
        if grad_dist is None:
            return None, None
        queries, targets, nearest = ctx.saved_tensors
        return tuple(_C.metrics.sided_distance_backward_cuda(grad_dist.contiguous(), queries, targets, nearest))


def sided_distance(p1, p2):
    r"""For every point of ``p1``: squared euclidean distance to, and index of, its nearest point in ``p2``
    (reference: kaolin/metrics/pointcloud.py:51-87).

    Args:
        p1 (torch.Tensor): :math:`(\text{batch_size}, \text{num_points1}, 3)`.
        p2 (torch.Tensor): :math:`(\text{batch_size}, \text{num_points2}, 3)`.

    Returns:
        (torch.Tensor, torch.LongTensor): squared distances :math:`(B, N_1)` and the (lowest) index of the nearest
        point of ``p2`` for every point of ``p1``.
    """
    return _SidedDistanceFunction.apply(p1, p2)


class _SidedDistancePairFunction(torch.autograd.Function):
    """``sided_distance(p1, p2)[0]`` and ``sided_distance(p2, p1)[0]`` from one binning pass over both clouds (``_C.metrics.sided_distance_pair_forward``); the gradients are the two reference backward calls."""

    @staticmethod
    def forward(ctx, p1, p2):
        a, b = p1.contiguous(), p2.contiguous()
        both = _C.metrics.sided_distance_pair_forward(a, b)
        if both is None:
            dist1, near1 = _C.metrics.sided_distance_forward_cuda(a, b)
            dist2, near2 = _C.metrics.sided_distance_forward_cuda(b, a)
        else:
            dist1, near1, dist2, near2 = both
        ctx.set_materialize_grads(False)
        ctx.save_for_backward(a, b, near1, near2)
        return dist1, dist2

    @staticmethod
    def backward(ctx, grad1, grad2):
        a, b, near1, near2 = ctx.saved_tensors
        grad_a = grad_b = None
        if grad1 is not None:
            grad_a, grad_b = _C.metrics.sided_distance_backward_cuda(grad1.contiguous(), a, b, near1)
        if grad2 is not None:
            more_b, more_a = _C.metrics.sided_distance_backward_cuda(grad2.contiguous(), b, a, near2)
            grad_a = more_a if grad_a is None else grad_a.add_(more_a)
            grad_b = more_b if grad_b is None else grad_b.add_(more_b)
        return grad_a, grad_b


class _ChamferDistanceFunction(torch.autograd.Function):
    """The whole of ``chamfer_distance`` as one autograd node for fp32 clouds on the GPU.  Large clouds: one operator
    (``_C.metrics.chamfer_distance_forward``: workspace fill + grid build + search; the search launch also reduces the
    two means and, when a gradient will be wanted, leaves d value / d points in the state) and a one-launch backward.
    Smaller clouds: the two searches, the reference's expression for the value and one gradient kernel for both clouds
    (``_C.metrics.chamfer_distance_backward``) instead of a chain of ~10 small autograd nodes."""

    @staticmethod
    def forward(ctx, p1, p2, w1, w2, squared):
        a, b = p1.contiguous(), p2.contiguous()
        ctx.set_materialize_grads(False)
        with_grad = p1.requires_grad or p2.requires_grad
        fused = _C.metrics.chamfer_distance_forward(a, b, w1, w2, squared, with_grad)
        if fused is not None:
            value, state = fused
            ctx.fused_shape = (a.shape[0], a.shape[1], b.shape[1])
            if with_grad:
                ctx.save_for_backward(state)
            return value
        ctx.fused_shape = None
        dist1, near1 = _C.metrics.sided_distance_forward_cuda(a, b)
        dist2, near2 = _C.metrics.sided_distance_forward_cuda(b, a)
        ctx.save_for_backward(a, b, near1, near2, dist1, dist2)
        ctx.weights, ctx.squared = (w1, w2), squared
        return _chamfer_value(dist1, dist2, w1, w2, squared)

    @staticmethod
    def backward(ctx, grad):
        if grad is None:
            return None, None, None, None, None
        if ctx.fused_shape is not None:
            state, = ctx.saved_tensors
            grad_a, grad_b = _C.metrics.chamfer_distance_backward_fused(grad.contiguous(), state, *ctx.fused_shape)
            return grad_a, grad_b, None, None, None
        a, b, near1, near2, dist1, dist2 = ctx.saved_tensors
        grad_a, grad_b = _C.metrics.chamfer_distance_backward(grad.contiguous(), ctx.weights[0], ctx.weights[1],
                                                              ctx.squared, a, b, near1, near2, dist1, dist2)
        return grad_a, grad_b, None, None, None


def _chamfer_value(to_p2, to_p1, w1, w2, squared):
    forward_term = (to_p2 if squared else to_p2.sqrt()).mean(dim=-1)
    backward_term = (to_p1 if squared else to_p1.sqrt()).mean(dim=-1)
    if w1 == 1 and w2 == 1:
        return forward_term + backward_term
    return w1 * forward_term + w2 * backward_term


def _nearest_both_ways(p1, p2):
    """(distances p1 -> p2, distances p2 -> p1).  Large fp32 / fp64 clouds on the GPU are binned once for both searches."""
    if p1.is_cuda and p1.dtype in (torch.float32, torch.float64) and p2.dtype == p1.dtype and p1.dim() == 3 and p2.dim() == 3:
        return _SidedDistancePairFunction.apply(p1, p2)
    return sided_distance(p1, p2)[0], sided_distance(p2, p1)[0]


def chamfer_distance(p1, p2, w1=1., w2=1., squared=True):
    r"""Chamfer distance: ``w1 * mean_i d(p1_i, p2) + w2 * mean_j d(p2_j, p1)`` with ``d`` the (squared) distance to
    the nearest point of the other cloud (reference: kaolin/metrics/pointcloud.py:89-136).

    Args:
        p1, p2 (torch.Tensor): :math:`(B, N_1, 3)` and :math:`(B, N_2, 3)`.
        w1, w2 (float): weights of the two directions. Default: 1.
        squared (bool): squared distances (default) or their square roots.

    Returns:
        (torch.Tensor): :math:`(B)`.
    """
    if (p1.is_cuda and p1.dtype == torch.float32 and p2.dtype == torch.float32 and p1.dim() == 3 and p2.dim() == 3
            and p1.size(1) > 0 and p2.size(1) > 0
            and isinstance(w1, (int, float)) and isinstance(w2, (int, float))):
        return _ChamferDistanceFunction.apply(p1, p2, w1, w2, squared)
    to_p2, to_p1 = _nearest_both_ways(p1, p2)
    return _chamfer_value(to_p2, to_p1, w1, w2, squared)


def f_score(gt_points, pred_points, radius=0.01, eps=1e-8):
    r"""F-score of a predicted cloud: a predicted point is a true positive when a ground-truth point lies within
    ``radius``; misses are ground-truth points with no prediction within ``radius``
    (reference: kaolin/metrics/pointcloud.py:138-184).

    Returns:
        (torch.Tensor): :math:`(B)`.
    """
    dtype = gt_points.dtype
    gt_to_pred, pred_to_gt = (d.sqrt() for d in _nearest_both_ways(gt_points, pred_points))
    missed = (gt_to_pred > radius).sum(dim=1).type(dtype)          # false negatives
    spurious = (pred_to_gt > radius).sum(dim=1).type(dtype)        # false positives
    hits = (pred_to_gt.shape[1] - spurious).type(dtype)            # true positives
    precision, recall = hits / (hits + spurious), hits / (hits + missed)
    return 2 * (precision * recall) / (precision + recall + eps)


def _sided_distance(p1, p2):
    """Dense torch formulation of :func:`sided_distance` (values only), any device: the oracle the reference's tests
    compare the operator with (kaolin/metrics/pointcloud.py:186-197).  O(N1 * N2) memory per batch item."""
    delta = p1.unsqueeze(2) - p2.unsqueeze(1)           # (B, N1, N2, 3)
    return (delta * delta).sum(dim=-1).min(dim=-1).values
