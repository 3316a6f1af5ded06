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


def _mean_nearest(src, dst, squared):
    d = sided_distance(src, dst)[0]
    return (d if squared else d.sqrt()).mean(dim=-1)


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
    forward_term = _mean_nearest(p1, p2, squared)
    backward_term = _mean_nearest(p2, p1, squared)
    if w1 == 1 and w2 == 1:
        return forward_term + backward_term
    return w1 * forward_term + w2 * backward_term


def f_score(gt_points, pred_points, radius=0.01, eps=1e-8):
    r"""F-score of a predicted cloud: a predicted point is a true positive when a ground-truth point lies within
    ``radius``; misses are ground-truth points with no prediction within ``radius``
    (reference: kaolin/metrics/pointcloud.py:138-184).

    Returns:
        (torch.Tensor): :math:`(B)`.
    """
    dtype = gt_points.dtype
    gt_to_pred = sided_distance(gt_points, pred_points)[0].sqrt()
    pred_to_gt = sided_distance(pred_points, gt_points)[0].sqrt()
    missed = (gt_to_pred > radius).sum(dim=1).type(dtype)          # false negatives
    spurious = (pred_to_gt > radius).sum(dim=1).type(dtype)        # false positives
    hits = (pred_to_gt.shape[1] - spurious).type(dtype)            # true positives
    precision, recall = hits / (hits + spurious), hits / (hits + missed)
    return 2 * (precision * recall) / (precision + recall + eps)
