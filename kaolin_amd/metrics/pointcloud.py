"""Point-cloud metrics: ``sided_distance``, ``chamfer_distance``, ``f_score``.

API mirror of kaolin/metrics/pointcloud.py (reference file:line cited per function).
The nearest-neighbour search and its gradient run in hand-written HIP kernels
(kaolin_amd/csrc/sided_distance.hip) reached through ``kaolin_amd._C.metrics``.
"""
import torch

from .. import _C

__all__ = ['sided_distance', 'chamfer_distance', 'f_score']


class _SidedDistanceFunction(torch.autograd.Function):
    """autograd shim, same contract as kaolin/metrics/pointcloud.py:20-49: saves
    (p1, p2, idx), idx is non-differentiable, backward returns (grad_p1, grad_p2)."""

    @staticmethod
    def forward(ctx, p1, p2):
        p1 = p1.contiguous()
        p2 = p2.contiguous()
        dist, idx = _C.metrics.sided_distance_forward_cuda(p1, p2)
        ctx.save_for_backward(p1, p2, idx)
        ctx.mark_non_differentiable(idx)
        return dist, idx

    @staticmethod
    def backward(ctx, grad_output_dist, grad_output_idx):
        p1, p2, idx = ctx.saved_tensors
        grad_p1, grad_p2 = _C.metrics.sided_distance_backward_cuda(
            grad_output_dist.contiguous(), p1, p2, idx)
        return grad_p1, grad_p2


def sided_distance(p1, p2):
    r"""For each point in :math:`p_{1i} \in P_1` finds the index and squared euclidean distance
    of the closest point in :math:`P_2` (reference: kaolin/metrics/pointcloud.py:51-87).

    Args:
        p1 (torch.Tensor): of shape :math:`(\text{batch_size}, \text{num_points1}, 3)`.
        p2 (torch.Tensor): of shape :math:`(\text{batch_size}, \text{num_points2}, 3)`.

    Returns:
        (torch.Tensor, torch.LongTensor): squared distances :math:`(B, N_1)` and, for every
        point of ``p1``, the (lowest) index of its nearest point in ``p2``.
    """
    dist, idx = _SidedDistanceFunction.apply(p1, p2)
    return dist, idx


def chamfer_distance(p1, p2, w1=1., w2=1., squared=True):
    r"""Chamfer distance between two point clouds: the weighted sum of the two mean sided
    distances (reference: kaolin/metrics/pointcloud.py:89-136).

    Args:
        p1, p2 (torch.Tensor): of shapes :math:`(B, N_1, 3)` and :math:`(B, N_2, 3)`.
        w1, w2 (float): weights of the p1->p2 and p2->p1 terms. Default: 1.
        squared (bool): use squared distances (default) or their square roots.

    Returns:
        (torch.Tensor): of shape :math:`(B)`.
    """
    sdist1 = sided_distance(p1, p2)[0]
    sdist2 = sided_distance(p2, p1)[0]
    if not squared:
        sdist1 = torch.sqrt(sdist1)
        sdist2 = torch.sqrt(sdist2)
    dist_to_p2 = sdist1.mean(dim=-1)
    dist_to_p1 = sdist2.mean(dim=-1)
    if w1 == 1 and w2 == 1:
        return dist_to_p2 + dist_to_p1
    return w1 * dist_to_p2 + w2 * dist_to_p1


def f_score(gt_points, pred_points, radius=0.01, eps=1e-8):
    r"""F-score of a predicted point cloud w.r.t. a ground-truth one: a prediction is a true
    positive when a ground-truth point lies within ``radius``
    (reference: kaolin/metrics/pointcloud.py:138-184).

    Returns:
        (torch.Tensor): of shape :math:`(B)`.
    """
    pred_distances = torch.sqrt(sided_distance(gt_points, pred_points)[0])
    gt_distances = torch.sqrt(sided_distance(pred_points, gt_points)[0])
    data_type = gt_points.dtype
    fn = torch.sum(pred_distances > radius, dim=1).type(data_type)
    fp = torch.sum(gt_distances > radius, dim=1).type(data_type)
    tp = (gt_distances.shape[1] - fp).type(data_type)
    precision = tp / (tp + fp)
    recall = tp / (tp + fn)
    return 2 * (precision * recall) / (precision + recall + eps)
