"""Render metrics (pure torch, no native code).

``mask_iou`` is the silhouette loss paired with ``dibr_soft_mask`` in the DIB-R training
loop.  Behaviour follows kaolin/metrics/render.py:18-40: soft intersection = product,
soft union = sum - product, loss = 1 - mean_b(I_b / (U_b + 1e-10)).
"""
import torch

__all__ = ['mask_iou']


def mask_iou(lhs_mask, rhs_mask):
    r"""IoU loss between two (soft) segmentation masks of shape :math:`(B, H, W)`.

    Returns:
        (torch.Tensor): scalar ``1 - mean IoU`` over the batch.
    """
    if lhs_mask.shape != rhs_mask.shape or lhs_mask.dim() != 3:
        raise AssertionError('mask_iou expects two masks of identical shape (B, H, W)')
    inter = (lhs_mask * rhs_mask).flatten(1)
    union = (lhs_mask + rhs_mask).flatten(1) - inter
    per_item = inter.sum(dim=1) / (union.sum(dim=1) + 1e-10)
    return 1.0 - per_item.mean()
