"""Render metrics.

``mask_iou`` is the silhouette loss paired with ``dibr_soft_mask`` in the DIB-R training loop.  Behaviour follows
kaolin/metrics/render.py:18-40: soft intersection = product, soft union = sum - product,
loss = 1 - mean_b(I_b / (U_b + 1e-10)).  On the GPU (float / double) it is one fused HIP pass each way
(kaolin_amd/csrc/render_metrics.hip); other inputs take the torch formulation below, which is also the
definition the fused path is tested against.
"""
import torch

from .. import _C

__all__ = ['mask_iou']


class _MaskIoUCuda(torch.autograd.Function):
    @staticmethod
    def forward(ctx, lhs_mask, rhs_mask):
        lhs, rhs = lhs_mask.contiguous(), rhs_mask.contiguous()
        loss, sums = _C.render.mesh.mask_iou_forward_fused(lhs, rhs)
        ctx.save_for_backward(lhs, rhs, sums)
        return loss

    @staticmethod
    def backward(ctx, grad_loss):
        lhs, rhs, sums = ctx.saved_tensors
        g_lhs = _C.render.mesh.mask_iou_backward_fused(grad_loss, rhs, sums) if ctx.needs_input_grad[0] else None
        g_rhs = _C.render.mesh.mask_iou_backward_fused(grad_loss, lhs, sums) if ctx.needs_input_grad[1] else None
        return g_lhs, g_rhs


def _mask_iou_torch(lhs_mask, rhs_mask):
    inter = (lhs_mask * rhs_mask).flatten(1)
    union = (lhs_mask + rhs_mask).flatten(1) - inter
    per_item = inter.sum(dim=1) / (union.sum(dim=1) + 1e-10)
    return 1.0 - per_item.mean()


def mask_iou(lhs_mask, rhs_mask):
    r"""IoU loss between two (soft) segmentation masks of shape :math:`(B, H, W)`.

    Returns:
        (torch.Tensor): scalar ``1 - mean IoU`` over the batch.
    """
    if lhs_mask.shape != rhs_mask.shape or lhs_mask.dim() != 3:
        raise AssertionError('mask_iou expects two masks of identical shape (B, H, W)')
    if (lhs_mask.is_cuda and rhs_mask.is_cuda and lhs_mask.dtype == rhs_mask.dtype and
            lhs_mask.dtype in (torch.float32, torch.float64) and lhs_mask.numel() > 0 and
            lhs_mask.shape[0] <= 65535):      # (the kernels put the batch on a grid dimension)
        return _MaskIoUCuda.apply(lhs_mask, rhs_mask)
    return _mask_iou_torch(lhs_mask, rhs_mask)
