"""Render metrics.

``mask_iou`` is the silhouette loss paired with ``dibr_soft_mask`` in the DIB-R training loop.  Behaviour follows
kaolin/metrics/render.py:18-40: soft intersection = product, soft union = sum - product,
loss = 1 - mean_b(I_b / (U_b + 1e-10)).  On the GPU (float / double) it is one fused HIP pass each way
(kaolin_amd/csrc/render_metrics.hip); other inputs take the torch formulation below, which is also the
definition the fused path is tested against.
"""
import torch

from .. import _C

__all__ = ['mask_iou', 'weighted_sum']


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
    if (lhs_mask.is_cuda and rhs_mask.is_cuda and lhs_mask.device == rhs_mask.device and lhs_mask.dtype == rhs_mask.dtype and
            lhs_mask.dtype in (torch.float32, torch.float64) and lhs_mask.numel() > 0 and
            lhs_mask.shape[0] <= 65535):      # (the kernels put the batch on a grid dimension)
        return _MaskIoUCuda.apply(lhs_mask, rhs_mask)
    return _mask_iou_torch(lhs_mask, rhs_mask)


class _WeightedSum2Cuda(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x1, w1, x2, w2):
        x1, w1 = x1.contiguous(), w1.contiguous()
        if x2 is not None:
            x2, w2 = x2.contiguous(), w2.contiguous()
        ctx.w = (w1, w2)
        return _C.render.mesh.weighted_sum2_forward(x1, w1, x2, w2)

    @staticmethod
    def backward(ctx, grad_out):
        w1, w2 = ctx.w
        g1, g2 = _C.render.mesh.weighted_sum2_backward(grad_out, w1, w2, ctx.needs_input_grad[0],
                                                       w2 is not None and ctx.needs_input_grad[2])
        return g1, None, g2, None


def _fusable(x, w):
    return (x.is_cuda and w.is_cuda and x.device == w.device and x.dtype == w.dtype and
            x.dtype in (torch.float32, torch.float64) and x.shape == w.shape and not w.requires_grad)


def weighted_sum(image, image_weights, mask=None, mask_weights=None):
    r"""The linear loss :math:`\sum image \cdot image\_weights + \sum mask \cdot mask\_weights` of one render's G-buffers
    against fixed weights (what gradient checks and benchmarks of a renderer back-propagate).

    Not a reference operator: in torch it is ``(image * image_weights).sum() + (mask * mask_weights).sum()`` -- two
    reductions, an add and two full-size products backward.  On the GPU both sums are one fused pass forward and both
    gradients one pass backward (kaolin_amd/csrc/render_metrics.hip); other inputs take the torch formulation.

    Args:
        image, image_weights (torch.Tensor): same shape and dtype.
        mask, mask_weights (torch.Tensor, optional): same shape and dtype as each other.

    Returns:
        (torch.Tensor): scalar.
    """
    if (mask is None) != (mask_weights is None):
        raise ValueError('weighted_sum expects mask and mask_weights together')
    pair2 = mask is not None
    if (_fusable(image, image_weights) and image.numel() > 0 and
            (not pair2 or (_fusable(mask, mask_weights) and mask.dtype == image.dtype and mask.device == image.device))):
        return _WeightedSum2Cuda.apply(image, image_weights, mask, mask_weights)
    out = (image * image_weights).sum()
    if pair2:
        out = out + (mask * mask_weights).sum()
    return out
