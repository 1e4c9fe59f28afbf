"""Fused segmentation loss of the reference train step (train.py:81,102-111):
criterion(masks_pred, true_masks) + dice_loss(softmax|sigmoid(masks_pred), one_hot(true_masks)).
One reduction kernel forward, one elementwise kernel backward (csrc/loss.hip)."""
import torch

from . import ops


class _SegLossFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, logits, masks):
        z = ops.to_nhwc(logits.detach())
        m = masks.detach()
        if m.dtype != torch.int64:
            m = m.long()
        m = m.contiguous()
        out4 = ops.seg_loss_fwd(z, m)
        ctx.z, ctx.m, ctx.out4, ctx.dtype = z, m, out4, logits.dtype
        return out4[0].clone()

    @staticmethod
    def backward(ctx, g):
        go = g.detach().float().reshape(1).contiguous()
        dl = ops.seg_loss_bwd(ctx.z, ctx.m, ctx.out4, go)
        return ops.to_nchw(dl), None


def segmentation_loss(logits, masks):
    """logits [N,C,H,W] (bf16/fp32, any memory format), masks [N,H,W] integer class ids (0/1 for C == 1)."""
    if not logits.is_cuda:
        raise RuntimeError('segmentation_loss runs on the GPU only (no CPU fallback path)')
    if logits.dim() != 4 or masks.shape != (logits.shape[0], logits.shape[2], logits.shape[3]):
        raise RuntimeError(f'segmentation_loss: logits {tuple(logits.shape)} vs masks {tuple(masks.shape)}')
    return _SegLossFn.apply(logits, masks)
