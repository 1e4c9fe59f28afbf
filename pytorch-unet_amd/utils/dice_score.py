"""Dice coefficient / loss with the reference's signatures (utils/dice_score.py:5-28) on MI355X kernels.

`dice_coeff` sums over the last two dims (or the last three when the input is 3-D and reduce_batch_first),
replaces an all-empty denominator by the numerator (=> dice 1.0) and returns the mean over the remaining
entries.  One reduction kernel produces the per-group sums; the backward is one elementwise kernel.
"""
import torch
from torch import Tensor

from unetk import ops


class _DiceFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, inp, tgt, n_groups, group_size, eps):
        x = inp.detach().float().contiguous()
        t = tgt.detach().float().contiguous()
        out, coef = ops.dice_fwd(x, t, n_groups, group_size, eps, True)
        ctx.t, ctx.coef, ctx.geo, ctx.in_shape, ctx.in_dtype = t, coef, (n_groups, group_size), inp.shape, inp.dtype
        return out.reshape(())

    @staticmethod
    def backward(ctx, g):
        n_groups, group_size = ctx.geo
        go = g.detach().float().contiguous()
        dx = ops.dice_bwd(ctx.t, ctx.coef, go, 1.0, n_groups, group_size).view(ctx.in_shape)
        return dx.to(ctx.in_dtype), None, None, None, None


def dice_coeff(input: Tensor, target: Tensor, reduce_batch_first: bool = False, epsilon: float = 1e-6):
    # one Dice value per reduction group (a mask, or a whole batch of masks), averaged
    assert input.size() == target.size()
    assert input.dim() == 3 or not reduce_batch_first
    if not input.is_cuda:
        raise RuntimeError('dice_coeff runs on the GPU only (no CPU fallback path)')
    n_red = 2 if input.dim() == 2 or not reduce_batch_first else 3
    group_size = 1
    for s in input.shape[input.dim() - n_red:]:
        group_size *= int(s)
    n_groups = max(input.numel() // max(group_size, 1), 1)
    if input.numel() == 0:
        raise RuntimeError('dice_coeff of an empty tensor')
    if torch.is_grad_enabled() and input.requires_grad:
        return _DiceFn.apply(input, target, n_groups, group_size, float(epsilon))
    out, _ = ops.dice_fwd(input.float().contiguous(), target.float().contiguous(), n_groups, group_size,
                          float(epsilon), False)
    return out.reshape(())


def multiclass_dice_coeff(input: Tensor, target: Tensor, reduce_batch_first: bool = False, epsilon: float = 1e-6):
    # classes fold into the batch dimension: every (image, class) plane is its own group
    return dice_coeff(input.flatten(0, 1), target.flatten(0, 1), reduce_batch_first, epsilon)


def dice_loss(input: Tensor, target: Tensor, multiclass: bool = False):
    # 1 - Dice: 0 for a perfect overlap
    fn = multiclass_dice_coeff if multiclass else dice_coeff
    return 1 - fn(input, target, reduce_batch_first=True)
