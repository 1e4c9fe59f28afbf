"""torch.autograd glue: one generic Function that runs a block's composite forward/backward (unetk.blocks)."""
import torch

from . import ops


class BlockFn(torch.autograd.Function):
    """apply(impl, n_inputs, *inputs, *params) -> output (logical NCHW, channels_last memory).

    `impl.fwd(*nhwc_inputs) -> (out_nhwc, saved)`; `impl.bwd(saved, dout_nhwc, grads, needs) -> tuple of input grads`
    where `grads` is a dict param -> fp32 gradient filled by the kernels."""

    @staticmethod
    def forward(ctx, impl, n_inputs, *tensors):
        inputs, params = tensors[:n_inputs], tensors[n_inputs:]
        from . import blocks
        prev, blocks.SAVING_FOR_BACKWARD = blocks.SAVING_FOR_BACKWARD, True
        try:
            out, saved = impl.fwd(*[ops.to_nhwc(t.detach()) for t in inputs])
        finally:
            blocks.SAVING_FOR_BACKWARD = prev
        ctx.impl, ctx.saved, ctx.params, ctx.n_inputs = impl, saved, params, n_inputs
        ctx.in_dtypes = [t.dtype for t in inputs]
        ctx.out_dtype = out.dtype
        return ops.to_nchw(out)

    @staticmethod
    def backward(ctx, gout):
        g = ops.to_nhwc(gout)
        if g.dtype != ctx.out_dtype:
            g = g.to(ctx.out_dtype)
        grads = {}
        needs = ctx.needs_input_grad[2:2 + ctx.n_inputs]
        dxs = ctx.impl.bwd(ctx.saved, g, grads, needs)
        outs = []
        for dx, need, dt in zip(dxs, needs, ctx.in_dtypes):
            if dx is None or not need:
                outs.append(None)
            else:
                dx = ops.to_nchw(dx)
                outs.append(dx if dx.dtype == dt else dx.to(dt))
        pg = []
        # (a trainer that steps from the flat gradient buffer takes the gradients there: unetk.ddp.FlatGrads)
        deliver = getattr(getattr(ctx.impl, 'net', None), '_grads_via_autograd', True)
        for p, need in zip(ctx.params, ctx.needs_input_grad[2 + ctx.n_inputs:]):
            gp = grads.get(p) if (need and deliver) else None
            if gp is not None and gp.dtype != p.dtype:
                gp = gp.to(p.dtype)
            pg.append(gp)
        ctx.saved = None
        return (None, None, *outs, *pg)


def run_block(impl, inputs, params):
    """Runs impl through autograd when a gradient is wanted, else straight through."""
    for t in inputs:
        if not t.is_cuda:
            raise RuntimeError('the MI355X U-Net runs on the GPU only: move the input with .to("cuda") '
                               '(there is no CPU fallback path)')
    want = torch.is_grad_enabled() and (any(t.requires_grad for t in inputs) or any(p.requires_grad for p in params))
    if want:
        return BlockFn.apply(impl, len(inputs), *inputs, *params)
    out, _ = impl.fwd(*[ops.to_nhwc(t) for t in inputs])
    return ops.to_nchw(out)
