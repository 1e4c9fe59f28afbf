"""clip_grad_norm_ + RMSprop as one fused pass over flat buffers.

The reference builds `optim.RMSprop(model.parameters(), lr, weight_decay=1e-8, momentum=0.999, foreach=True)`
(train.py:77-78) and each step runs `clip_grad_norm_(model.parameters(), 1.0)` then `optimizer.step()`
(train.py:115-116): about ten foreach passes over 31 M parameters.  Here the parameters are re-pointed (same
nn.Parameter objects, same state_dict) at views of ONE flat fp32 buffer laid out exactly like the flat gradient
buffer of unetk.ddp.FlatGrads, the optimizer state is two more flat buffers, and a step is two kernels
(include/unetk.h: unetk_clip_rmsprop_step).  Subclasses torch.optim.Optimizer so LR schedulers
(train.py:79 ReduceLROnPlateau) keep working through param_groups[0]['lr'].
"""
import torch

from . import _native as nat
from ._native import check, fn, stream_ptr


class ClipRMSprop(torch.optim.Optimizer):
    def __init__(self, flat_grads, lr=1e-2, alpha=0.99, eps=1e-8, weight_decay=0.0, momentum=0.0, max_norm=None):
        self.fg = flat_grads
        super().__init__(flat_grads.params, dict(lr=lr, alpha=alpha, eps=eps, weight_decay=weight_decay,
                                                 momentum=momentum, max_norm=max_norm))
        self.pflat = self.square_avg = self.momentum_buf = None
        self.total_norm = None          # device fp64 scalar: ||grad||_2 before clipping (clip_grad_norm_'s result)
        self._flatten()

    def _flatten(self):
        """(Re-)point every parameter at its slice of one flat buffer; optimizer state survives a re-flatten."""
        fg = self.fg
        dev = fg.params[0].device
        if dev.type != 'cuda':
            raise RuntimeError('ClipRMSprop runs on the MI355X kernels: parameters must be on a cuda device')
        pflat = torch.zeros(fg.total, dtype=torch.float32, device=dev)
        for p in fg.params:
            if p.dtype != torch.float32:
                raise RuntimeError('ClipRMSprop: parameters must be fp32 (the reference keeps fp32 master weights)')
            off = fg.offsets[id(p)]
            view = pflat[off:off + p.numel()].view(p.shape)
            view.copy_(p.data)
            p.data = view
        self.pflat = pflat
        if self.square_avg is None:
            self.square_avg = torch.zeros_like(pflat)
            self.momentum_buf = torch.zeros_like(pflat)
            self._ws = torch.empty(int(fn['unetk_clip_rmsprop_workspace_bytes']()), dtype=torch.uint8, device=dev)
            self.total_norm = torch.zeros((), dtype=torch.float64, device=dev)

    def _still_flat(self):
        base = self.pflat.data_ptr()
        return all(p.data_ptr() == base + 4 * self.fg.offsets[id(p)] for p in self.fg.params)

    @torch.no_grad()
    def step(self, closure=None):
        if closure is not None:
            raise RuntimeError('ClipRMSprop.step takes no closure')
        grads = self.fg.flat
        if grads is None:
            raise RuntimeError('ClipRMSprop.step: no gradients (run backward through the UNet first)')
        if not self._still_flat():           # e.g. net.to(memory_format=...) after construction re-allocated them
            self._flatten()
        g = self.param_groups[0]
        max_norm = g['max_norm'] if g['max_norm'] is not None else 0.0
        check(fn['unetk_clip_rmsprop_step'](self.pflat.data_ptr(), grads.data_ptr(), self.square_avg.data_ptr(),
                                            self.momentum_buf.data_ptr(), self.fg.total, self._ws.data_ptr(),
                                            self._ws.numel(), self.total_norm.data_ptr(), float(max_norm),
                                            float(g['lr']), float(g['alpha']), float(g['eps']),
                                            float(g['weight_decay']), float(g['momentum']), stream_ptr()),
              'unetk_clip_rmsprop_step')
        # the kernel wrote the parameters behind PyTorch's back: bump their version counters like any in-place op
        # would, so autograd's checks and the packed-weight cache (blocks.packed_weight) see the update
        torch.autograd.graph.increment_version(self.fg.params)
        return None
