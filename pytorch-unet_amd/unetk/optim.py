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
        self._steps = 0
        self._flatten()

    def zero_grad(self, set_to_none=True):
        """train.py:113.  The flat gradient buffer is rewritten by every backward pass, so there is nothing to zero;
        the call marks the previous pass's gradients as discarded (a backward without it or step() in between would be
        gradient accumulation, which the flat buffer does not do: FlatGrads.initial_grads raises)."""
        self.fg.consume()
        super().zero_grad(set_to_none=set_to_none)

    # ---- checkpoint / resume in torch.optim.RMSprop's own format (train.py saves only the model's state_dict, but a
    # user who checkpoints the optimizer can move between the two, in BOTH directions): per-parameter 'step',
    # 'square_avg', 'momentum_buffer', keyed by the parameter's index in net.parameters() order -- the order a stock
    # optimizer built from `net.parameters()` maps indices by (it checks group sizes only, not shapes)
    def state_dict(self):
        g = self.param_groups[0]
        order = list(self.fg.module_order)
        group = {k: v for k, v in g.items() if k != 'params'}
        group.update(centered=False, params=list(range(len(order))), unetk_order='module')
        state = {}
        if self._steps > 0:
            for i, p in enumerate(order):
                off, n = self.fg.offsets[id(p)], p.numel()
                st = {'step': torch.tensor(float(self._steps)),
                      'square_avg': self.square_avg[off:off + n].view(p.shape).clone()}
                if g['momentum'] > 0:
                    st['momentum_buffer'] = self.momentum_buf[off:off + n].view(p.shape).clone()
                state[i] = st
        return {'state': state, 'param_groups': [group]}

    def load_state_dict(self, state_dict):
        groups = state_dict['param_groups']
        if len(groups) != 1 or len(groups[0]['params']) != len(self.fg.params):
            raise ValueError('ClipRMSprop.load_state_dict: expected one parameter group over the same parameters')
        if groups[0].get('centered', False):
            raise ValueError('ClipRMSprop.load_state_dict: centered RMSprop state cannot be loaded')
        g = self.param_groups[0]
        for k in ('lr', 'alpha', 'eps', 'weight_decay', 'momentum', 'max_norm'):
            if k in groups[0]:
                g[k] = groups[0][k]
        self.square_avg.zero_()
        self.momentum_buf.zero_()
        steps = 0
        order = self._saved_order(groups[0])
        for i, st in state_dict['state'].items():
            p = order[int(i)]
            off, n = self.fg.offsets[id(p)], p.numel()
            if tuple(st['square_avg'].shape) != tuple(p.shape):
                raise ValueError('ClipRMSprop.load_state_dict: state shape does not match the parameter')
            self.square_avg[off:off + n].view(p.shape).copy_(st['square_avg'])
            if 'momentum_buffer' in st and st['momentum_buffer'] is not None:
                self.momentum_buf[off:off + n].view(p.shape).copy_(st['momentum_buffer'])
            steps = max(steps, int(float(st.get('step', 0))))
        self._steps = steps

    def _saved_order(self, group):
        """Checkpoints of a stock torch.optim.RMSprop and of this class (marker `unetk_order`) number the parameters in
        net.parameters() order; files this class wrote before it carried the marker (round 2: 'max_norm' present, no
        marker) used gradient-completion order and still load."""
        if 'max_norm' in group and group.get('unetk_order') != 'module':
            return self.fg.params
        return list(self.fg.module_order)

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
        self._steps += 1
        self.fg.consume()
        return None
