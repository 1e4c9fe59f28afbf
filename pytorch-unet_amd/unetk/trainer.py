"""The reference's inner training step (train.py:77-81, 97-117) on the MI355X kernels, data-parallel.

  images -> channels_last fp32 ; autocast forward ; CE/BCE + dice ; zero_grad(set_to_none) ; backward
  (gradient buckets all-reduced over RCCL while backward runs) ; clip_grad_norm_(1.0) ; RMSprop(lr 1e-5,
  weight_decay 1e-8, momentum 0.999, foreach) step.

amp_dtype: torch.bfloat16 (default: BASELINE.json's configurations; its range needs no loss scaling, so no scaler
runs) or torch.float16 -- what the reference's `--amp` gives on a GPU (train.py:100 leaves autocast's dtype at torch's
default) -- with the reference's GradScaler sequence (train.py:80,114-117): scale(loss).backward(), unscale_,
clip, step (skipped when a gradient overflowed), update.  Clip + RMSprop run as one fused pass over the flat gradient /
parameter / state buffers (unetk.optim.ClipRMSprop); `fused_optimizer=False` keeps torch's
foreach RMSprop with the clip applied to the flat gradient buffer (same arithmetic as
torch.nn.utils.clip_grad_norm_: coef = max_norm / (total_norm + 1e-6), clamped to 1).
"""
import torch

from .ddp import FlatGrads
from .loss import segmentation_loss


class Trainer:
    def __init__(self, net, learning_rate=1e-5, weight_decay=1e-8, momentum=0.999, gradient_clipping=1.0, amp=True,
                 process_group=None, bucket_bytes=25 << 20, fused_optimizer=True, force_collectives=False,
                 amp_dtype=torch.bfloat16):
        if amp_dtype not in (torch.bfloat16, torch.float16):
            raise ValueError('amp_dtype must be torch.bfloat16 or torch.float16')
        self.net = net
        self.amp = amp
        self.amp_dtype = amp_dtype
        # train.py:80 -- enabled only where it is needed: fp16's 5 exponent bits
        self.grad_scaler = torch.amp.GradScaler('cuda', enabled=bool(amp) and amp_dtype == torch.float16)
        self.clip = gradient_clipping
        self.sync = FlatGrads(net, process_group, bucket_bytes, force_collectives, grads_to_autograd=False)
        self.fused = fused_optimizer
        if fused_optimizer:
            from .optim import ClipRMSprop
            self.optimizer = ClipRMSprop(self.sync, lr=learning_rate, weight_decay=weight_decay, momentum=momentum,
                                         max_norm=gradient_clipping)
        else:
            self.optimizer = torch.optim.RMSprop(net.parameters(), lr=learning_rate, weight_decay=weight_decay,
                                                 momentum=momentum, foreach=True)

    def step(self, images, true_masks):
        """One optimisation step; returns the (device) loss tensor without forcing a host sync."""
        net = self.net
        images = images.to(dtype=torch.float32, memory_format=torch.channels_last)
        with torch.autocast('cuda', dtype=self.amp_dtype, enabled=self.amp):
            masks_pred = net(images)
            loss = segmentation_loss(masks_pred, true_masks)
        self.optimizer.zero_grad(set_to_none=True)
        scaler = self.grad_scaler
        scaler.scale(loss).backward()                    # identity when the scaler is disabled
        self.sync.finish()
        flat = self.sync.flat
        if scaler.is_enabled():
            # the (rank-averaged) flat buffer holds SCALED gradients: GradScaler unscales and inf-checks what the
            # optimizer's parameters carry in .grad, so hand it the flat views; an overflow on any rank reached every
            # rank through the all-reduce, so all ranks skip the same steps
            self.sync.bind_param_grads()
            scaler.unscale_(self.optimizer)
        if not self.fused and flat is not None:
            if self.clip is not None:
                total_norm = torch.linalg.vector_norm(flat)
                flat.mul_(torch.clamp(self.clip / (total_norm + 1e-6), max=1.0))
            # the flat buffer (averaged over ranks, clipped) is authoritative: hand its views to torch's optimizer
            self.sync.bind_param_grads()
        scaler.step(self.optimizer)                      # optimizer.step(), unless a gradient was inf/nan
        scaler.update()
        self.sync.consume()
        return loss.detach()
