"""UNet assembly with the reference's surface (unet/unet_model.py:6-48) on MI355X kernels.

forward() runs the whole network as ONE autograd node (unetk.blocks composites): the four skip tensors are
produced directly inside the decoder's concat buffers (no torch.cat copy, unet_parts.py:67), the gradient each
skip receives from the decoder is folded into the max-pool backward kernel instead of a separate accumulation
pass, and parameter gradients become available block by block in reverse order so a data-parallel trainer can
bucket-all-reduce them while the rest of backward still runs (unetk.ddp).
"""
import torch
import torch.nn as nn

from unetk import blocks, ops
from unetk.autograd import BlockFn
from .unet_parts import DoubleConv, Down, Up, OutConv


class _UNetImpl:
    def __init__(self, net, dtype):
        self.net, self.dtype = net, dtype
        self.will_backward = False

    # ---- forward
    def fwd(self, x):
        net, dt = self.net, self.dtype
        n, h, w, _, _ = ops.nhwc_meta(x, 'UNet input')
        downs = [net.down1, net.down2, net.down3, net.down4]
        ups = [net.up1, net.up2, net.up3, net.up4]
        enc = [net.inc] + [d.maxpool_conv[1] for d in downs]
        saved = {'enc': [], 'pool_in': [], 'up': [], 'dec': [], 'cats': []}
        blocks.pack_all(net, dt, self.will_backward)         # one launch re-packs every conv weight that changed
        bn_channels = sum(m.num_features for m in net.modules() if isinstance(m, nn.BatchNorm2d))
        ops.begin_stats_arena(x.device, bn_channels)       # one memset for all BatchNorm statistics of this pass
        # encoder: level k's output is written straight into the first half of that level's concat buffer
        cur, hh, ww = x, h, w
        skips = []
        for k, dc in enumerate(enc):
            c_out = dc.double_conv[3].out_channels
            if k < 4:
                c_up = ups[3 - k].conv.double_conv[0].in_channels - c_out   # what the decoder's first conv expects
                cat = ops.new_nhwc(n, hh, ww, c_out + c_up, dt, x.device)
                out_view = cat[..., :c_out]
            else:
                cat, out_view = None, None
            pooled = None
            if k < 4:
                if hh < 2 or ww < 2:
                    raise RuntimeError('UNet: input too small for four 2x2 poolings')
                pooled = ops.new_nhwc(n, hh // 2, ww // 2, c_out, dt, x.device)   # Down's MaxPool2d, fused into BN+ReLU
            a, s = blocks.double_conv_fwd(dc, cur, dt, out=out_view, pool_out=pooled)
            saved['enc'].append(s)
            if k < 4:
                skips.append((cat, c_out))
                saved['pool_in'].append(a)
                cur = pooled
                hh, ww = hh // 2, ww // 2
            else:
                cur = a
        # decoder
        for k, up in enumerate(ups):
            cat, c_skip = skips[3 - k]
            saved['up'].append(blocks.up_into_cat(up.up, cur, cat, c_skip, dt))
            # the last block's BatchNorm+ReLU kernel also emits OutConv's logits where it can
            head = {'conv': net.outc.conv, 'logits_only': True} if k == 3 else None
            cur, s = blocks.double_conv_fwd(up.conv, cat, dt, head=head)
            saved['dec'].append(s)
            saved['cats'].append(c_skip)
        if head is not None and 'logits' in head:
            logits, hs = head['logits'], head['saved']
        else:
            logits, hs = blocks.head_fwd(net.outc.conv, cur, dt)
        saved['head'] = hs
        return logits, saved

    # ---- backward
    def bwd(self, saved, dlogits, grads, needs):
        net = self.net
        downs = [net.down1, net.down2, net.down3, net.down4]
        ups = [net.up1, net.up2, net.up3, net.up4]
        enc = [net.inc] + [d.maxpool_conv[1] for d in downs]
        ready = getattr(net, '_grad_ready_hook', None)
        init = getattr(net, '_initial_grads', None)
        if init is not None:
            grads.update(init())          # gradients land in one flat, bucket-ordered buffer (unetk.ddp)
        # per-channel accumulators of the 18 BatchNorm backward passes + OutConv's gradient replicas (<= 4 x 65 values)
        ops.begin_stats_arena(dlogits.device,
                              sum(m.num_features for m in net.modules() if isinstance(m, nn.BatchNorm2d)) + 160 +
                              sum(u.conv.double_conv[0].in_channels for u in ups))

        def done(params):
            if ready is not None:
                ready([p for p in params if p is not None], grads)

        # a layer's weight gradient is queued behind the NEXT layer's BatchNorm-backward reduction (blocks.conv_bn_relu_bwd)
        blocks.DEFER_WGRAD = blocks.DEFER_WGRAD_ENABLED
        try:
            out = self._bwd_body(saved, dlogits, grads, needs, net, ups, enc, done)
            ops.run_deferred()               # the last layer's
            return out
        finally:
            blocks.DEFER_WGRAD = False
            ops.drop_deferred()              # (only non-empty after an exception)

    def _bwd_body(self, saved, dlogits, grads, needs, net, ups, enc, done):
        # OutConv's input gradient is formed inside the BatchNorm backward of up4's last layer where it can be
        d, dhead = blocks.head_bwd(net.outc.conv, saved['head'], dlogits, grads, True, fuse_dx=True)
        if dhead is None:
            done([net.outc.conv.weight, net.outc.conv.bias])
        dcats = [None] * 4
        sums2 = None
        for k in (3, 2, 1, 0):
            up = ups[k]
            # ConvTranspose2d's bias gradient = channel sums of the concat gradient's second half: accumulated by the
            # epilogue of the dgrad that writes it
            sums = None
            if isinstance(up.up, nn.ConvTranspose2d):
                sums = ops.new_stats(up.conv.double_conv[0].in_channels, dlogits.device)
            # OutConv's gradients come out of up4's BatchNorm backward (dhead): reported with that layer
            first = [net.outc.conv.weight, net.outc.conv.bias] if (k == 3 and dhead is not None) else None
            dcat = blocks.double_conv_bwd(up.conv, saved['dec'][k], d, grads, True, dhead=dhead if k == 3 else None,
                                          dx_sums=sums, done=done, done_first=first, sums2=sums2)
            dcats[k] = dcat
            # the block's low-res input is the activation of the previous block's (k = 0: down4's) last layer: the
            # ConvTranspose2d dgrad that writes its gradient also forms that layer's BN-backward sums where it can
            below = saved['dec'][k - 1][1] if k > 0 else saved['enc'][4][1]          # (x, z, BNState)
            d, sums2 = blocks.up_bwd(up.up, saved['up'][k], dcat, saved['cats'][k], grads, True, dcat_sums=sums,
                                     done=done, bn_prev=(below[1], below[2]))
        # encoder, deepest first.  A block's output fed the skip connection and the next block's max-pool: its
        # gradient = first half of dcat + scatter(pool gradient), formed inside the BatchNorm backward kernels
        dpool = None
        for k in (4, 3, 2, 1, 0):
            need_dx = k > 0 or needs[0]
            d = blocks.double_conv_bwd(enc[k], saved['enc'][k], d, grads, need_dx, dpool=dpool, done=done,
                                       sums2=sums2 if k == 4 else None)
            dpool = None
            if k > 0:
                dcat = dcats[4 - k]
                c_skip = saved['cats'][4 - k]
                dskip = dcat[..., :c_skip]
                if ops.pool_bwd_fusable(c_skip, dskip.dtype):
                    d, dpool = dskip, d
                else:
                    d = ops.maxpool_bwd(saved['pool_in'][k - 1], d, dskip, out=dskip)
        return (d,)


class UNet(nn.Module):
    """reference unet_model.py:6-36: 5 encoder stages 64..1024/factor, 4 decoder stages, 1x1 head."""

    def __init__(self, n_channels, n_classes, bilinear=False):
        super().__init__()
        self.n_channels = n_channels
        self.n_classes = n_classes
        self.bilinear = bilinear
        factor = 2 if bilinear else 1
        self.inc = DoubleConv(n_channels, 64)
        self.down1 = Down(64, 128)
        self.down2 = Down(128, 256)
        self.down3 = Down(256, 512)
        self.down4 = Down(512, 1024 // factor)
        self.up1 = Up(1024, 512 // factor, bilinear)
        self.up2 = Up(512, 256 // factor, bilinear)
        self.up3 = Up(256, 128 // factor, bilinear)
        self.up4 = Up(128, 64, bilinear)
        self.outc = OutConv(64, n_classes)

    def hot_path_params(self):
        """Parameters in the order the whole-network autograd node takes them."""
        ps = blocks.double_conv_params(self.inc)
        for d in (self.down1, self.down2, self.down3, self.down4):
            ps += blocks.double_conv_params(d.maxpool_conv[1])
        for u in (self.up1, self.up2, self.up3, self.up4):
            ps += blocks.up_params(u)
        return ps + [self.outc.conv.weight, self.outc.conv.bias]

    def forward(self, x):
        if not x.is_cuda:
            raise RuntimeError('the MI355X U-Net runs on the GPU only: move the input with .to("cuda") '
                               '(there is no CPU fallback path)')
        if x.dim() != 4 or x.shape[1] != self.n_channels:
            raise RuntimeError(f'expected input [N, {self.n_channels}, H, W], got {tuple(x.shape)}')
        impl = _UNetImpl(self, blocks.activation_dtype(x))
        params = self.hot_path_params()
        if torch.is_grad_enabled() and (x.requires_grad or any(p.requires_grad for p in params)):
            impl.will_backward = True          # also pack the dgrad weight streams in the forward's single launch
            return BlockFn.apply(impl, 1, x, *params)
        out, _ = impl.fwd(ops.to_nhwc(x))
        return ops.to_nchw(out)

    def use_checkpointing(self):
        """reference unet_model.py:38-48 calls the *module* torch.utils.checkpoint and raises TypeError
        (SURVEY.md 8a row A9), so its OOM fallback can never succeed.  Activation memory here is sized for
        288 GB of HBM, so checkpointing is not needed; the method is kept for API compatibility
        and fails the same way the reference does."""
        raise TypeError("'module' object is not callable")
