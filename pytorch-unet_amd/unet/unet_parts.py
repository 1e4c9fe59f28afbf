"""U-Net building blocks with the reference's module surface (unet/unet_parts.py:8-77) on MI355X kernels.

Class names, constructor signatures and sub-module attribute names (`double_conv`, `maxpool_conv`, `up`, `conv`)
are the reference's, so `state_dict()` keys, shapes and dtypes are identical and its checkpoints load with
strict=True.  The stock nn layers below are PARAMETER HOLDERS only: their own forward is never called.  Every
forward()/backward dispatches through the C ABI of libunetk.so (include/unetk.h) to hand-written gfx950 kernels;
there is no PyTorch/MIOpen fallback -- CPU tensors raise.
"""
import torch
import torch.nn as nn

from unetk import blocks, ops
from unetk.autograd import run_block


class _DoubleConvImpl:
    def __init__(self, mod, dtype):
        self.mod, self.dtype = mod, dtype

    def fwd(self, x):
        return blocks.double_conv_fwd(self.mod, x, self.dtype)

    def bwd(self, saved, dout, grads, needs):
        return (blocks.double_conv_bwd(self.mod, saved, dout, grads, needs[0]),)


class DoubleConv(nn.Module):
    """(conv3x3 => BatchNorm => ReLU) * 2   -- reference unet_parts.py:8-25"""

    def __init__(self, in_channels, out_channels, mid_channels=None):
        super().__init__()
        mid = mid_channels if mid_channels else out_channels
        layers = []
        for cin, cout in ((in_channels, mid), (mid, out_channels)):
            layers += [nn.Conv2d(cin, cout, kernel_size=3, padding=1, bias=False), nn.BatchNorm2d(cout),
                       nn.ReLU(inplace=True)]
        self.double_conv = nn.Sequential(*layers)

    def forward(self, x):
        impl = _DoubleConvImpl(self, blocks.activation_dtype(x))
        return run_block(impl, (x,), blocks.double_conv_params(self))


class _DownImpl:
    def __init__(self, mod, dtype):
        self.dc, self.dtype = mod.maxpool_conv[1], dtype

    def fwd(self, x):
        if x.dtype != self.dtype:
            x = x.to(self.dtype)
        pooled = ops.maxpool_fwd(x)
        out, saved = blocks.double_conv_fwd(self.dc, pooled, self.dtype)
        return out, (x, saved)

    def bwd(self, saved, dout, grads, needs):
        x, dc_saved = saved
        dpool = blocks.double_conv_bwd(self.dc, dc_saved, dout, grads, needs[0])
        return (ops.maxpool_bwd(x, dpool) if needs[0] else None,)


class Down(nn.Module):
    """MaxPool2d(2) then DoubleConv   -- reference unet_parts.py:28-39"""

    def __init__(self, in_channels, out_channels):
        super().__init__()
        self.maxpool_conv = nn.Sequential(nn.MaxPool2d(2), DoubleConv(in_channels, out_channels))

    def forward(self, x):
        impl = _DownImpl(self, blocks.activation_dtype(x))
        return run_block(impl, (x,), blocks.double_conv_params(self.maxpool_conv[1]))


class _UpImpl:
    def __init__(self, mod, dtype):
        self.mod, self.dtype = mod, dtype

    def fwd(self, x1, x2):
        dt = self.dtype
        if x2.dtype != dt:
            x2 = x2.to(dt)
        n, hs, ws, c_skip, _ = ops.nhwc_meta(x2, 'Up skip input')
        c_up = self.mod.up.out_channels if isinstance(self.mod.up, nn.ConvTranspose2d) else x1.shape[3]
        cat = ops.new_nhwc(n, hs, ws, c_skip + c_up, dt, x2.device)
        ops.copy_channels(x2, cat[..., :c_skip])                       # torch.cat([x2, x1], dim=1): skip first
        up_saved = blocks.up_into_cat(self.mod.up, x1, cat, c_skip, dt)
        out, dc_saved = blocks.double_conv_fwd(self.mod.conv, cat, dt)
        return out, (up_saved, dc_saved, c_skip)

    def bwd(self, saved, dout, grads, needs):
        up_saved, dc_saved, c_skip = saved
        dcat = blocks.double_conv_bwd(self.mod.conv, dc_saved, dout, grads, True)
        dx1 = blocks.up_bwd(self.mod.up, up_saved, dcat, c_skip, grads, needs[0])
        return dx1, (dcat[..., :c_skip] if needs[1] else None)


class Up(nn.Module):
    """Up-sample (bilinear align_corners=True | ConvTranspose2d k2 s2), pad to the skip size,
    cat([skip, up]) and DoubleConv   -- reference unet_parts.py:42-68"""

    def __init__(self, in_channels, out_channels, bilinear=True):
        super().__init__()
        if bilinear:
            self.up = nn.Upsample(scale_factor=2, mode='bilinear', align_corners=True)
            self.conv = DoubleConv(in_channels, out_channels, in_channels // 2)
        else:
            self.up = nn.ConvTranspose2d(in_channels, in_channels // 2, kernel_size=2, stride=2)
            self.conv = DoubleConv(in_channels, out_channels)

    def forward(self, x1, x2):
        impl = _UpImpl(self, blocks.activation_dtype(x2))
        return run_block(impl, (x1, x2), blocks.up_params(self))


class _OutConvImpl:
    def __init__(self, mod, dtype):
        self.conv, self.dtype = mod.conv, dtype

    def fwd(self, x):
        return blocks.head_fwd(self.conv, x, self.dtype)

    def bwd(self, saved, dout, grads, needs):
        return (blocks.head_bwd(self.conv, saved, dout, grads, needs[0])[0],)


class OutConv(nn.Module):
    """Conv2d k=1 with bias   -- reference unet_parts.py:71-77"""

    def __init__(self, in_channels, out_channels):
        super().__init__()
        self.conv = nn.Conv2d(in_channels, out_channels, kernel_size=1)

    def forward(self, x):
        impl = _OutConvImpl(self, blocks.activation_dtype(x))
        return run_block(impl, (x,), [self.conv.weight, self.conv.bias])
