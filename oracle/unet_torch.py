"""TEST INFRASTRUCTURE ONLY -- stock-PyTorch restatement of the reference U-Net.

This is the arithmetic the reference runs (every FLOP there is an ATen op; SURVEY.md
section 8c "Where the arithmetic actually lives"), composed from stock `torch.nn` layers in
the same order so that `state_dict()` keys match row S of SURVEY 8a.  It is used

  * as the CPU checker for whole-network parity on the GPU box (the reference's own files
    never travel there), and
  * as `cpu_baseline` (kind "port") in bench.py.

Pinned against the imported reference by tests/test_oracle_golden.py (same weights, same
input -> identical logits/grads as the goldens in tests/golden/).  Never imported by the
product package.

Reference lines restated:
  block()      unet/unet_parts.py:15-22   conv3x3(pad 1, no bias) -> BN -> ReLU, twice
  Encoder step unet/unet_parts.py:33-36   MaxPool2d(2) then block
  Decoder step unet/unet_parts.py:49-68   upsample (bilinear AC | convT k2 s2), pad, cat([skip, up]), block
  Head         unet/unet_parts.py:74      conv1x1 with bias
  Net wiring   unet/unet_model.py:13-36
  Loss         train.py:81,102-111 + utils/dice_score.py:5-28
"""
import torch
import torch.nn as nn
import torch.nn.functional as F


def _block(cin, cout, cmid=None):
    cmid = cmid or cout
    holder = nn.Module()
    holder.double_conv = nn.Sequential(
        nn.Conv2d(cin, cmid, 3, padding=1, bias=False), nn.BatchNorm2d(cmid), nn.ReLU(inplace=True),
        nn.Conv2d(cmid, cout, 3, padding=1, bias=False), nn.BatchNorm2d(cout), nn.ReLU(inplace=True))
    return holder


class _Enc(nn.Module):
    def __init__(self, cin, cout):
        super().__init__()
        self.maxpool_conv = nn.Sequential(nn.MaxPool2d(2), _block(cin, cout))
        # make `maxpool_conv.1.double_conv.*` keys: _block returns a holder with .double_conv

    def forward(self, x):
        return self.maxpool_conv[1].double_conv(self.maxpool_conv[0](x))


class _Dec(nn.Module):
    def __init__(self, cin, cout, bilinear):
        super().__init__()
        if bilinear:
            self.up = nn.Upsample(scale_factor=2, mode='bilinear', align_corners=True)
            self.conv = _block(cin, cout, cin // 2)
        else:
            self.up = nn.ConvTranspose2d(cin, cin // 2, kernel_size=2, stride=2)
            self.conv = _block(cin, cout)

    def forward(self, low, skip):
        low = self.up(low)
        dy = skip.shape[2] - low.shape[2]
        dx = skip.shape[3] - low.shape[3]
        low = F.pad(low, [dx // 2, dx - dx // 2, dy // 2, dy - dy // 2])
        return self.conv.double_conv(torch.cat([skip, low], dim=1))


class TorchUNet(nn.Module):
    """Same constructor, attributes and state_dict keys as the reference UNet."""

    def __init__(self, n_channels, n_classes, bilinear=False):
        super().__init__()
        self.n_channels, self.n_classes, self.bilinear = n_channels, n_classes, bilinear
        f = 2 if bilinear else 1
        self.inc = _block(n_channels, 64)
        self.down1 = _Enc(64, 128)
        self.down2 = _Enc(128, 256)
        self.down3 = _Enc(256, 512)
        self.down4 = _Enc(512, 1024 // f)
        self.up1 = _Dec(1024, 512 // f, bilinear)
        self.up2 = _Dec(512, 256 // f, bilinear)
        self.up3 = _Dec(256, 128 // f, bilinear)
        self.up4 = _Dec(128, 64, bilinear)
        self.outc = nn.Module()
        self.outc.conv = nn.Conv2d(64, n_classes, kernel_size=1)

    def forward(self, x):
        s1 = self.inc.double_conv(x)
        s2 = self.down1(s1)
        s3 = self.down2(s2)
        s4 = self.down3(s3)
        b = self.down4(s4)
        y = self.up1(b, s4)
        y = self.up2(y, s3)
        y = self.up3(y, s2)
        y = self.up4(y, s1)
        return self.outc.conv(y)


def dice_coeff(inp, tgt, reduce_batch_first=False, epsilon=1e-6):
    """utils/dice_score.py:5-17."""
    assert inp.size() == tgt.size()
    assert inp.dim() == 3 or not reduce_batch_first
    dims = (-1, -2) if inp.dim() == 2 or not reduce_batch_first else (-1, -2, -3)
    inter = 2 * (inp * tgt).sum(dim=dims)
    sets = inp.sum(dim=dims) + tgt.sum(dim=dims)
    sets = torch.where(sets == 0, inter, sets)
    return ((inter + epsilon) / (sets + epsilon)).mean()


def multiclass_dice_coeff(inp, tgt, reduce_batch_first=False, epsilon=1e-6):
    """utils/dice_score.py:20-22."""
    return dice_coeff(inp.flatten(0, 1), tgt.flatten(0, 1), reduce_batch_first, epsilon)


def dice_loss(inp, tgt, multiclass=False):
    """utils/dice_score.py:25-28."""
    fn = multiclass_dice_coeff if multiclass else dice_coeff
    return 1 - fn(inp, tgt, reduce_batch_first=True)


def segmentation_loss(logits, mask, n_classes):
    """train.py:81,102-111: CE (or BCE-with-logits) + dice loss."""
    if n_classes == 1:
        z = logits.squeeze(1)
        t = mask.float()
        return F.binary_cross_entropy_with_logits(z, t) + dice_loss(torch.sigmoid(z), t, multiclass=False)
    loss = F.cross_entropy(logits, mask)
    return loss + dice_loss(F.softmax(logits, dim=1).float(),
                            F.one_hot(mask, n_classes).permute(0, 3, 1, 2).float(), multiclass=True)


def train_step(model, opt, images, masks, clip=1.0):
    """train.py:97-117 on CPU, no AMP: forward, loss, zero_grad, backward, clip, RMSprop step."""
    logits = model(images)
    loss = segmentation_loss(logits, masks, model.n_classes)
    opt.zero_grad(set_to_none=True)
    loss.backward()
    torch.nn.utils.clip_grad_norm_(model.parameters(), clip)
    opt.step()
    return loss
