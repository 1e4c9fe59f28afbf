"""torch.hub entry point with the reference's signature (hubconf.py:4-22)."""
import torch

from unet import UNet as _UNet


def unet_carvana(pretrained=False, scale=0.5):
    """UNet trained on the Carvana dataset (https://www.kaggle.com/c/carvana-image-masking-challenge/data).
    `scale` selects which released checkpoint to fetch when pretrained=True (0.5 or 1.0); the checkpoint's
    'mask_values' entry is dropped and the rest is loaded with the reference's key layout (strict)."""
    net = _UNet(n_channels=3, n_classes=2, bilinear=False)
    if pretrained:
        if scale == 0.5:
            checkpoint = 'https://github.com/milesial/Pytorch-UNet/releases/download/v3.0/unet_carvana_scale0.5_epoch2.pth'
        elif scale == 1.0:
            checkpoint = 'https://github.com/milesial/Pytorch-UNet/releases/download/v3.0/unet_carvana_scale1.0_epoch2.pth'
        else:
            raise RuntimeError('Only 0.5 and 1.0 scales are available')
        state_dict = torch.hub.load_state_dict_from_url(checkpoint, progress=True)
        state_dict.pop('mask_values', None)
        net.load_state_dict(state_dict)
    return net
