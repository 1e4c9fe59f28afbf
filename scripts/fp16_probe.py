"""Measures what the fp16 path delivers on the separated-logits fixture and a literal GradScaler loop (used once to set
the bars of tests/test_gpu_trainstep.py's fp16 tests)."""
import os
import sys

import numpy as np
import torch
import torch.nn.functional as F

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, 'pytorch-unet_amd'), os.path.join(ROOT, 'tests')]
from conftest import golden, l2_err, rel_err  # noqa: E402
from oracle import weights as W  # noqa: E402
from test_gpu_trainstep import _closed_form, _img, npy, TAG  # noqa: E402
from unet import UNet  # noqa: E402

g = golden(TAG)
img = _img()
for dt in (torch.bfloat16, torch.float16):
    net = _closed_form(UNet(3, 2, False), g['bias_train']).cuda().train()
    with torch.no_grad(), torch.autocast('cuda', dtype=dt):
        z = net(img)
    print(dt, 'train', 'agree', float((npy(z).argmax(1) == g['logits_train'].argmax(1)).mean()),
          'l2', l2_err(npy(z), g['logits_train']), 'rel', rel_err(npy(z), g['logits_train']))
    with torch.no_grad():
        net.outc.conv.bias.copy_(torch.from_numpy(g['bias_eval']).cuda())
    net.eval()
    with torch.no_grad(), torch.autocast('cuda', dtype=dt):
        e = net(img)
    print(dt, 'eval', 'agree', float((npy(e).argmax(1) == g['logits_eval'].argmax(1)).mean()),
          'l2', l2_err(npy(e), g['logits_eval']), 'rel', rel_err(npy(e), g['logits_eval']))
