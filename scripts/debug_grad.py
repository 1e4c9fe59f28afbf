import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, 'pytorch-unet_amd'), os.path.join(ROOT, 'tests')]
import numpy as np, torch
from conftest import golden, l2_err, UNET_CASES
import test_gpu_unet as T
from unet import UNet
from unetk.loss import segmentation_loss
for tag, ncls, bil, shape in UNET_CASES:
    g = golden(tag)
    names = g['grad_names'].tolist()
    n32 = dict(zip(names, g['grad_norms'].tolist())); n64 = dict(zip(names, g['grad_norms64'].tolist()))
    for rep in range(2):
        net = T.load_closed_form(UNet(3, ncls, bil)).cuda().train()
        img = T.cl(T.wave(tag + '.img', shape, 0.5, 0.5))
        mask = torch.from_numpy(T.W.int_wave(tag + '.mask', (shape[0],) + shape[2:], max(ncls, 2))).cuda()
        loss = segmentation_loss(net(img), mask); loss.backward()
        params = dict(net.named_parameters())
        errs = sorted(((l2_err(T.npy(params[k[7:]].grad), g[k]), k[7:]) for k in g.files if k.startswith('grad64.')), reverse=True)
        print(tag, rep, [(round(e, 5), k, round(abs(n32[k] - n64[k]) / n64[k], 5)) for e, k in errs[:4]])
