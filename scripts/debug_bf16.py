import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'pytorch-unet_amd')); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import numpy as np, torch
from unet import UNet
from test_gpu_unet import load_closed_form, wave, cl, npy
from conftest import rel_err, golden, UNET_CASES
tag, ncls, bil, shape = UNET_CASES[0]
g = golden(tag)
for mode in ('eval', 'train'):
    net = load_closed_form(UNet(3, ncls, bil)).cuda()
    net.train(mode == 'train')
    img = cl(wave(tag + '.img', shape, 0.5, 0.5))
    outs = {}
    for dt in ('fp32', 'bf16'):
        net2 = load_closed_form(UNet(3, ncls, bil)).cuda(); net2.train(mode == 'train')
        with torch.no_grad(), torch.autocast('cuda', dtype=torch.bfloat16, enabled=(dt == 'bf16')):
            x1 = net2.inc(img); x2 = net2.down1(x1); x3 = net2.down2(x2); x4 = net2.down3(x3); x5 = net2.down4(x4)
            u1 = net2.up1(x5, x4); u2 = net2.up2(u1, x3); u3 = net2.up3(u2, x2); u4 = net2.up4(u3, x1); lo = net2.outc(u4)
            whole = net2(img) if mode == 'eval' else None
        outs[dt] = dict(x1=x1, x2=x2, x3=x3, x4=x4, x5=x5, u1=u1, u2=u2, u3=u3, u4=u4, lo=lo, whole=whole)
    for k in outs['fp32']:
        if outs['fp32'][k] is None: continue
        print(mode, k, 'bf16 vs fp32: %.4g' % rel_err(npy(outs['bf16'][k]), npy(outs['fp32'][k])), 'max %.3g' % float(outs['fp32'][k].abs().max()))
    print(mode, 'fp32 modular logits vs golden', rel_err(npy(outs['fp32']['lo']), g['logits_eval' if mode == 'eval' else 'logits_train']))
    if mode == 'eval':
        print('fp32 whole vs golden', rel_err(npy(outs['fp32']['whole']), g['logits_eval']), 'bf16 whole vs golden', rel_err(npy(outs['bf16']['whole']), g['logits_eval']))
