import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'pytorch-unet_amd')); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import numpy as np, torch
from oracle import ops_numpy as O, weights as W
from unetk import ops, blocks
from unet.unet_parts import DoubleConv
from test_gpu_unet import load_closed_form, wave, cl, npy
from conftest import rel_err

def nhwc(a): return np.ascontiguousarray(np.transpose(a, (0, 2, 3, 1))).astype(np.float64)
dc = load_closed_form(DoubleConv(64, 128), 'dc.').cuda().train()
x = cl(wave('dc.x', (2, 64, 16, 23)))
gy = cl(wave('dc.gy', (2, 128, 16, 23), 0.01))
xn = ops.to_nhwc(x); gn = ops.to_nhwc(gy)
a2, (s1, s2) = blocks.double_conv_fwd(dc, xn, torch.float32)
seq = dc.double_conv
w = lambda k, s: W.tensor_for_key('dc.' + k, s, 'float32').astype(np.float64)
x64 = nhwc(W.wave('dc.x', (2, 64, 16, 23))); g64 = nhwc(W.wave('dc.gy', (2, 128, 16, 23), 0.01))
w1 = w('double_conv.0.weight', (128, 64, 3, 3)); w2 = w('double_conv.3.weight', (128, 128, 3, 3))
g1, b1 = w('double_conv.1.weight', (128,)), w('double_conv.1.bias', (128,))
g2, b2 = w('double_conv.4.weight', (128,)), w('double_conv.4.bias', (128,))
y1 = O.conv3x3_fwd(x64, w1); z1, m1, v1, is1 = O.bn_train_fwd(y1, g1, b1); a1 = O.relu(z1)
y2 = O.conv3x3_fwd(a1, w2); z2, m2, v2, is2 = O.bn_train_fwd(y2, g2, b2); A2 = O.relu(z2)
print('y1', rel_err(s1[1].cpu().numpy(), y1), 'a1', rel_err(s2[0].cpu().numpy(), a1), 'y2', rel_err(s2[1].cpu().numpy(), y2), 'a2', rel_err(a2.cpu().numpy(), A2))
d2, dg2, db2 = O.bn_relu_train_bwd(y2, g2, m2, is2, A2, g64)
grads = {}
dy2 = ops.bn_relu_bwd(gn, s2[1], s2[2], seq[4].weight, torch.empty(128, device='cuda'), torch.empty(128, device='cuda'))
print('d2', rel_err(dy2.cpu().numpy(), d2))
da1_ref, dw2_ref = O.conv3x3_bwd(a1, w2, d2)
da1 = ops.conv3x3_dgrad(dy2, blocks.packed_weight(seq[3], 1, torch.float32, 128, 128), 128)
print('da1', rel_err(da1.cpu().numpy(), da1_ref))
dw2 = torch.empty(128, 128, 3, 3, device='cuda'); ops.conv3x3_wgrad(s2[0], dy2, dw2)
print('dw2', rel_err(dw2.cpu().numpy(), dw2_ref))
d1_ref, dg1, db1 = O.bn_relu_train_bwd(y1, g1, m1, is1, a1, da1_ref)
d1 = ops.bn_relu_bwd(da1, s1[1], s1[2], seq[1].weight, torch.empty(128, device='cuda'), torch.empty(128, device='cuda'))
print('d1', rel_err(d1.cpu().numpy(), d1_ref))
dx_ref, dw1_ref = O.conv3x3_bwd(x64, w1, d1_ref)
dx = ops.conv3x3_dgrad(d1, blocks.packed_weight(seq[0], 1, torch.float32, 128, 64), 64)
print('dx', rel_err(dx.cpu().numpy(), dx_ref))
dx2 = ops.conv3x3_dgrad(torch.from_numpy(d1_ref).float().cuda(), blocks.packed_weight(seq[0], 1, torch.float32, 128, 64), 64)
print('dx from ref d1', rel_err(dx2.cpu().numpy(), dx_ref))
dw1 = torch.empty(128, 64, 3, 3, device='cuda'); ops.conv3x3_wgrad(s1[0], d1, dw1)
print('dw1', rel_err(dw1.cpu().numpy(), dw1_ref))
err = np.abs(dy2.cpu().numpy() - d2)
print('n bad d2', (err > 1e-4 * np.abs(d2).max()).sum(), 'of', err.size)
maskg = (a2.cpu().numpy() > 0); maskr = (A2 > 0)
print('mask flips', (maskg != maskr).sum(), 'min |z2| at flips', np.abs(z2[maskg != maskr]) if (maskg != maskr).sum() < 20 else 'many')
# oracle with GPU mask
d2b, _, _ = O.bn_relu_train_bwd(y2, g2, m2, is2, a2.cpu().numpy().astype(np.float64), g64)
print('d2 vs oracle with gpu mask', rel_err(dy2.cpu().numpy(), d2b))
