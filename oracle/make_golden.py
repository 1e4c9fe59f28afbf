"""TEST INFRASTRUCTURE ONLY -- generate tests/golden/* by running the imported reference.

Run in the build container (where /root/reference exists):
    python oracle/make_golden.py
Imports `/root/reference/unet` and `/root/reference/utils/dice_score.py` on CPU (torch fp32,
plus one bf16-autocast pass for tolerance calibration), feeds them closed-form weights/inputs
from oracle/weights.py and writes small .npz/.json fixtures.  The fixtures are data only
(inputs are regenerated from the formula; expected outputs are stored).  The GPU box never
sees /root/reference; tests there read these files.
"""
import json
import os
import sys

import numpy as np
import torch
import torch.nn.functional as F

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path.insert(0, ROOT)
sys.path.insert(0, '/root/reference')

from oracle import weights as W  # noqa: E402
from unet import UNet  # noqa: E402  (the reference)
from unet.unet_parts import DoubleConv, Down, Up, OutConv  # noqa: E402
from utils.dice_score import dice_coeff, multiclass_dice_coeff, dice_loss  # noqa: E402

OUT = os.path.join(ROOT, 'tests', 'golden')
os.makedirs(OUT, exist_ok=True)
torch.set_num_threads(8)
torch.manual_seed(0)


def load_closed_form(module, prefix=''):
    sd = module.state_dict()
    new = {}
    for k, v in sd.items():
        arr = W.tensor_for_key(prefix + k, tuple(v.shape), str(v.dtype))
        new[k] = torch.from_numpy(arr).to(v.dtype)
    module.load_state_dict(new, strict=True)
    return module


BLOCK_DTYPE = torch.float64   # per-block goldens: the reference's modules run in fp64 so that
#                               cancellation-prone gradients (BN beta/gamma behind another BN) are exact;
#                               whole-net goldens below run in fp32 (the reference's own dtype) as well.


def t(name, shape, scale=1.0, offset=0.0, dtype=None):
    return torch.from_numpy(W.wave(name, shape, scale, offset)).to(dtype or BLOCK_DTYPE)


def npy(x):
    return x.detach().float().numpy()      # fp64 results are stored rounded to fp32 (file size)


def grads_of(module):
    return {('grad.' + k): npy(p.grad) for k, p in module.named_parameters()}


def buffers_of(module, tag):
    return {(tag + '.' + k): v.detach().numpy() for k, v in module.named_buffers()}


def save(name, **arrs):
    np.savez_compressed(os.path.join(OUT, name + '.npz'), **arrs)
    print(name, sum(a.nbytes for a in arrs.values()) // 1024, 'KiB raw')


# ------------------------------------------------------------------ 1. state_dict manifests
manifest = {}
for bil in (False, True):
    for ncls in (1, 2, 21):
        m = UNet(3, ncls, bil)
        manifest[f'bilinear={bil},n_classes={ncls}'] = {
            'entries': [[k, list(v.shape), str(v.dtype)] for k, v in m.state_dict().items()],
            'n_params': sum(p.numel() for p in m.parameters()),
        }
with open(os.path.join(OUT, 'state_dict_manifest.json'), 'w') as f:
    json.dump(manifest, f)

# ------------------------------------------------------------------ 2. per-block fwd+bwd, fp32
# DoubleConv(64 -> 128) train step on N2 x 64 x 16 x 23 (odd W), then eval forward.
dc = DoubleConv(64, 128).to(BLOCK_DTYPE)
load_closed_form(dc, 'dc.')
x = t('dc.x', (2, 64, 16, 23)).requires_grad_(True)
dc.train()
y = dc(x)
gy = t('dc.gy', tuple(y.shape), 0.01)
y.backward(gy)
arrs = {'y_train': npy(y), 'dx': npy(x.grad), **grads_of(dc), **buffers_of(dc, 'after')}
dc.eval()
arrs['y_eval'] = npy(dc(x.detach()))
save('double_conv', **arrs)

# DoubleConv(128 -> 64, mid 64) as used by the bilinear decoder
dcm = DoubleConv(128, 64, 64).to(BLOCK_DTYPE)
load_closed_form(dcm, 'dcm.')
x = t('dcm.x', (1, 128, 9, 33)).requires_grad_(True)
y = dcm(x)
gy = t('dcm.gy', tuple(y.shape), 0.01)
y.backward(gy)
save('double_conv_mid', y_train=npy(y), dx=npy(x.grad), **grads_of(dcm))

# Down(64 -> 128) on odd H and W (17 x 23 -> 8 x 11)
dn = Down(64, 128).to(BLOCK_DTYPE)
load_closed_form(dn, 'dn.')
x = t('dn.x', (2, 64, 17, 23)).requires_grad_(True)
y = dn(x)
gy = t('dn.gy', tuple(y.shape), 0.01)
y.backward(gy)
pooled = F.max_pool2d(x.detach(), 2)
save('down', y_train=npy(y), dx=npy(x.grad), pooled=npy(pooled), **grads_of(dn))

# Up, both flavours: x1 [2,128,8,11] -> 16x22, skip [2,64,17,23] => pad 1 bottom, 1 right
for bil in (False, True):
    up = Up(128, 64, bil).to(BLOCK_DTYPE)
    load_closed_form(up, f'up{int(bil)}.')
    # bilinear keeps channels, so the low-res input already has in/2 channels (unet_model.py:17-22)
    x1 = t('up.x1', (2, 64 if bil else 128, 8, 11)).requires_grad_(True)
    x2 = t('up.x2', (2, 64, 17, 23)).requires_grad_(True)
    y = up(x1, x2)
    gy = t('up.gy', tuple(y.shape), 0.01)
    y.backward(gy)
    with torch.no_grad():
        raw = up.up(x1)
    save('up_bilinear' if bil else 'up_convT', y_train=npy(y), dx1=npy(x1.grad), dx2=npy(x2.grad),
         upsampled=npy(raw), **grads_of(up))

# Up with larger size gap (diff 3 -> pad 1 left/top, 2 right/bottom)
up = Up(128, 64, False).to(BLOCK_DTYPE)
load_closed_form(up, 'up0.')
x1 = t('upg.x1', (1, 128, 4, 5)).requires_grad_(True)
x2 = t('upg.x2', (1, 64, 11, 13)).requires_grad_(True)
y = up(x1, x2)
gy = t('upg.gy', tuple(y.shape), 0.01)
y.backward(gy)
save('up_convT_gap3', y_train=npy(y), dx1=npy(x1.grad), dx2=npy(x2.grad), **grads_of(up))

# OutConv for 1 / 2 / 21 classes
for ncls in (1, 2, 21):
    oc = OutConv(64, ncls).to(BLOCK_DTYPE)
    load_closed_form(oc, f'oc{ncls}.')
    x = t('oc.x', (2, 64, 9, 13)).requires_grad_(True)
    y = oc(x)
    gy = t('oc.gy', tuple(y.shape), 0.1)
    y.backward(gy)
    save(f'outconv_{ncls}', y=npy(y), dx=npy(x.grad), **grads_of(oc))

# ------------------------------------------------------------------ 3. dice + loss
arrs = {}
a = t('dice.a', (3, 4, 9, 7), 0.5, 0.5)
b = (t('dice.b', (3, 4, 9, 7)) > 0).to(BLOCK_DTYPE)
arrs['a'], arrs['b'] = npy(a), npy(b)
arrs['dice_2d'] = npy(dice_coeff(a[0, 0], b[0, 0]))
arrs['dice_3d_rbf0'] = npy(dice_coeff(a[:, 0], b[:, 0], reduce_batch_first=False))
arrs['dice_3d_rbf1'] = npy(dice_coeff(a[:, 0], b[:, 0], reduce_batch_first=True))
arrs['mc_rbf0'] = npy(multiclass_dice_coeff(a, b, reduce_batch_first=False))
arrs['mc_rbf1'] = npy(multiclass_dice_coeff(a, b, reduce_batch_first=True))
arrs['loss_mc'] = npy(dice_loss(a, b, multiclass=True))
arrs['loss_bin'] = npy(dice_loss(a[:, 0], b[:, 0], multiclass=False))
z = torch.zeros(2, 5, 6, dtype=BLOCK_DTYPE)
arrs['dice_empty'] = npy(dice_coeff(z, z, reduce_batch_first=True))
arrs['dice_empty_rbf0'] = npy(dice_coeff(z, z, reduce_batch_first=False))
ag = a.clone().requires_grad_(True)
dice_loss(ag, b, multiclass=True).backward()
arrs['dloss_mc_da'] = npy(ag.grad)
ag = a[:, 0].clone().requires_grad_(True)
dice_coeff(ag, b[:, 0], reduce_batch_first=False).backward()
arrs['ddice_rbf0_da'] = npy(ag.grad)
save('dice', **arrs)

for ncls in (1, 2, 21):
    logits = t(f'loss{ncls}.z', (2, ncls, 12, 19), 2.0, dtype=torch.float32).requires_grad_(True)
    if ncls == 1:
        mask = torch.from_numpy(W.int_wave('loss.mask', (2, 12, 19), 2))
        loss = F.binary_cross_entropy_with_logits(logits.squeeze(1), mask.float())
        loss = loss + dice_loss(torch.sigmoid(logits.squeeze(1)), mask.float(), multiclass=False)
    else:
        mask = torch.from_numpy(W.int_wave('loss.mask', (2, 12, 19), ncls))
        loss = F.cross_entropy(logits, mask)
        loss = loss + dice_loss(F.softmax(logits, dim=1).float(),
                                F.one_hot(mask, ncls).permute(0, 3, 1, 2).float(), multiclass=True)
    loss.backward()
    arrs = {'loss': npy(loss), 'dlogits': npy(logits.grad), 'mask': mask.numpy()}
    # all-background mask (empty foreground)
    lz = logits.detach().clone().requires_grad_(True)
    m0 = torch.zeros_like(mask)
    if ncls == 1:
        l0 = F.binary_cross_entropy_with_logits(lz.squeeze(1), m0.float()) + \
            dice_loss(torch.sigmoid(lz.squeeze(1)), m0.float(), multiclass=False)
    else:
        l0 = F.cross_entropy(lz, m0) + dice_loss(F.softmax(lz, dim=1).float(),
                                                 F.one_hot(m0, ncls).permute(0, 3, 1, 2).float(), multiclass=True)
    l0.backward()
    arrs['loss_empty'] = npy(l0)
    arrs['dlogits_empty'] = npy(lz.grad)
    save(f'loss_{ncls}', **arrs)


# ------------------------------------------------------------------ 4. whole network
def seg_loss(model, logits, mask):
    if model.n_classes == 1:
        return F.binary_cross_entropy_with_logits(logits.squeeze(1), mask.float()) + \
            dice_loss(torch.sigmoid(logits.squeeze(1)), mask.float(), multiclass=False)
    return F.cross_entropy(logits, mask) + dice_loss(
        F.softmax(logits, dim=1).float(),
        F.one_hot(mask, model.n_classes).permute(0, 3, 1, 2).float(), multiclass=True)


KEEP_FULL = ('outc.conv.weight', 'outc.conv.bias', 'inc.double_conv.0.weight', 'inc.double_conv.1.weight',
             'up4.up.weight', 'up4.up.bias', 'down4.maxpool_conv.1.double_conv.4.bias',
             'up4.conv.double_conv.0.weight')

# Image sizes are chosen so that the deepest BatchNorm (H/16 x W/16 x N samples per channel) still sees ~100
# samples: with the 4 samples of a 32x47 image the reference's OWN fp32 and fp64 runs disagree by 10% in
# gradient norms (ReLU masks flip on values within rounding of zero), which would pin nothing.
for tag, (ncls, bil, shape) in {
    'unet_c2_convT': (2, False, (2, 3, 96, 142)),
    'unet_c2_bilinear': (2, True, (2, 3, 96, 142)),
    'unet_c21_convT': (21, False, (2, 3, 64, 80)),
    'unet_c1_convT': (1, False, (2, 3, 112, 99)),
}.items():
    net = UNet(3, ncls, bil)
    load_closed_form(net)
    net.train()
    img = t(tag + '.img', shape, 0.5, 0.5, dtype=torch.float32)
    mask = torch.from_numpy(W.int_wave(tag + '.mask', (shape[0],) + shape[2:], max(ncls, 2)))
    logits = net(img)
    loss = seg_loss(net, logits, mask)
    loss.backward()
    arrs = {'logits_train': npy(logits), 'loss': npy(loss)}
    names, norms = [], []
    for k, p in net.named_parameters():
        names.append(k)
        norms.append(float(p.grad.double().norm()))
    arrs['grad_norms'] = np.array(norms)
    arrs['grad_names'] = np.array(names)
    for k, v in net.named_buffers():
        if k.startswith('inc.') or k.startswith('up4.conv') or k.startswith('down4.'):
            arrs['after.' + k] = v.detach().numpy()
    # the same step in fp64 (reference modules cast with .double()): exact gradients
    net64 = load_closed_form(UNet(3, ncls, bil)).double().train()
    logits64 = net64(img.double())
    seg_loss(net64, logits64, mask).backward()
    norms64 = []
    grads32 = {k: p.grad.detach().double() for k, p in net.named_parameters()}
    for k, p in net64.named_parameters():
        norms64.append(float(p.grad.norm()))
        if k in KEEP_FULL:
            arrs['grad64.' + k] = p.grad.detach().numpy().astype(np.float32)
            # how far the reference's OWN fp32 gradient is from the exact one (L2-relative): the yardstick for the
            # gradients that sit behind ReLU masks of near-zero pre-activations
            arrs['grad_gap32.' + k] = np.array(float((grads32[k] - p.grad).norm() / p.grad.norm()))
    arrs['grad_norms64'] = np.array(norms64)
    arrs['logits_train64'] = npy(logits64)
    # the reference's own bf16 training-mode forward (fresh net, batch statistics): the yardstick for what
    # bf16 can agree on, since the reference publishes no bf16 numbers
    netb = load_closed_form(UNet(3, ncls, bil)).train()
    with torch.no_grad(), torch.autocast('cpu', dtype=torch.bfloat16):
        arrs['logits_train_bf16'] = npy(netb(img))
    net.eval()      # NB: eval AFTER the training step above, i.e. with the updated running statistics
    with torch.no_grad():
        arrs['logits_eval'] = npy(net(img))
        # bf16 autocast (SURVEY section 5 "AMP detail"): tolerance calibration only
        with torch.autocast('cpu', dtype=torch.bfloat16):
            arrs['logits_eval_bf16'] = npy(net(img))
    save(tag, **arrs)

# ------------------------------------------------------------------ 5. validation round (evaluate.py:8-40)
from evaluate import evaluate as ref_evaluate   # noqa: E402  (reference module; imports tqdm only)

for tag, (ncls, shape) in {'eval_c2': (2, (2, 3, 96, 142)), 'eval_c21': (21, (2, 3, 64, 80))}.items():
    net = UNet(3, ncls, False)
    load_closed_form(net)
    batches = []
    for b in range(2):
        img = t(f'{tag}.img{b}', shape, 0.5, 0.5, dtype=torch.float32)
        mask = torch.from_numpy(W.int_wave(f'{tag}.mask{b}', (shape[0],) + shape[2:], ncls))
        batches.append({'image': img, 'mask': mask})
    score = ref_evaluate(net, batches, torch.device('cpu'), False)
    assert net.training                                   # evaluate() restores train mode (evaluate.py:39)
    # how close to a tie the argmax decisions are: second-best margin quantiles, for the tolerance of the GPU test
    net.eval()
    with torch.no_grad():
        top2 = net(batches[0]['image']).topk(2, dim=1).values
    margin = (top2[:, 0] - top2[:, 1]).flatten()
    save(tag, dice=npy(score), margin_q01=np.array(float(margin.quantile(0.01))),
         n_batches=np.array(len(batches)))
try:                                                      # n_classes == 1: the reference's own evaluate() asserts
    net1 = load_closed_form(UNet(3, 1, False))
    ref_evaluate(net1, [{'image': t('eval_c1.img', (1, 3, 32, 48), 0.5, 0.5, dtype=torch.float32),
                         'mask': torch.zeros((1, 32, 48), dtype=torch.long)}], torch.device('cpu'), False)
    c1 = 'returned'
except AssertionError:
    c1 = 'AssertionError'
save('eval_c1_behaviour', raised=np.array(c1))

# ------------------------------------------------------------------ 6. predict_img (predict.py:15-33)
# predict.py itself cannot be imported here (it imports torchvision, absent offline): its body is run line by line
# on the reference's own net and BasicDataset.preprocess (importable).
from PIL import Image   # noqa: E402
from utils.data_loading import BasicDataset as RefDataset   # noqa: E402

for tag, ncls in {'predict_c2': 2, 'predict_c1': 1}.items():
    net = load_closed_form(UNet(3, ncls, False))
    rgb = (W.wave(tag + '.rgb', (70, 101, 3), 0.5, 0.5) * 255).astype(np.uint8)           # H=70, W=101 "photo"
    full_img = Image.fromarray(rgb)
    net.eval()
    img = torch.from_numpy(RefDataset.preprocess(None, full_img, 0.5, is_mask=False))    # predict.py:21
    img = img.unsqueeze(0).to(dtype=torch.float32)
    with torch.no_grad():
        output = net(img)
        up = F.interpolate(output, (full_img.size[1], full_img.size[0]), mode='bilinear')  # predict.py:27
        mask = up.argmax(dim=1) if ncls > 1 else torch.sigmoid(up) > 0.5                   # predict.py:28-31
    out = mask[0].long().squeeze().numpy()
    if ncls > 1:
        top2 = up.topk(2, dim=1).values
        margin = (top2[:, 0] - top2[:, 1])[0].numpy()
    else:
        margin = (torch.sigmoid(up) - 0.5).abs()[0, 0].numpy()
    save(tag, rgb=rgb, preprocessed=img[0].numpy(), mask=out.astype(np.int64), margin=margin.astype(np.float32))

# (section 7, BasicDataset.preprocess, lives in oracle/make_golden_preprocess.py)

# checkpoint file shaped like train.py:163-165 writes it (state_dict + 'mask_values'), tiny variant
# is impossible (widths are hard-coded), so only the key manifest above pins the layout.
print('done')
