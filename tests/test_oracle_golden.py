"""Pin the oracle (oracle/ops_numpy.py, oracle/unet_torch.py) to outputs of the imported
reference (tests/golden/*.npz, made by oracle/make_golden.py).  CPU only."""
import json
import os

import numpy as np
import pytest
import torch

from conftest import GOLDEN, UNET_CASES, golden, rel_err
from oracle import ops_numpy as O
from oracle import weights as W
from oracle.unet_torch import TorchUNet, segmentation_loss

TOL = 2e-6     # fp64 reference run (stored as fp32) vs float64 restatement


def nhwc(a):
    return np.ascontiguousarray(np.transpose(a, (0, 2, 3, 1))).astype(np.float64)


def nchw(a):
    return np.transpose(a, (0, 3, 1, 2))


def w64(key, shape):
    return W.tensor_for_key(key, shape, 'float32').astype(np.float64)


def wave64(name, shape, scale=1.0, offset=0.0):
    return W.wave(name, shape, scale, offset).astype(np.float64)


def double_conv_np(prefix, x, cin, cmid, cout, dy=None):
    w1 = w64(prefix + 'double_conv.0.weight', (cmid, cin, 3, 3))
    g1, b1 = w64(prefix + 'double_conv.1.weight', (cmid,)), w64(prefix + 'double_conv.1.bias', (cmid,))
    w2 = w64(prefix + 'double_conv.3.weight', (cout, cmid, 3, 3))
    g2, b2 = w64(prefix + 'double_conv.4.weight', (cout,)), w64(prefix + 'double_conv.4.bias', (cout,))
    y1 = O.conv3x3_fwd(x, w1)
    z1, m1, v1, is1 = O.bn_train_fwd(y1, g1, b1)
    a1 = O.relu(z1)
    y2 = O.conv3x3_fwd(a1, w2)
    z2, m2, v2, is2 = O.bn_train_fwd(y2, g2, b2)
    a2 = O.relu(z2)
    out = {'y': a2, 'stats': (m1, v1, m2, v2)}
    if dy is not None:
        d2, dg2, db2 = O.bn_relu_train_bwd(y2, g2, m2, is2, a2, dy)
        da1, dw2 = O.conv3x3_bwd(a1, w2, d2)
        d1, dg1, db1 = O.bn_relu_train_bwd(y1, g1, m1, is1, a1, da1)
        dx, dw1 = O.conv3x3_bwd(x, w1, d1)
        out.update(dx=dx, grads={'double_conv.0.weight': dw1, 'double_conv.1.weight': dg1,
                                 'double_conv.1.bias': db1, 'double_conv.3.weight': dw2,
                                 'double_conv.4.weight': dg2, 'double_conv.4.bias': db2})
    return out


def test_double_conv_matches_reference():
    g = golden('double_conv')
    x = nhwc(W.wave('dc.x', (2, 64, 16, 23)))
    dy = nhwc(W.wave('dc.gy', (2, 128, 16, 23), 0.01))
    r = double_conv_np('dc.', x, 64, 128, 128, dy)
    assert rel_err(nchw(r['y']), g['y_train']) < TOL
    assert rel_err(nchw(r['dx']), g['dx']) < TOL
    for k, v in r['grads'].items():
        assert rel_err(v, g['grad.' + k]) < TOL, k
    # running stats after one step (momentum 0.1, unbiased variance)
    m1, v1, m2, v2 = r['stats']
    cnt = 2 * 16 * 23
    rm, rv = O.bn_running_update(w64('dc.double_conv.1.running_mean', (128,)),
                                 w64('dc.double_conv.1.running_var', (128,)), m1, v1, cnt)
    assert rel_err(rm, g['after.double_conv.1.running_mean']) < TOL
    assert rel_err(rv, g['after.double_conv.1.running_var']) < TOL
    assert int(g['after.double_conv.1.num_batches_tracked']) == 1
    # eval mode uses the updated running stats
    w1 = w64('dc.double_conv.0.weight', (128, 64, 3, 3))
    z = O.bn_eval_fwd(O.conv3x3_fwd(x, w1), w64('dc.double_conv.1.weight', (128,)),
                      w64('dc.double_conv.1.bias', (128,)), rm, rv)
    rm2, rv2 = O.bn_running_update(w64('dc.double_conv.4.running_mean', (128,)),
                                   w64('dc.double_conv.4.running_var', (128,)), m2, v2, cnt)
    z2 = O.bn_eval_fwd(O.conv3x3_fwd(O.relu(z), w64('dc.double_conv.3.weight', (128, 128, 3, 3))),
                       w64('dc.double_conv.4.weight', (128,)), w64('dc.double_conv.4.bias', (128,)), rm2, rv2)
    assert rel_err(nchw(O.relu(z2)), g['y_eval']) < TOL


def test_double_conv_mid_channels():
    g = golden('double_conv_mid')
    x = nhwc(W.wave('dcm.x', (1, 128, 9, 33)))
    dy = nhwc(W.wave('dcm.gy', (1, 64, 9, 33), 0.01))
    r = double_conv_np('dcm.', x, 128, 64, 64, dy)
    assert rel_err(nchw(r['y']), g['y_train']) < TOL
    assert rel_err(nchw(r['dx']), g['dx']) < TOL


def test_down_matches_reference():
    g = golden('down')
    x = nhwc(W.wave('dn.x', (2, 64, 17, 23)))
    p = O.maxpool2_fwd(x)
    assert np.array_equal(nchw(p).astype(np.float32), g['pooled'])
    dy = nhwc(W.wave('dn.gy', (2, 128, 8, 11), 0.01))
    r = double_conv_np('dn.maxpool_conv.1.', p, 64, 128, 128, dy)
    assert rel_err(nchw(r['y']), g['y_train']) < TOL
    dx = O.maxpool2_bwd(x, r['dx'])
    assert rel_err(nchw(dx), g['dx']) < TOL
    assert np.all(dx[:, 16, :, :] == 0) and np.all(dx[:, :, 22, :] == 0)   # dropped odd row/col


@pytest.mark.parametrize('bilinear', [False, True])
def test_up_matches_reference(bilinear):
    g = golden('up_bilinear' if bilinear else 'up_convT')
    c1 = 64 if bilinear else 128
    x1 = nhwc(W.wave('up.x1', (2, c1, 8, 11)))
    x2 = nhwc(W.wave('up.x2', (2, 64, 17, 23)))
    pre = f'up{int(bilinear)}.'
    if bilinear:
        u = O.bilinear2x_ac_fwd(x1)
    else:
        wt, bt = w64(pre + 'up.weight', (128, 64, 2, 2)), w64(pre + 'up.bias', (64,))
        u = O.convT2x2_fwd(x1, wt, bt)
    assert rel_err(nchw(u), g['upsampled']) < TOL
    cat = O.pad_cat(x2, u)
    assert cat.shape == (2, 17, 23, 128)
    assert np.all(cat[:, 16, :, 64:] == 0) and np.all(cat[:, :, 22, 64:] == 0)
    dy = nhwc(W.wave('up.gy', (2, 64, 17, 23), 0.01))
    r = double_conv_np(pre + 'conv.', cat, 128, 64, 64, dy)
    assert rel_err(nchw(r['y']), g['y_train']) < TOL
    dskip, dup = O.pad_cat_bwd(x2.shape, u.shape, r['dx'])
    assert rel_err(nchw(dskip), g['dx2']) < TOL
    if bilinear:
        dx1 = O.bilinear2x_ac_bwd(x1.shape, dup)
    else:
        dx1, dwt, dbt = O.convT2x2_bwd(x1, wt, dup)
        assert rel_err(dwt, g['grad.up.weight']) < TOL
        assert rel_err(dbt, g['grad.up.bias']) < TOL
    assert rel_err(nchw(dx1), g['dx1']) < TOL


def test_up_gap3_padding_split():
    g = golden('up_convT_gap3')
    x1 = nhwc(W.wave('upg.x1', (1, 128, 4, 5)))
    x2 = nhwc(W.wave('upg.x2', (1, 64, 11, 13)))
    wt, bt = w64('up0.up.weight', (128, 64, 2, 2)), w64('up0.up.bias', (64,))
    cat = O.pad_cat(x2, O.convT2x2_fwd(x1, wt, bt))
    r = double_conv_np('up0.conv.', cat, 128, 64, 64)
    assert rel_err(nchw(r['y']), g['y_train']) < TOL


@pytest.mark.parametrize('ncls', [1, 2, 21])
def test_outconv_matches_reference(ncls):
    g = golden(f'outconv_{ncls}')
    x = nhwc(W.wave('oc.x', (2, 64, 9, 13)))
    w, b = w64(f'oc{ncls}.conv.weight', (ncls, 64, 1, 1)), w64(f'oc{ncls}.conv.bias', (ncls,))
    y = O.conv1x1_fwd(x, w, b)
    assert rel_err(nchw(y), g['y']) < TOL
    dx, dw, db = O.conv1x1_bwd(x, w, nhwc(W.wave('oc.gy', (2, ncls, 9, 13), 0.1)))
    assert rel_err(nchw(dx), g['dx']) < TOL
    assert rel_err(dw, g['grad.conv.weight']) < TOL
    assert rel_err(db, g['grad.conv.bias']) < TOL


def test_dice_matches_reference():
    g = golden('dice')
    a, b = g['a'].astype(np.float64), g['b'].astype(np.float64)
    assert abs(O.dice_coeff(a[0, 0], b[0, 0]) - g['dice_2d']) < 1e-6
    assert abs(O.dice_coeff(a[:, 0], b[:, 0], False) - g['dice_3d_rbf0']) < 1e-6
    assert abs(O.dice_coeff(a[:, 0], b[:, 0], True) - g['dice_3d_rbf1']) < 1e-6
    assert abs(O.multiclass_dice_coeff(a, b, False) - g['mc_rbf0']) < 1e-6
    assert abs(O.multiclass_dice_coeff(a, b, True) - g['mc_rbf1']) < 1e-6
    z = np.zeros((2, 5, 6))
    assert O.dice_coeff(z, z, True) == 1.0 == float(g['dice_empty'])
    assert float(g['dice_empty_rbf0']) == 1.0
    with pytest.raises(AssertionError):
        O.dice_coeff(a, b, True)          # 4-D with reduce_batch_first trips dice_score.py:8


@pytest.mark.parametrize('ncls', [1, 2, 21])
def test_loss_matches_reference(ncls):
    g = golden(f'loss_{ncls}')
    z = nhwc(W.wave(f'loss{ncls}.z', (2, ncls, 12, 19), 2.0))
    loss, dz = O.seg_loss_fwd_bwd(z, g['mask'])
    assert abs(loss - g['loss']) < 2e-6 * max(1, abs(float(g['loss'])))
    assert rel_err(nchw(dz), g['dlogits']) < 1e-4
    loss0, dz0 = O.seg_loss_fwd_bwd(z, np.zeros_like(g['mask']))
    assert abs(loss0 - g['loss_empty']) < 2e-6 * max(1, abs(float(g['loss_empty'])))
    assert rel_err(nchw(dz0), g['dlogits_empty']) < 1e-4


def test_state_dict_manifest():
    with open(os.path.join(GOLDEN, 'state_dict_manifest.json')) as f:
        man = json.load(f)
    for bil in (False, True):
        for ncls in (1, 2, 21):
            ent = man[f'bilinear={bil},n_classes={ncls}']
            sd = TorchUNet(3, ncls, bil).state_dict()
            assert [[k, list(v.shape), str(v.dtype)] for k, v in sd.items()] == ent['entries']
    assert len(man['bilinear=False,n_classes=2']['entries']) == 118
    assert len(man['bilinear=True,n_classes=2']['entries']) == 110
    assert man['bilinear=False,n_classes=2']['n_params'] == 31037698
    assert man['bilinear=True,n_classes=2']['n_params'] == 17263042


def load_closed_form(net):
    sd = net.state_dict()
    net.load_state_dict({k: torch.from_numpy(W.tensor_for_key(k, tuple(v.shape), str(v.dtype))).to(v.dtype)
                         for k, v in sd.items()}, strict=True)


@pytest.mark.parametrize('tag,ncls,bil,shape', UNET_CASES)
def test_torch_restatement_matches_reference(tag, ncls, bil, shape):
    g = golden(tag)
    net = TorchUNet(3, ncls, bil)
    load_closed_form(net)
    net.train()
    img = torch.from_numpy(W.wave(tag + '.img', shape, 0.5, 0.5))
    mask = torch.from_numpy(W.int_wave(tag + '.mask', (shape[0],) + shape[2:], max(ncls, 2)))
    logits = net(img)
    loss = segmentation_loss(logits, mask, ncls)
    loss.backward()
    assert rel_err(logits.detach().numpy(), g['logits_train']) < 1e-5
    assert abs(float(loss) - float(g['loss'])) < 1e-5
    # fp32 gradients: bit-for-bit the same ATen ops as the reference => identical norms
    norms = dict(zip(g['grad_names'].tolist(), g['grad_norms'].tolist()))
    for k, p in net.named_parameters():
        assert abs(float(p.grad.double().norm()) - norms[k]) <= 1e-5 * max(norms[k], 1e-6), k
    net.eval()
    with torch.no_grad():
        assert rel_err(net(img).numpy(), g['logits_eval']) < 1e-5
