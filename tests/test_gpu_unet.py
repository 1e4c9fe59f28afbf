"""Whole-network and per-module parity of the MI355X U-Net against fixtures produced by the imported reference
(tests/golden, oracle/make_golden.py) and against the stock-PyTorch restatement (oracle/unet_torch.py) on CPU."""
import numpy as np
import pytest
import torch

from conftest import UNET_CASES, golden, l2_err, rel_err
from oracle import weights as W

pytestmark = pytest.mark.gpu


def load_closed_form(mod, prefix=''):
    sd = mod.state_dict()
    mod.load_state_dict({k: torch.from_numpy(W.tensor_for_key(prefix + k, tuple(v.shape), str(v.dtype))).to(v.dtype)
                         for k, v in sd.items()}, strict=True)
    return mod


def wave(name, shape, scale=1.0, offset=0.0):
    return torch.from_numpy(W.wave(name, shape, scale, offset))


def cl(t):
    return t.cuda().contiguous(memory_format=torch.channels_last)


def npy(t):
    return t.detach().float().cpu().numpy()


# ------------------------------------------------------------------------------------- modules, fp32
def test_double_conv_module_fp32():
    from unet.unet_parts import DoubleConv
    g = golden('double_conv')
    dc = load_closed_form(DoubleConv(64, 128), 'dc.').cuda().train()
    x = cl(wave('dc.x', (2, 64, 16, 23))).requires_grad_(True)
    y = dc(x)
    assert y.shape == (2, 128, 16, 23)
    y.backward(cl(wave('dc.gy', (2, 128, 16, 23), 0.01)))
    assert rel_err(npy(y), g['y_train']) < 1e-4
    assert rel_err(npy(x.grad), g['dx']) < 1e-3
    for k, p in dc.named_parameters():
        assert rel_err(npy(p.grad), g['grad.' + k]) < 1e-3, k
    for k, b in dc.named_buffers():
        ref = g['after.' + k]
        assert rel_err(npy(b), ref) < 1e-5 if ref.ndim else int(b) == int(ref), k
    dc.eval()
    with torch.no_grad():
        assert rel_err(npy(dc(x.detach())), g['y_eval']) < 1e-4


def test_down_module_fp32():
    from unet.unet_parts import Down
    g = golden('down')
    dn = load_closed_form(Down(64, 128), 'dn.').cuda().train()
    x = cl(wave('dn.x', (2, 64, 17, 23))).requires_grad_(True)
    y = dn(x)
    y.backward(cl(wave('dn.gy', (2, 128, 8, 11), 0.01)))
    assert rel_err(npy(y), g['y_train']) < 1e-4
    assert rel_err(npy(x.grad), g['dx']) < 1e-3
    for k, p in dn.named_parameters():
        assert rel_err(npy(p.grad), g['grad.' + k]) < 2e-2, k     # BN1 grads behind BN2 are cancellation sums


@pytest.mark.parametrize('bilinear', [False, True])
def test_up_module_fp32(bilinear):
    from unet.unet_parts import Up
    g = golden('up_bilinear' if bilinear else 'up_convT')
    up = load_closed_form(Up(128, 64, bilinear), f'up{int(bilinear)}.').cuda().train()
    x1 = cl(wave('up.x1', (2, 64 if bilinear else 128, 8, 11))).requires_grad_(True)
    x2 = cl(wave('up.x2', (2, 64, 17, 23))).requires_grad_(True)
    y = up(x1, x2)
    y.backward(cl(wave('up.gy', (2, 64, 17, 23), 0.01)))
    assert rel_err(npy(y), g['y_train']) < 1e-4
    assert rel_err(npy(x1.grad), g['dx1']) < 1e-3
    assert rel_err(npy(x2.grad), g['dx2']) < 1e-3
    for k in ('up.weight', 'up.bias', 'conv.double_conv.3.weight'):
        if 'grad.' + k in g:
            assert rel_err(npy(dict(up.named_parameters())[k].grad), g['grad.' + k]) < 1e-3, k


def test_up_module_uneven_padding():
    from unet.unet_parts import Up
    g = golden('up_convT_gap3')
    up = load_closed_form(Up(128, 64, False), 'up0.').cuda().train()
    y = up(cl(wave('upg.x1', (1, 128, 4, 5))), cl(wave('upg.x2', (1, 64, 11, 13))))
    assert rel_err(npy(y), g['y_train']) < 1e-4


@pytest.mark.parametrize('ncls', [1, 2, 21])
def test_outconv_module_fp32(ncls):
    from unet.unet_parts import OutConv
    g = golden(f'outconv_{ncls}')
    oc = load_closed_form(OutConv(64, ncls), f'oc{ncls}.').cuda()
    x = cl(wave('oc.x', (2, 64, 9, 13))).requires_grad_(True)
    y = oc(x)
    y.backward(cl(wave('oc.gy', (2, ncls, 9, 13), 0.1)))
    assert rel_err(npy(y), g['y']) < 1e-4
    assert rel_err(npy(x.grad), g['dx']) < 1e-4
    assert rel_err(npy(oc.conv.weight.grad), g['grad.conv.weight']) < 1e-4
    assert rel_err(npy(oc.conv.bias.grad), g['grad.conv.bias']) < 1e-4


def test_dice_api():
    from utils.dice_score import dice_coeff, multiclass_dice_coeff, dice_loss
    g = golden('dice')
    a = torch.from_numpy(g['a']).cuda()
    b = torch.from_numpy(g['b']).cuda()
    assert abs(float(dice_coeff(a[0, 0], b[0, 0])) - float(g['dice_2d'])) < 1e-6
    assert abs(float(dice_coeff(a[:, 0], b[:, 0], reduce_batch_first=False)) - float(g['dice_3d_rbf0'])) < 1e-6
    assert abs(float(dice_coeff(a[:, 0], b[:, 0], reduce_batch_first=True)) - float(g['dice_3d_rbf1'])) < 1e-6
    assert abs(float(multiclass_dice_coeff(a, b, reduce_batch_first=False)) - float(g['mc_rbf0'])) < 1e-6
    assert abs(float(dice_loss(a, b, multiclass=True)) - float(g['loss_mc'])) < 1e-6
    assert abs(float(dice_loss(a[:, 0], b[:, 0], multiclass=False)) - float(g['loss_bin'])) < 1e-6
    ag = a.clone().requires_grad_(True)
    dice_loss(ag, b, multiclass=True).backward()
    assert rel_err(npy(ag.grad), g['dloss_mc_da']) < 1e-5
    with pytest.raises(AssertionError):
        dice_coeff(a, b, reduce_batch_first=True)      # 4-D + reduce_batch_first: dice_score.py:8
    with pytest.raises(AssertionError):
        dice_coeff(a, b[:, :2])                        # size mismatch: dice_score.py:7
    z = torch.zeros((2, 5, 6), device='cuda')
    assert float(dice_coeff(z, z, reduce_batch_first=True)) == 1.0


# ------------------------------------------------------------------------------------- whole network
CASES = UNET_CASES


@pytest.mark.parametrize('tag,ncls,bil,shape', CASES)
def test_unet_fp32_train_step_matches_reference(tag, ncls, bil, shape):
    from unet import UNet
    from unetk.loss import segmentation_loss
    g = golden(tag)
    net = load_closed_form(UNet(3, ncls, bil)).cuda().train()
    img = cl(wave(tag + '.img', shape, 0.5, 0.5))
    mask = torch.from_numpy(W.int_wave(tag + '.mask', (shape[0],) + shape[2:], max(ncls, 2))).cuda()
    logits = net(img)
    assert logits.shape == (shape[0], ncls, shape[2], shape[3]) and logits.dtype == torch.float32
    loss = segmentation_loss(logits, mask)
    loss.backward()
    # BASELINE.json north_star: fp32 logits within 1e-4 relative of the reference
    assert rel_err(npy(logits), g['logits_train']) < 1e-4
    assert abs(float(loss) - float(g['loss'])) < 1e-4
    names = g['grad_names'].tolist()
    norms64 = dict(zip(names, g['grad_norms64'].tolist()))
    params = dict(net.named_parameters())
    worst = 0.0
    for k in names:
        got = float(params[k].grad.double().norm())
        worst = max(worst, abs(got - norms64[k]) / max(norms64[k], 1e-7))
        assert abs(got - norms64[k]) <= 2e-2 * max(norms64[k], 1e-7), (k, got, norms64[k])
    # full gradients against the reference's fp64 run.  Gradients behind ReLU masks of near-zero pre-activations
    # (the 40-sample BatchNorm at the bottleneck above all) move by ~1 % when one mask flips: the reference's OWN
    # fp32 gradient is grad_gap32 away from the exact one (1.2 % for down4's BN bias in the 21-class case), so the
    # bar is 1e-2 or twice that gap, whichever is larger.
    for k in [k[7:] for k in g.files if k.startswith('grad64.')]:
        tol = max(1e-2, 2.0 * float(g['grad_gap32.' + k]))
        assert l2_err(npy(params[k].grad), g['grad64.' + k]) < tol, (k, tol)
    for k in [k[6:] for k in g.files if k.startswith('after.')]:
        b = dict(net.named_buffers())[k]
        ref = g['after.' + k]
        if ref.ndim == 0:
            assert int(b) == int(ref), k
        else:
            assert rel_err(npy(b), ref) < 1e-4, k
    net.eval()
    with torch.no_grad():
        ev = net(img)
    assert rel_err(npy(ev), g['logits_eval']) < 1e-4


@pytest.mark.parametrize('tag,ncls,bil,shape', CASES)
def test_unet_bf16_autocast_matches_reference(tag, ncls, bil, shape):
    from unet import UNet
    g = golden(tag)
    net = load_closed_form(UNet(3, ncls, bil)).cuda().train()
    img = cl(wave(tag + '.img', shape, 0.5, 0.5))
    with torch.no_grad():
        net(img)                    # the golden's eval pass ran after one training forward: same running stats
        net.eval()
        with torch.autocast('cuda', dtype=torch.bfloat16):
            logits = net(img)
    assert logits.dtype == torch.bfloat16
    # north_star: bf16 within 2e-2 relative of the reference fp32 logits
    assert rel_err(npy(logits), g['logits_eval']) < 2e-2
    if ncls > 1:
        agree = (npy(logits).argmax(1) == g['logits_eval'].argmax(1)).mean()
        ref_bf16 = (g['logits_eval_bf16'].argmax(1) == g['logits_eval'].argmax(1)).mean()
        # north_star asks for >= 99.9 % argmax agreement; on these random-weight fixtures (logits within 0.3 of
        # each other) the reference's OWN bf16 run agrees with its fp32 run less often than that, so the
        # reference's bf16 agreement is the yardstick
        assert agree >= min(0.999, ref_bf16 - 0.005), (agree, ref_bf16)


@pytest.mark.parametrize('tag,ncls,bil,shape', CASES)
def test_unet_bf16_train_forward_vs_reference_bf16_gap(tag, ncls, bil, shape):
    """Training-mode (batch statistics) bf16 forward.  With random weights the reference's own CPU bf16 autocast
    run is 4-7 % away from its fp32 run (fixture logits_train_bf16), beyond the 2e-2 of north_star; the bar here is
    2e-2 or 1.25x the reference's own bf16 deviation, whichever is larger."""
    from unet import UNet
    g = golden(tag)
    net = load_closed_form(UNet(3, ncls, bil)).cuda().train()
    img = cl(wave(tag + '.img', shape, 0.5, 0.5))
    with torch.no_grad(), torch.autocast('cuda', dtype=torch.bfloat16):
        logits = net(img)
    ref_gap = rel_err(g['logits_train_bf16'], g['logits_train'])
    assert rel_err(npy(logits), g['logits_train']) < max(2e-2, 1.25 * ref_gap), ref_gap


def test_unet_bf16_train_step_close_to_fp32():
    from unet import UNet
    from unetk.loss import segmentation_loss
    tag, ncls, bil, shape = CASES[0]
    g = golden(tag)
    net = load_closed_form(UNet(3, ncls, bil)).cuda().train()
    img = cl(wave(tag + '.img', shape, 0.5, 0.5))
    mask = torch.from_numpy(W.int_wave(tag + '.mask', (shape[0],) + shape[2:], 2)).cuda()
    with torch.autocast('cuda', dtype=torch.bfloat16):
        logits = net(img)
        loss = segmentation_loss(logits, mask)
    loss.backward()
    ref_gap = rel_err(g['logits_train_bf16'], g['logits_train'])
    assert rel_err(npy(logits), g['logits_train']) < max(2e-2, 1.25 * ref_gap)
    assert abs(float(loss) - float(g['loss'])) < 2e-2
    norms64 = dict(zip(g['grad_names'].tolist(), g['grad_norms64'].tolist()))
    for k, p in net.named_parameters():
        assert p.grad is not None and p.grad.dtype == torch.float32 and torch.isfinite(p.grad).all(), k
    # gradients of the last layers (short bf16 paths) agree closely; deep ones accumulate bf16 rounding
    for k in ('outc.conv.weight', 'outc.conv.bias', 'up4.conv.double_conv.4.bias'):
        got = float(dict(net.named_parameters())[k].grad.double().norm())
        assert abs(got - norms64[k]) <= 5e-2 * norms64[k], (k, got, norms64[k])


def test_checkpoint_roundtrip_strict():
    """state_dict written the way train.py:163-165 does (plus 'mask_values') loads both ways with strict=True."""
    from oracle.unet_torch import TorchUNet
    from unet import UNet
    ref = load_closed_form(TorchUNet(3, 2, False))
    ckpt = ref.state_dict()
    ckpt['mask_values'] = [0, 1]
    sd = dict(ckpt)
    del sd['mask_values']                      # train.py:205-207 / predict.py:94
    net = UNet(3, 2, False)
    net.load_state_dict(sd, strict=True)
    back = TorchUNet(3, 2, False)
    back.load_state_dict(net.state_dict(), strict=True)
    for (k1, v1), (k2, v2) in zip(ref.state_dict().items(), back.state_dict().items()):
        assert k1 == k2 and v1.dtype == v2.dtype and torch.equal(v1, v2)


def test_cpu_input_fails_loudly():
    from unet import UNet
    net = UNet(3, 2)
    with pytest.raises(RuntimeError):
        net(torch.zeros(1, 3, 32, 32))


def test_hubconf_entry():
    import hubconf
    net = hubconf.unet_carvana(pretrained=False)
    assert (net.n_channels, net.n_classes, net.bilinear) == (3, 2, False)
    with pytest.raises(TypeError):
        net.use_checkpointing()


def test_trainer_fused_optimizer_matches_foreach_rmsprop():
    """Two Trainer steps with the fused clip+RMSprop (flat buffers) against torch's foreach RMSprop + flat clip on an
    identically initialised net: same parameters afterwards, parameters still the nn.Parameter objects of the module
    (state_dict contract), LR changes through param_groups are honoured."""
    from unet import UNet
    from unetk.trainer import Trainer
    img = cl(wave('trainer.img', (2, 3, 48, 64), 0.5, 0.5))
    mask = torch.from_numpy(W.int_wave('trainer.mask', (2, 48, 64), 2)).cuda()
    nets = [load_closed_form(UNet(3, 2, False)).cuda().train() for _ in range(2)]
    trainers = [Trainer(nets[0], amp=False, learning_rate=1e-3, fused_optimizer=True),
                Trainer(nets[1], amp=False, learning_rate=1e-3, fused_optimizer=False)]
    keys_before = [k for k, _ in nets[0].named_parameters()]
    for step in range(2):
        for t in trainers:
            t.optimizer.param_groups[0]['lr'] = 1e-3 if step == 0 else 5e-4
        losses = [float(t.step(img, mask)) for t in trainers]
        assert abs(losses[0] - losses[1]) <= 1e-5 * abs(losses[1])
    assert [k for k, _ in nets[0].named_parameters()] == keys_before
    pa, pb = dict(nets[0].named_parameters()), dict(nets[1].named_parameters())
    for k in pa:
        assert pa[k].shape == pb[k].shape
        # RMSprop's first steps move every element by ~lr/sqrt(1-alpha) * sign(g): elements whose gradient is at the
        # rounding noise of the two (atomically ordered) backward passes may step differently, so compare in L2
        # (the arithmetic itself is pinned element-wise by test_gpu_ops.py::test_clip_rmsprop_matches_torch)
        # ... and the second step's gradients inherit that difference; the losses above agree to 1e-5
        assert l2_err(npy(pa[k]), npy(pb[k])) < 5e-3, k
    assert float(trainers[0].optimizer.total_norm) > 0
    sd = nets[0].state_dict()
    assert sd['inc.double_conv.0.weight'].data_ptr() == nets[0].inc.double_conv[0].weight.data_ptr()


# ------------------------------------------------------------------------------------- validation round (8f row 1)
@pytest.mark.parametrize('dtype', [torch.float32, torch.bfloat16])
@pytest.mark.parametrize('n,c,h,w', [(2, 2, 17, 23), (3, 21, 31, 45), (1, 5, 64, 70)])
def test_eval_dice_matches_reference_composition(dtype, n, c, h, w):
    """unetk_eval_dice against the reference's one_hot/argmax/multiclass_dice_coeff chain (evaluate.py:31-36),
    including ties (first maximum wins), an image without foreground, and a class that is never predicted."""
    import torch.nn.functional as F
    from unetk import ops
    from utils.dice_score import multiclass_dice_coeff
    g = torch.Generator().manual_seed(n * 100 + c)
    logits = (torch.randn((n, c, h, w), generator=g) * 2).round().to(dtype).cuda()      # rounded: many exact ties
    logits = logits.contiguous(memory_format=torch.channels_last)
    mask = torch.randint(0, c, (n, h, w), generator=g).cuda()
    mask[0] = 0                                                                          # background only
    got = ops.eval_dice(logits, mask, first_class=1)
    mt = F.one_hot(mask, c).permute(0, 3, 1, 2).float()
    mp = F.one_hot(logits.float().argmax(dim=1), c).permute(0, 3, 1, 2).float()
    ref = multiclass_dice_coeff(mp[:, 1:].contiguous(), mt[:, 1:].contiguous(), reduce_batch_first=False)
    assert abs(float(got) - float(ref)) < 1e-6


@pytest.mark.parametrize('tag,ncls,shape', [('eval_c2', 2, (2, 3, 96, 142)), ('eval_c21', 21, (2, 3, 64, 80))])
def test_evaluate_matches_reference(tag, ncls, shape):
    from unet import UNet
    from evaluate import evaluate
    g = golden(tag)
    net = load_closed_form(UNet(3, ncls, False)).cuda().train()
    batches = [{'image': wave(f'{tag}.img{b}', shape, 0.5, 0.5),
                'mask': torch.from_numpy(W.int_wave(f'{tag}.mask{b}', (shape[0],) + shape[2:], ncls))}
               for b in range(int(g['n_batches']))]
    score = evaluate(net, batches, torch.device('cuda'), False)
    assert net.training
    # fp32 logits agree to ~1e-5 and 99 % of the argmax margins exceed margin_q01 >> that, so hardly a pixel can flip
    assert float(g['margin_q01']) > 1e-3
    assert abs(float(score) - float(g['dice'])) <= 2e-3 * max(float(g['dice']), 1e-3)
    bf = evaluate(net, batches, torch.device('cuda'), True)                              # bf16 autocast path runs
    assert 0.0 <= float(bf) <= 1.0


def test_evaluate_binary_net_asserts_like_the_reference():
    """evaluate.py:26-29 hands the un-squeezed [N,1,H,W] prediction to dice_coeff, whose size assertion fires
    (fixture eval_c1_behaviour records it from the reference itself)."""
    from unet import UNet
    from evaluate import evaluate
    assert str(golden('eval_c1_behaviour')['raised']) == 'AssertionError'
    net = load_closed_form(UNet(3, 1, False)).cuda()
    batch = [{'image': wave('eval_c1.img', (1, 3, 32, 48), 0.5, 0.5), 'mask': torch.zeros((1, 32, 48), dtype=torch.long)}]
    with pytest.raises(AssertionError):
        evaluate(net, batch, torch.device('cuda'), False)


# ------------------------------------------------------------------------------------- predict_img (8f row 3)
@pytest.mark.parametrize('dtype', [torch.float32, torch.bfloat16])
@pytest.mark.parametrize('n,c,h,w,oh,ow', [(1, 2, 35, 50, 70, 101), (2, 21, 16, 23, 40, 31), (1, 1, 20, 20, 57, 64),
                                           (1, 3, 30, 40, 15, 20)])
def test_predict_mask_matches_interpolate_argmax(dtype, n, c, h, w, oh, ow):
    """unetk_predict_mask against F.interpolate(bilinear) + argmax / sigmoid > 0.5 (predict.py:27-31) evaluated by
    torch on the CPU in fp32; pixels may differ only where the decision margin is at rounding level."""
    import torch.nn.functional as F
    from unetk import ops
    g = torch.Generator().manual_seed(7 * c + h)
    logits = torch.randn((n, c, h, w), generator=g).to(dtype)
    got = ops.predict_mask(logits.cuda().contiguous(memory_format=torch.channels_last), oh, ow, 0.5).cpu()
    up = F.interpolate(logits.float(), (oh, ow), mode='bilinear')
    if c > 1:
        ref = up.argmax(dim=1)
        top2 = up.topk(2, dim=1).values
        margin = top2[:, 0] - top2[:, 1]
    else:
        ref = (torch.sigmoid(up) > 0.5)[:, 0].long()
        margin = up.abs()[:, 0]
    bad = got != ref
    assert bad.float().mean() < 1e-3
    assert (margin[bad] < 1e-5).all()


@pytest.mark.parametrize('tag,ncls', [('predict_c2', 2), ('predict_c1', 1)])
def test_predict_img_matches_reference(tag, ncls):
    from PIL import Image
    from unet import UNet
    from predict import predict_img, mask_to_image
    from utils.data_loading import BasicDataset
    g = golden(tag)
    full_img = Image.fromarray(g['rgb'])
    pre = BasicDataset.preprocess(None, full_img, 0.5, is_mask=False)
    assert pre.shape == g['preprocessed'].shape and np.abs(pre - g['preprocessed']).max() < 1e-6
    net = load_closed_form(UNet(3, ncls, False)).cuda()
    mask = predict_img(net, full_img, torch.device('cuda'), scale_factor=0.5, out_threshold=0.5)
    assert mask.shape == g['mask'].shape == (70, 101) and mask.dtype == np.int64
    bad = mask != g['mask']
    assert bad.mean() < 2e-3
    assert (g['margin'][bad] < 1e-3).all()           # only near-ties of the fp32 logits (1e-5 relative) may flip
    img = mask_to_image(mask, [0, 1])
    assert img.size == (101, 70) and img.mode == '1'
    assert mask_to_image(mask, [[0, 0, 0], [255, 255, 255]]).mode == 'RGB'


def test_preprocess_mask_and_errors():
    from PIL import Image
    from utils.data_loading import BasicDataset
    m = np.zeros((10, 12), dtype=np.uint8)
    m[2:6, 3:9] = 255
    out = BasicDataset.preprocess([0, 255], Image.fromarray(m), 0.5, is_mask=True)
    assert out.shape == (5, 6) and out.dtype == np.int64 and set(np.unique(out)) == {0, 1}
    rgb = np.zeros((10, 12, 3), dtype=np.uint8)
    rgb[..., 0] = 200
    out = BasicDataset.preprocess([[0, 0, 0], [200, 0, 0]], Image.fromarray(rgb), 1.0, is_mask=True)
    assert (out == 1).all()
    with pytest.raises(AssertionError):
        BasicDataset.preprocess(None, Image.fromarray(m), 0.01, is_mask=False)


def test_pack_plan_equals_per_layer_packing():
    """blocks.pack_all (one launch for every conv weight) writes exactly what per-layer unetk_pack_weight writes, for
    both dtypes and with/without the dgrad streams, and follows in-place weight updates."""
    from unet import UNet
    from unetk import blocks, ops
    net = load_closed_form(UNet(3, 2, False)).cuda()
    for dtype in (torch.bfloat16, torch.float32):
        for need_dgrad in (False, True):
            blocks.pack_all(net, dtype, need_dgrad)
            n_checked = 0
            for m in net.modules():
                cache = m.__dict__.get('_unetk_pack', {})
                for (kind, dt), (ver, ptr, buf) in cache.items():
                    if dt != dtype:
                        continue
                    ref = ops.pack_weight(kind, m.weight, m.out_channels, m.in_channels, dtype)
                    assert ver == m.weight._version and ptr == m.weight.data_ptr()
                    assert torch.equal(buf, ref), (type(m).__name__, kind)
                    n_checked += 1
            assert n_checked >= (21 if not need_dgrad else 42)
    w = net.up1.up.weight
    with torch.no_grad():
        w.mul_(2.0)
    blocks.pack_all(net, torch.bfloat16, True)
    _, _, buf = net.up1.up.__dict__['_unetk_pack'][(2, torch.bfloat16)]
    assert torch.equal(buf, ops.pack_weight(2, w, net.up1.up.out_channels, net.up1.up.in_channels, torch.bfloat16))
