"""Full-size checks (BASELINE.json configs[1] layer shapes: 959x640 and its pyramid) through the C ABI.

The NumPy oracle would need minutes at these sizes, so the checks are size-independent properties instead:

* integer-valued operands: x, dy in {-1,0,1} and sparse weights in {-1,0,1} make every product and every partial
  sum an exactly representable integer (|sum| <= 256 for the bf16 outputs, < 2^24 for the fp32 weight gradients), so
  the MFMA kernels must agree BIT-EXACTLY with an independent evaluation of the same contraction (im2col + fp32
  GEMM by torch on the same GPU), whatever the tiling, split-K order or rounding point;
* the fused BatchNorm statistics must equal the exact integer sums of the stored output;
* pooling: max-pool forward equals torch's, and its backward routes every gradient to exactly one input.
"""
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

H16 = [torch.bfloat16, torch.float16]      # both 16-bit element types run the same kernels (csrc/common.h: H16<T>)


def _ints(shape, seed, density=1.0):
    g = torch.Generator().manual_seed(seed)
    v = torch.randint(-1, 2, shape, generator=g).float()
    if density < 1.0:
        v = v * (torch.rand(shape, generator=g) < density).float()
    return v


def _im2col(x_nhwc):
    """[1,H,W,C] fp32 -> [H*W, C*9] with (c, kh, kw) column order (F.unfold's)."""
    x = x_nhwc.permute(0, 3, 1, 2).float()
    return F.unfold(x, kernel_size=3, padding=1)[0].t()


CONV_SHAPES = [
    (1, 640, 959, 64, 64),       # L0, 64-channel tiles (NB=2)
    (1, 640, 959, 128, 64),      # L0 decoder conv on the concat input
    (1, 320, 479, 128, 128),     # 128-channel tiles (NB=4)
    (2, 40, 59, 1024, 1024),     # bottleneck: 8 n-tiles, 64 K-chunks, ragged 16x32 tiles
    (1, 1280, 1918, 64, 64),     # BASELINE.json configs[3] (1918x1280), L0: the widest in-image index range
    (1, 1280, 1918, 128, 64),    # ... its decoder conv on the concat input (pixel pitch 128)
]


@pytest.mark.parametrize('dt16', H16)
@pytest.mark.parametrize('n,h,w,cin,cout', CONV_SHAPES)
def test_conv3x3_fprop_stats_dgrad_wgrad_exact_on_integers(n, h, w, cin, cout, dt16):
    from unetk import ops
    dens = min(1.0, 24.0 / (9 * cin))                   # ~24 non-zero taps per output: |y| <= 256, exact in bf16 and fp16
    x = _ints((n, h, w, cin), 1).cuda()
    wt = _ints((cout, cin, 3, 3), 2, dens).cuda()
    xb = x.to(dt16)
    stats = ops.new_stats(cout, 'cuda')
    y = ops.conv3x3_fprop(xb, ops.pack_weight(0, wt, cout, cin, dt16), cout, stats=stats)
    wmat = wt.reshape(cout, cin * 9).t().contiguous()
    s1 = torch.zeros(cout, dtype=torch.float64, device='cuda')
    s2 = torch.zeros(cout, dtype=torch.float64, device='cuda')
    for i in range(n):
        ref = (_im2col(x[i:i + 1]) @ wmat).reshape(h, w, cout)
        assert ref.abs().max() <= 256
        assert torch.equal(y[i].float(), ref), 'fprop'
        s1 += ref.double().sum((0, 1))
        s2 += (ref.double() ** 2).sum((0, 1))
    got = stats.sum(0)
    assert torch.equal(got[:, 0], s1) and torch.equal(got[:, 1], s2), 'BatchNorm statistics'

    # dgrad: dx = conv(dy, rot180(w)^T)
    dy = _ints((n, h, w, cout), 3).cuda()
    dens_d = min(1.0, 24.0 / (9 * cout))
    wd = _ints((cout, cin, 3, 3), 4, dens_d).cuda()
    dx = ops.conv3x3_dgrad(dy.to(dt16), ops.pack_weight(1, wd, cout, cin, dt16), cin)
    wrot = wd.flip(2, 3).permute(1, 0, 2, 3).reshape(cin, cout * 9).t().contiguous()
    for i in range(n):
        ref = (_im2col(dy[i:i + 1]) @ wrot).reshape(h, w, cin)
        assert torch.equal(dx[i].float(), ref), 'dgrad'

    # wgrad: dw[co,ci,kh,kw] = sum_pixels x[p + tap][ci] * dy[p][co]  (integers < 2^24: exact in fp32)
    dw = torch.empty((cout, cin, 3, 3), device='cuda')
    ops.conv3x3_wgrad(xb, dy.to(dt16), dw)
    ref = torch.zeros((cout, cin * 9), device='cuda')
    for i in range(n):
        ref += dy[i].reshape(h * w, cout).t() @ _im2col(x[i:i + 1])
    assert ref.abs().max() < 2 ** 24
    assert torch.equal(dw.reshape(cout, cin * 9), ref), 'wgrad'


@pytest.mark.parametrize('dt16', H16)
@pytest.mark.parametrize('n,h,w,cin', [(1, 320, 479, 128), (2, 40, 59, 1024)])
def test_convT_fwd_dgrad_wgrad_exact_on_integers(n, h, w, cin, dt16):
    from unetk import ops
    cout = cin // 2
    hs, ws = 2 * h, 2 * w + 1                           # the reference's odd widths: one F.pad column on the right
    x = _ints((n, h, w, cin), 5).cuda()
    wt = _ints((cin, cout, 2, 2), 6, min(1.0, 64.0 / cin)).cuda()
    bias = _ints((cout,), 7).cuda()
    cat = torch.full((n, hs, ws, cin), 3.0, dtype=dt16, device='cuda')
    ops.convT2x2_fwd(x.to(dt16), ops.pack_weight(2, wt, cout, cin, dt16), bias, cat, cout, cout, 0, 0)
    ref = torch.empty((n, 2 * h, 2 * w, cout), device='cuda')          # independent evaluation: four GEMMs
    for a in (0, 1):
        for b in (0, 1):
            ref[:, a::2, b::2] = torch.einsum('nijc,ck->nijk', x, wt[:, :, a, b]) + bias
    assert ref.abs().max() <= 256
    assert torch.equal(cat[:, :, :2 * w, cout:].float(), ref), 'convT forward'
    assert torch.equal(cat[:, :, 2 * w:, cout:].float(), torch.zeros_like(cat[:, :, 2 * w:, cout:]).float()), 'pad'
    assert torch.equal(cat[..., :cout].float(), torch.full_like(cat[..., :cout], 3.0).float()), 'skip half untouched'

    dcat = _ints((n, hs, ws, cin), 8).cuda()
    wd = _ints((cin, cout, 2, 2), 9, min(1.0, 16.0 / cout)).cuda()
    dx = ops.convT2x2_dgrad(dcat.to(dt16), cout, ops.pack_weight(3, wd, cout, cin, dt16), cin, cout, h, w,
                            0, 0)
    dup = dcat[:, :, :2 * w, cout:].permute(0, 3, 1, 2).contiguous()
    ref = sum(torch.einsum('nkij,ck->nijc', dup[:, :, a::2, b::2], wd[:, :, a, b]) for a in (0, 1) for b in (0, 1))
    assert torch.equal(dx.float(), ref), 'convT dgrad'

    dw = torch.empty_like(wt)
    db = torch.empty(cout, device='cuda')
    ops.convT2x2_wgrad(x.to(dt16), dcat.to(dt16), cout, dw, db, 0, 0)
    # dw[ci,co,a,b] = sum_{n,i,j} x[n,i,j,ci] * dup[n,co,2i+a,2j+b]
    ref = torch.stack([torch.stack([torch.einsum('nijc,nkij->ck', x, dup[:, :, a::2, b::2]) for b in (0, 1)], -1)
                       for a in (0, 1)], -2)
    assert torch.equal(dw, ref), 'convT wgrad'
    assert torch.equal(db, dup.sum((0, 2, 3))), 'convT bias gradient'


def test_maxpool_full_size():
    from unetk import ops
    n, h, w, c = 2, 640, 959, 64
    g = torch.Generator().manual_seed(11)
    x = torch.rand((n, h, w, c), generator=g).cuda().bfloat16()
    out = ops.maxpool_fwd(x)
    ref = F.max_pool2d(x.float().permute(0, 3, 1, 2), 2).permute(0, 2, 3, 1)
    assert torch.equal(out.float(), ref)
    dpool = _ints((n, h // 2, w // 2, c), 12).cuda().bfloat16()
    dx = ops.maxpool_bwd(x, dpool)
    # every pooled gradient lands on exactly one input of its window; the dropped odd column gets none
    assert torch.equal(dx[:, :, :w - 1].float().reshape(n, h // 2, 2, w // 2, 2, c).sum((2, 4)), dpool.float())
    assert torch.equal(dx[:, :, w - 1:].float(), torch.zeros((n, h, 1, c), device='cuda'))
    picked = (dx != 0).float().reshape(n, h, w, c)[:, :, :w - 1].reshape(n, h // 2, 2, w // 2, 2, c).sum((2, 4))
    assert torch.equal(picked, (dpool != 0).float())


@pytest.mark.parametrize('n,hh,ww', [(2, 640, 959), (1, 1280, 1918)])
def test_unet_full_size_train_step_runs_and_is_finite(n, hh, ww):
    """One bf16 training step at the benchmark's image sizes (BASELINE.json configs[1] and configs[3]): shapes,
    finiteness, and the fused loss equals the reference's composition (train.py:102-111) evaluated by torch on the
    logits the network produced."""
    from unet import UNet
    from unetk.loss import segmentation_loss
    from utils.dice_score import dice_loss
    torch.manual_seed(0)
    net = UNet(3, 2, False).cuda().to(memory_format=torch.channels_last).train()
    img = torch.rand((n, 3, hh, ww), device='cuda').contiguous(memory_format=torch.channels_last)
    mask = (torch.rand((n, hh, ww), device='cuda') < 0.2).long()
    with torch.autocast('cuda', dtype=torch.bfloat16):
        logits = net(img)
    assert logits.shape == (n, 2, hh, ww)
    loss = segmentation_loss(logits, mask)
    loss.backward()
    lf = logits.detach().float()
    ref = F.cross_entropy(lf, mask) + dice_loss(F.softmax(lf, dim=1),
                                                F.one_hot(mask, 2).permute(0, 3, 1, 2).float(), multiclass=True)
    assert abs(float(loss.detach()) - float(ref)) < 1e-4
    for k, p in net.named_parameters():
        assert p.grad is not None and torch.isfinite(p.grad).all(), k
    bn = net.inc.double_conv[1]
    assert int(bn.num_batches_tracked) == 1 and torch.isfinite(bn.running_var).all()


@pytest.mark.parametrize('n,h,w,c', [(1, 320, 479, 128), (2, 40, 59, 1024), (2, 9, 11, 128)])
def test_convT_bias_gradient_from_dgrad_sums_exact_on_integers(n, h, w, c):
    """The ConvTranspose2d bias gradient taken from the fused channel sums of the conv3x3 dgrad that writes the concat
    gradient (minus its F.pad border): integer operands -> must equal the exact sum over the up-sampled window."""
    from unetk import ops
    cout = c // 2                                        # decoder conv: c (concat) -> cout; transposed conv: c -> c/2
    hs, ws = 2 * h + 1, 2 * w + 1                        # odd sizes: one pad row at the bottom, one column on the right
    dy = _ints((n, hs, ws, cout), 21).cuda()
    wd = _ints((cout, c, 3, 3), 22, min(1.0, 24.0 / (9 * cout))).cuda()
    sums = ops.new_stats(c, 'cuda')
    dcat = ops.conv3x3_dgrad(dy.bfloat16(), ops.pack_weight(1, wd, cout, c, torch.bfloat16), c, sums=sums)
    assert dcat.float().abs().max() <= 256
    x = _ints((n, h, w, c), 23).cuda().bfloat16()
    dw = torch.empty((c, c // 2, 2, 2), device='cuda')
    db = torch.empty(c // 2, device='cuda')
    ops.convT2x2_wgrad(x, dcat, c // 2, dw, db, 0, 0, sums=sums)
    ref = dcat[:, :2 * h, :2 * w, c // 2:].float().sum((0, 1, 2))
    assert torch.equal(db, ref)
    db2 = torch.empty_like(db)
    ops.convT2x2_wgrad(x, dcat, c // 2, dw, db2, 0, 0)
    assert torch.equal(db2, ref)


def test_bilinear_batch64_inference_fused_epilogue_matches_unfused():
    """BASELINE.json configs[2]: UNet(bilinear=True).eval(), bf16, batch 64 at 959x640.  The inference path folds
    eval-mode BatchNorm + ReLU (+ the encoder's max-pool) into the conv epilogue (unetk_conv3x3_fprop_bn_relu); the
    same net through the unfused kernels (conv -> bn_relu_apply -> maxpool, the bf16 pre-BN tensor materialised as in
    the reference) must give the same masks and logits within bf16 rounding."""
    from unet import UNet
    from unetk import blocks
    torch.manual_seed(0)
    net = UNet(3, 2, True).cuda().to(memory_format=torch.channels_last)
    g = torch.Generator(device='cuda').manual_seed(5)
    img = torch.rand((64, 3, 640, 959), device='cuda', generator=g).contiguous(memory_format=torch.channels_last)
    net.train()
    with torch.no_grad(), torch.autocast('cuda', dtype=torch.bfloat16):
        net(img[:2])                                    # running statistics away from their (0, 1) initial values
    net.eval()
    with torch.inference_mode(), torch.autocast('cuda', dtype=torch.bfloat16):
        fused = net(img)
        assert blocks.FUSE_EVAL_EPILOGUE
        blocks.FUSE_EVAL_EPILOGUE = False
        try:
            plain = net(img)
        finally:
            blocks.FUSE_EVAL_EPILOGUE = True
    assert fused.shape == (64, 2, 640, 959) and fused.dtype == torch.bfloat16
    f, p = fused.float(), plain.float()
    assert torch.isfinite(f).all()
    assert float((f - p).abs().max() / p.abs().max()) <= 2e-2
    # where the two classes are further apart than bf16 noise the masks agree; overall agreement stays >= 99.9 %
    agree = float((f.argmax(1) == p.argmax(1)).float().mean())
    margin = (p[:, 0] - p[:, 1]).abs()
    clear = margin > 2e-2 * p.abs().max()
    assert bool((f.argmax(1) == p.argmax(1))[clear].all())
    assert agree >= 0.999 or float(clear.float().mean()) < 0.999, agree


def _ints_cuda(shape, seed):
    g = torch.Generator(device='cuda').manual_seed(seed)
    return torch.randint(-1, 2, shape, generator=g, device='cuda', dtype=torch.int8).bfloat16()


@pytest.mark.parametrize('cin,cout', [(128, 64), (64, 64)])
def test_batch8_1918x1280_image7_equals_the_same_image_alone(cin, cout):
    """BASELINE.json configs[3] at its per-GPU batch: [8, 1280, 1918, 128] holds 2.51e9 elements (> 2^31) and
    19.6 M pixels, so every per-image base, pixel index and byte offset is exercised beyond 32 bits.  Integer operands:
    the last image of the batch must come out BIT-EXACTLY as when it is run alone (and as the im2col evaluation, which
    the N = 1 cases above pin), for fprop + statistics, dgrad (whose 128-channel output is the > 2^31 tensor in the
    128 -> 64 case), BatchNorm+ReLU apply and its backward apply."""
    from unetk import ops, _native as nat
    n, h, w = 8, 1280, 1918
    x = _ints_cuda((n, h, w, cin), 31)
    assert cin < 128 or x.numel() > 2 ** 31
    wt = _ints((cout, cin, 3, 3), 32, min(1.0, 24.0 / (9 * cin))).cuda()
    wpk = ops.pack_weight(0, wt, cout, cin, torch.bfloat16)
    stats = ops.new_stats(cout, 'cuda')
    y = ops.conv3x3_fprop(x, wpk, cout, stats=stats)
    s_ref = torch.zeros((cout, 2), dtype=torch.float64, device='cuda')
    for i in (0, 3, 7):
        st1 = ops.new_stats(cout, 'cuda')
        y1 = ops.conv3x3_fprop(x[i:i + 1], wpk, cout, stats=st1)
        assert torch.equal(y[i], y1[0]), f'fprop image {i}'
    ref = (_im2col(x[7:8].float()) @ wt.reshape(cout, cin * 9).t().contiguous()).reshape(h, w, cout)
    assert torch.equal(y[7].float(), ref), 'fprop image 7 vs im2col'
    del ref
    yf = y.view(-1, cout)
    for lo in range(0, yf.shape[0], 1 << 22):            # exact integer sums of the stored output, in slices
        blk = yf[lo:lo + (1 << 22)].double()
        s_ref[:, 0] += blk.sum(0)
        s_ref[:, 1] += (blk * blk).sum(0)
    assert torch.equal(stats.sum(0), s_ref), 'BatchNorm statistics over all 8 images'

    # dgrad: the gradient tensor has cout channels, the result cin
    dy = _ints_cuda((n, h, w, cout), 33)
    wd = _ints((cout, cin, 3, 3), 34, min(1.0, 24.0 / (9 * cout))).cuda()
    wpd = ops.pack_weight(1, wd, cout, cin, torch.bfloat16)
    dx = ops.conv3x3_dgrad(dy, wpd, cin)
    for i in (0, 7):
        assert torch.equal(dx[i], ops.conv3x3_dgrad(dy[i:i + 1], wpd, cin)[0]), f'dgrad image {i}'
    wrot = wd.flip(2, 3).permute(1, 0, 2, 3).reshape(cin, cout * 9).t().contiguous()
    assert torch.equal(dx[7].float(), (_im2col(dy[7:8].float()) @ wrot).reshape(h, w, cin)), 'dgrad image 7 vs im2col'
    # wgrad over the vertically stacked batch (8 x 1281 virtual rows): integer sums < 2^24, so the batch's gradient is
    # EXACTLY the sum of the eight single-image gradients
    dw = torch.empty((cout, cin, 3, 3), device='cuda')
    ops.conv3x3_wgrad(x, dy, dw)
    acc = torch.zeros_like(dw)
    dw1 = torch.empty_like(dw)
    for i in range(n):
        ops.conv3x3_wgrad(x[i:i + 1], dy[i:i + 1], dw1)
        acc += dw1
    assert acc.abs().max() < 2 ** 24
    assert torch.equal(dw, acc), 'wgrad over 8 images vs sum of single-image wgrads'
    del dy, y, yf

    # BatchNorm + ReLU forward apply and backward apply on the [8, 1280, 1918, cin] tensors, fixed coefficients
    g = torch.Generator().manual_seed(35)
    st = ops.BNState(cin, 'cuda')
    st.buf.copy_(torch.rand((4, cin), generator=g) + 0.5)
    st.shift.sub_(1.0)
    a = ops.bn_relu_apply(dx, st)
    assert torch.equal(a[7], ops.bn_relu_apply(dx[7:8], st)[0]), 'bn_relu_apply image 7'
    assert torch.equal(a[7].float(), torch.relu(dx[7].float() * st.scale + st.shift).bfloat16().float())
    coef = (torch.rand((3, cin), generator=g) + 0.25).cuda()
    dz = torch.empty_like(dx)
    dz1 = torch.empty_like(dx[7:8])
    for dout, z, out in ((x if cin == 128 else a, dx, dz), ((x if cin == 128 else a)[7:8], dx[7:8], dz1)):
        nn_, hh_, ww_, c_, ld = ops.nhwc_meta(z, 'z')
        nat.check(nat.fn['unetk_bn_relu_bwd_apply'](dout.data_ptr(), ops.nhwc_meta(dout, 'd')[4], z.data_ptr(), ld,
                                                    st.scale.data_ptr(), st.shift.data_ptr(), st.mean.data_ptr(),
                                                    coef.data_ptr(), out.data_ptr(), c_, nn_ * hh_ * ww_, c_, nat.BF16,
                                                    ops.stream_ptr()), 'unetk_bn_relu_bwd_apply')
    assert torch.equal(dz[7], dz1[0]), 'bn_relu_bwd_apply image 7'
    assert bool((dz[7] != 0).any())
    # the reduction pass over all 8 images = the sum of the eight single-image passes (a workgroup's partial sums are
    # fp32, the accumulators across workgroups fp64: agreement to fp32 rounding, a wrong image would be off by O(1))
    dout = x if cin == 128 else a
    sums = ops.new_stats(cin, 'cuda')
    one = torch.zeros((cin, 2), dtype=torch.float64, device='cuda')
    npix1 = h * w
    nat.check(nat.fn['unetk_bn_relu_bwd_reduce'](dout.data_ptr(), cin, dx.data_ptr(), cin, st.scale.data_ptr(),
                                                 st.shift.data_ptr(), st.mean.data_ptr(), st.invstd.data_ptr(),
                                                 sums.data_ptr(), n * npix1, cin, nat.BF16, ops.stream_ptr()), 'reduce')
    for i in range(n):
        s1 = ops.new_stats(cin, 'cuda')
        nat.check(nat.fn['unetk_bn_relu_bwd_reduce'](dout[i:i + 1].data_ptr(), cin, dx[i:i + 1].data_ptr(), cin,
                                                     st.scale.data_ptr(), st.shift.data_ptr(), st.mean.data_ptr(),
                                                     st.invstd.data_ptr(), s1.data_ptr(), npix1, cin, nat.BF16,
                                                     ops.stream_ptr()), 'reduce')
        one += s1.sum(0)
    got = sums.sum(0)
    assert float((got - one).abs().max()) <= 1e-4 * float(one.abs().max())


def test_cfg4_21_classes_fp32_512_batch16_train_step():
    """BASELINE.json configs[4] at full size: UNet(3, 21) fp32 (no autocast), one train step on [16, 3, 512, 512]:
    the fused multi-class CE + dice loss equals the reference's composition (train.py:102-111) evaluated by torch on
    the logits the network produced, every gradient is finite, and the 21-class head went through conv1x1 kernels."""
    from unet import UNet
    from unetk.loss import segmentation_loss
    from utils.dice_score import dice_loss
    torch.manual_seed(0)
    n, hh, ww, ncls = 16, 512, 512, 21
    net = UNet(3, ncls, False).cuda().to(memory_format=torch.channels_last).train()
    g = torch.Generator(device='cuda').manual_seed(7)
    img = torch.rand((n, 3, hh, ww), device='cuda', generator=g).contiguous(memory_format=torch.channels_last)
    mask = torch.randint(0, ncls, (n, hh, ww), device='cuda', generator=g)
    logits = net(img)
    assert logits.shape == (n, ncls, hh, ww) and logits.dtype == torch.float32
    loss = segmentation_loss(logits, mask)
    loss.backward()
    lf = logits.detach()
    ref = F.cross_entropy(lf, mask) + dice_loss(F.softmax(lf, dim=1),
                                                F.one_hot(mask, ncls).permute(0, 3, 1, 2).float(), multiclass=True)
    assert abs(float(loss.detach()) - float(ref)) < 1e-4 * max(1.0, abs(float(ref)))
    for k, p in net.named_parameters():
        assert p.grad is not None and torch.isfinite(p.grad).all(), k
        assert float(p.grad.abs().max()) > 0, k
    # the gradient of the loss w.r.t. the logits agrees with autograd through the reference's composition
    lr = lf.clone().requires_grad_(True)
    (F.cross_entropy(lr, mask) + dice_loss(F.softmax(lr, dim=1), F.one_hot(mask, ncls).permute(0, 3, 1, 2).float(),
                                           multiclass=True)).backward()
    from unetk import ops
    out4 = ops.seg_loss_fwd(ops.to_nhwc(lf), mask)
    out4 = out4[1] if isinstance(out4, tuple) else out4
    dl = ops.seg_loss_bwd(ops.to_nhwc(lf), mask, out4)
    got = ops.to_nchw(dl)
    assert float((got - lr.grad).abs().max()) <= 1e-4 * float(lr.grad.abs().max())
