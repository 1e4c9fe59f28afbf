"""GPU parity of the MFMA conv3x3 kernels (fprop, fused BN statistics, dgrad) through the C ABI,
against the fp64 NumPy oracle (oracle/ops_numpy.py) on the same seeded inputs."""
import numpy as np
import pytest
import torch

from conftest import rel_err

pytestmark = pytest.mark.gpu

# fp32: BASELINE.json's 1e-4 relative; bf16: 2e-2 relative (inputs are pre-rounded to bf16, so the
# only error left is accumulation order + one output rounding, far inside that budget).
# fp16: three more mantissa bits than bf16 -> 3e-3.
TOL = {torch.float32: 1e-4, torch.bfloat16: 2e-2, torch.float16: 3e-3}
ALL = [torch.float32, torch.bfloat16, torch.float16]


def _rand(shape, seed, scale=1.0):
    g = torch.Generator().manual_seed(seed)
    return (torch.rand(shape, generator=g) * 2 - 1) * scale


@pytest.mark.parametrize('dtype', ALL)
@pytest.mark.parametrize('n,h,w,cin,cout', [
    (1, 8, 32, 64, 64),        # exactly one tile
    (2, 16, 23, 64, 128),      # ragged width (the 959 -> 479 -> ... case), two n-tiles
    (1, 13, 70, 128, 64),      # ragged height and width, 3 column tiles
    (1, 5, 3, 64, 64),         # image smaller than a tile
])
def test_conv3x3_fprop_and_stats(dtype, n, h, w, cin, cout):
    from oracle import ops_numpy as O
    from unetk import ops
    x = _rand((n, h, w, cin), 1).to(dtype)
    wt = _rand((cout, cin, 3, 3), 2, (3.0 / (9 * cin)) ** 0.5)
    wq = wt.to(dtype).float()
    xd = x.cuda()
    wpk = ops.pack_weight(0, wq.cuda(), cout, cin, dtype)
    stats = ops.new_stats(cout, 'cuda')
    y = ops.conv3x3_fprop(xd, wpk, cout, stats=stats)
    torch.cuda.synchronize()
    ref = O.conv3x3_fwd(x.double().numpy(), wq.double().numpy())
    got = y.float().cpu().numpy()
    assert rel_err(got, ref) < TOL[dtype]
    # fused statistics are taken on the stored (rounded) outputs
    s = stats.sum(0).cpu().numpy()
    g64 = got.astype(np.float64).reshape(-1, cout)
    assert np.allclose(s[:, 0], g64.sum(0), rtol=1e-5, atol=1e-4)
    assert np.allclose(s[:, 1], (g64 * g64).sum(0), rtol=1e-5, atol=1e-4)


@pytest.mark.parametrize('dtype', ALL)
def test_conv3x3_fprop_channel_slices(dtype):
    """Input read from, and output written into, channel slices of wider NHWC buffers (the concat trick)."""
    from oracle import ops_numpy as O
    from unetk import ops
    n, h, w, cin, cout = 1, 9, 33, 64, 64
    xb = _rand((n, h, w, 192), 3).to(dtype).cuda()
    yb = torch.full((n, h, w, 128), 7.0, dtype=dtype, device='cuda')
    wq = _rand((cout, cin, 3, 3), 4, 0.05).to(dtype).float()
    wpk = ops.pack_weight(0, wq.cuda(), cout, cin, dtype)
    ops.conv3x3_fprop(xb[..., 64:128], wpk, cout, out=yb[..., 64:128])
    torch.cuda.synchronize()
    ref = O.conv3x3_fwd(xb[..., 64:128].double().cpu().numpy(), wq.double().numpy())
    assert rel_err(yb[..., 64:128].float().cpu().numpy(), ref) < TOL[dtype]
    assert torch.all(yb[..., :64] == 7.0) and torch.all(yb[..., 128:] == 7.0)   # neighbours untouched


@pytest.mark.parametrize('dtype', ALL)
@pytest.mark.parametrize('n,h,w,cin,cout', [(2, 16, 23, 64, 128), (1, 9, 40, 128, 64)])
def test_conv3x3_dgrad(dtype, n, h, w, cin, cout):
    from oracle import ops_numpy as O
    from unetk import ops
    dy = _rand((n, h, w, cout), 5).to(dtype)
    wq = _rand((cout, cin, 3, 3), 6, 0.05).to(dtype).float()
    wpk = ops.pack_weight(1, wq.cuda(), cout, cin, dtype)
    dx = ops.conv3x3_dgrad(dy.cuda(), wpk, cin)
    torch.cuda.synchronize()
    ref, _ = O.conv3x3_bwd(np.zeros((n, h, w, cin)), wq.double().numpy(), dy.double().numpy())
    assert rel_err(dx.float().cpu().numpy(), ref) < TOL[dtype]


def test_conv3x3_rejects_bad_shapes():
    from unetk import ops
    x = torch.zeros((1, 8, 8, 48), dtype=torch.bfloat16, device='cuda')
    with pytest.raises(RuntimeError):
        ops.conv3x3_fprop(x, torch.zeros(9 * 48 * 64, dtype=torch.bfloat16, device='cuda'), 64)
    with pytest.raises(RuntimeError):
        ops.conv3x3_fprop(torch.zeros((1, 8, 8, 64), dtype=torch.bfloat16), torch.zeros(9 * 64 * 64, dtype=torch.bfloat16), 64)


class _State:
    """The four BatchNorm rows the backward kernels read (ops.BNState without the module plumbing)."""

    def __init__(self, c, seed):
        g = torch.Generator().manual_seed(seed)
        self.mean = (torch.rand(c, generator=g) - 0.5).cuda()
        self.invstd = (0.5 + torch.rand(c, generator=g)).cuda()
        gamma = (0.5 + torch.rand(c, generator=g)) * torch.where(torch.rand(c, generator=g) < 0.2, -1.0, 1.0)
        beta = (torch.rand(c, generator=g) - 0.5) * 0.5
        self.scale = (gamma.cuda() * self.invstd)
        self.shift = (beta.cuda() - self.mean * self.scale)


@pytest.mark.parametrize('dtype', ALL)
@pytest.mark.parametrize('n,h,w,cin,cout', [
    (1, 16, 32, 64, 64),        # one tile of the 64-channel kernel (bf16: the pipelined k32 kernel, weights resident)
    (2, 37, 75, 64, 64),        # ragged both ways, several tiles per workgroup
    (1, 21, 45, 128, 128),      # the 128-channel n-tile kernel
    (2, 9, 33, 256, 128),       # GEMM-N = 256: two n-tiles, GEMM-K = 128
    (1, 19, 50, 64, 128),       # GEMM-N = 64, GEMM-K = 128: weights streamed (not the UNet's case, but the kernel's)
    (1, 5, 3, 64, 64),          # image smaller than a tile
])
def test_conv3x3_dgrad_with_bn_reduce_of_the_producer_layer(dtype, n, h, w, cin, cout):
    """unetk_conv3x3_dgrad_bn_reduce == unetk_conv3x3_dgrad followed by unetk_bn_relu_bwd_reduce: the gradient
    bit for bit, the two per-channel sums to fp32 summation-order accuracy."""
    from unetk import ops
    from unetk._native import fn, check, dtype_code, stream_ptr
    dy = _rand((n, h, w, cout), 11).to(dtype).cuda()
    wq = _rand((cout, cin, 3, 3), 12, 0.05).to(dtype).float()
    wpk = ops.pack_weight(1, wq.cuda(), cout, cin, dtype)
    z = _rand((n, h, w, cin), 13, 2.0).to(dtype).cuda()
    st = _State(cin, 14)
    ref_dx = ops.conv3x3_dgrad(dy, wpk, cin)
    ref_sums = ops.new_stats(cin, 'cuda')
    check(fn['unetk_bn_relu_bwd_reduce'](ref_dx.data_ptr(), cin, z.data_ptr(), cin, st.scale.data_ptr(),
                                         st.shift.data_ptr(), st.mean.data_ptr(), st.invstd.data_ptr(),
                                         ref_sums.data_ptr(), n * h * w, cin, dtype_code(dtype), stream_ptr()), 'reduce')
    dx, sums = ops.conv3x3_dgrad(dy, wpk, cin, bn_prev=(z, st))
    torch.cuda.synchronize()
    assert torch.equal(dx, ref_dx)
    a, b = sums.sum(0).cpu().numpy(), ref_sums.sum(0).cpu().numpy()
    # scale of a channel's sums: sum |dz|, sum |dz xhat| -- cancellation must not hide an error
    d = ref_dx.double().reshape(-1, cin)
    zz = z.double().reshape(-1, cin)
    m = (zz * st.scale.double() + st.shift.double() > 0)
    mag1 = (d.abs() * m).sum(0).cpu().numpy()
    mag2 = ((d * (zz - st.mean.double()) * st.invstd.double()).abs() * m).sum(0).cpu().numpy()
    assert np.all(np.abs(a[:, 0] - b[:, 0]) <= 2e-5 * mag1 + 1e-6)
    assert np.all(np.abs(a[:, 1] - b[:, 1]) <= 2e-5 * mag2 + 1e-6)
    assert mag1.min() > 0 and np.abs(b).max() > 0
    # ... and the masked sums themselves against fp64 (the stand-alone pass has its own tests; this pins the fused one)
    e1 = (d * m).sum(0).cpu().numpy()
    e2 = (d * m * (zz - st.mean.double()) * st.invstd.double()).sum(0).cpu().numpy()
    assert np.all(np.abs(a[:, 0] - e1) <= 2e-5 * mag1 + 1e-6)
    assert np.all(np.abs(a[:, 1] - e2) <= 2e-5 * mag2 + 1e-6)


@pytest.mark.parametrize('cin,cmid', [(128, 64), (64, 64)])
def test_double_conv_backward_same_with_and_without_fused_bn_reduce(cin, cmid):
    """blocks.double_conv_bwd with FUSE_BN_REDUCE on/off: identical input gradient, BN/conv gradients to summation
    order.  The block's mid channel count is 64 -- the shape whose second conv's dgrad really carries the reduction
    (asserted), so the two runs take different kernels."""
    from unetk import blocks, ops
    from unet.unet_parts import DoubleConv
    torch.manual_seed(3)
    dc = DoubleConv(cin, cmid).cuda().train()
    assert ops.dgrad_bn_reduce_is_fused(cmid, cmid, torch.bfloat16)
    x = _rand((2, 26, 37, cin), 21).to(torch.bfloat16).cuda()
    da = _rand((2, 26, 37, cmid), 22).to(torch.bfloat16).cuda()
    outs = []
    before = blocks.FUSE_BN_REDUCE
    for fuse in (False, True):
        blocks.FUSE_BN_REDUCE = fuse
        try:
            a2, saved = blocks.double_conv_fwd(dc, x, torch.bfloat16)
            grads = {}
            dx = blocks.double_conv_bwd(dc, saved, da, grads, True)
            torch.cuda.synchronize()
            outs.append((dx.clone(), {k: v.clone() for k, v in grads.items()}))
        finally:
            blocks.FUSE_BN_REDUCE = before
    (dx0, g0), (dx1, g1) = outs
    assert float((dx0.float() - dx1.float()).abs().max()) <= 1e-2 * float(dx0.float().abs().max())
    for p in g0:
        assert float((g0[p] - g1[p]).abs().max()) <= 2e-3 * float(g0[p].abs().max()) + 1e-7


@pytest.mark.parametrize('dtype', ALL)
@pytest.mark.parametrize('n,h,w,cin,hs,ws', [
    (1, 8, 32, 128, 16, 64),      # exactly one 256-pixel tile
    (2, 13, 21, 128, 27, 43),     # ragged last tile, odd skip size (F.pad offsets), GEMM-K = 4 * 64
    (1, 20, 40, 256, 40, 80),     # two n-tiles, several tiles per workgroup
    (3, 5, 7, 512, 10, 14),       # less than one tile in all; four n-tiles
    (2, 6, 5, 64, 12, 10),        # GEMM-N not a multiple of 128: no fused kernel, the two-kernel path of the same call
])
def test_convT_dgrad_with_bn_reduce_of_the_layer_below(dtype, n, h, w, cin, hs, ws):
    """unetk_convT2x2_dgrad_bn_reduce == unetk_convT2x2_dgrad followed by unetk_bn_relu_bwd_reduce: the gradient bit
    for bit, the two per-channel sums to fp32 summation-order accuracy and against fp64."""
    from unetk import ops
    from unetk._native import fn, check, dtype_code, stream_ptr
    cout = 64 if cin == 64 else cin // 2
    pt, pl = (hs - 2 * h) // 2, (ws - 2 * w) // 2
    dcat = _rand((n, hs, ws, cin), 31).to(dtype).cuda()                  # [skip | up] gradient, up half = channels cout..
    wq = _rand((cin, cout, 2, 2), 32, 0.05).to(dtype).float()
    wpk = ops.pack_weight(3, wq.cuda(), cout, cin, dtype)
    z = _rand((n, h, w, cin), 33, 2.0).to(dtype).cuda()
    st = _State(cin, 34)
    assert ops.convT_dgrad_bn_reduce_is_fused(cin, cout, dtype) == (dtype != torch.float32 and cin % 128 == 0)
    ref_dx = ops.convT2x2_dgrad(dcat, cin - cout, wpk, cin, cout, h, w, pt, pl)
    ref_sums = ops.new_stats(cin, 'cuda')
    check(fn['unetk_bn_relu_bwd_reduce'](ref_dx.data_ptr(), cin, z.data_ptr(), cin, st.scale.data_ptr(),
                                         st.shift.data_ptr(), st.mean.data_ptr(), st.invstd.data_ptr(),
                                         ref_sums.data_ptr(), n * h * w, cin, dtype_code(dtype), stream_ptr()), 'reduce')
    dx, sums = ops.convT2x2_dgrad(dcat, cin - cout, wpk, cin, cout, h, w, pt, pl, bn_prev=(z, st))
    torch.cuda.synchronize()
    assert torch.equal(dx, ref_dx)
    a, b = sums.sum(0).cpu().numpy(), ref_sums.sum(0).cpu().numpy()
    d = ref_dx.double().reshape(-1, cin)
    zz = z.double().reshape(-1, cin)
    m = (zz * st.scale.double() + st.shift.double() > 0)
    mag1 = (d.abs() * m).sum(0).cpu().numpy()
    mag2 = ((d * (zz - st.mean.double()) * st.invstd.double()).abs() * m).sum(0).cpu().numpy()
    assert np.all(np.abs(a[:, 0] - b[:, 0]) <= 2e-5 * mag1 + 1e-6)
    assert np.all(np.abs(a[:, 1] - b[:, 1]) <= 2e-5 * mag2 + 1e-6)
    assert mag1.min() > 0 and np.abs(b).max() > 0
    e1 = (d * m).sum(0).cpu().numpy()
    e2 = (d * m * (zz - st.mean.double()) * st.invstd.double()).sum(0).cpu().numpy()
    assert np.all(np.abs(a[:, 0] - e1) <= 2e-5 * mag1 + 1e-6)
    assert np.all(np.abs(a[:, 1] - e2) <= 2e-5 * mag2 + 1e-6)


def test_whole_net_backward_same_with_and_without_the_convT_bn_reduce():
    """UNet(3, 2) bf16 train-mode backward with blocks.FUSE_BN_REDUCE_UP on/off: same loss, parameter gradients to
    summation order (the fused epilogue replaces four stand-alone reduction passes)."""
    from unetk import blocks
    from unet import UNet
    torch.manual_seed(5)
    net = UNet(3, 2, False).cuda().to(memory_format=torch.channels_last).train()
    x = _rand((2, 3, 48, 80), 41).cuda().contiguous(memory_format=torch.channels_last)
    target = (torch.rand((2, 48, 80), generator=torch.Generator().manual_seed(42)) < 0.3).long().cuda()
    outs = []
    before = blocks.FUSE_BN_REDUCE_UP
    for fuse in (False, True):
        blocks.FUSE_BN_REDUCE_UP = fuse
        try:
            net.zero_grad(set_to_none=True)
            with torch.autocast('cuda', dtype=torch.bfloat16):
                loss = torch.nn.functional.cross_entropy(net(x).float(), target)
            loss.backward()
            torch.cuda.synchronize()
            outs.append((float(loss), {k: p.grad.clone() for k, p in net.named_parameters()}))
        finally:
            blocks.FUSE_BN_REDUCE_UP = before
    (l0, g0), (l1, g1) = outs
    assert abs(l0 - l1) <= 1e-5 * abs(l0)
    # the sums differ in fp32 summation order -> last-bit differences of the BN-backward coefficients -> single bf16
    # roundings of the gradients below them: L2-relative (1 ulp of bf16 is 4e-3..8e-3 of an element)
    # (bias gradients are sums with heavy cancellation over the few pixels of this small image: 0.9 % measured)
    worst = max((float((g0[k] - g1[k]).norm() / (g0[k].norm() + 1e-30)), k) for k in g0)
    assert worst[0] <= 2e-2, worst
    for k in g0:
        assert float((g0[k] - g1[k]).abs().max()) <= 3e-2 * float(g0[k].abs().max()) + 1e-7, k
