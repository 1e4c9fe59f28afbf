"""GPU parity, through the C ABI, of every non-conv3x3-fprop kernel of the hot path against the fp64 NumPy
oracle (oracle/ops_numpy.py): wgrad, first layer, transposed conv (+pad+concat), BN+ReLU fwd/bwd, max-pool,
bilinear up-sampling, 1x1 head, loss and dice.  Inputs are seeded and pre-rounded to the kernel dtype."""
import numpy as np
import pytest
import torch

from conftest import rel_err

pytestmark = pytest.mark.gpu

# BASELINE.json north_star tolerances (fp32 1e-4, bf16 2e-2); fp16 carries three more mantissa bits than bf16: 3e-3
TOL = {torch.float32: 1e-4, torch.bfloat16: 2e-2, torch.float16: 3e-3}
BOTH = [torch.float32, torch.bfloat16, torch.float16]
H16 = [torch.bfloat16, torch.float16]


def tol(dtype, t16, t32):
    """Bound for a result rounded once to `dtype`: t16 is the bf16 figure, fp16 gets a sixth of it, fp32 t32."""
    return {torch.bfloat16: t16, torch.float16: t16 / 6}.get(dtype, t32)


def _rand(shape, seed, scale=1.0, offset=0.0):
    g = torch.Generator().manual_seed(seed)
    return (torch.rand(shape, generator=g) * 2 - 1) * scale + offset


def _np(t):
    return t.detach().double().cpu().numpy()


# --------------------------------------------------------------------------------------- wgrad
# (h >= 33 with c_out % 128 == 0: the 16x16x32 kernel, whose 32-row pixel columns run over the vertically stacked batch:
#  34 virtual rows per image is its smallest case, 3 x 34 = 102 rows leave a ragged last tile, w = 7 a ragged column)
@pytest.mark.parametrize('n,h,w,cin,cout', [(1, 8, 32, 64, 64), (2, 16, 23, 64, 128), (3, 13, 70, 128, 64),
                                            (3, 33, 7, 64, 128), (2, 45, 18, 128, 256), (5, 40, 9, 64, 128),
                                            (2, 35, 10, 128, 64),     # ... with the operands' roles swapped
                                            (3, 37, 11, 64, 64), (1, 70, 6, 192, 64),     # ... its 64 x 64 variant
                                            (1, 33, 1, 128, 128), (2, 34, 3, 64, 64)])   # narrower than one 4-column tile
@pytest.mark.parametrize('dt16', H16)
def test_conv3x3_wgrad_bf16(n, h, w, cin, cout, dt16):
    from oracle import ops_numpy as O
    from unetk import ops
    x = _rand((n, h, w, cin), 11).to(dt16)
    dy = _rand((n, h, w, cout), 12).to(dt16)
    dw = torch.full((cout, cin, 3, 3), 9.0, device='cuda')
    ops.conv3x3_wgrad(x.cuda(), dy.cuda(), dw)
    _, ref = O.conv3x3_bwd(_np(x), np.zeros((cout, cin, 3, 3)), _np(dy))
    assert rel_err(_np(dw), ref) < 1e-4        # bf16 inputs are exact in both; only fp32 accumulation differs
    dw2 = torch.empty_like(dw)
    ops.conv3x3_wgrad(x.cuda(), dy.cuda(), dw2)
    assert torch.equal(dw, dw2)                 # fixed-order split-K reduction: bitwise reproducible


def test_conv3x3_wgrad_channel_slices():
    from oracle import ops_numpy as O
    from unetk import ops
    xb = _rand((2, 9, 33, 192), 13).bfloat16().cuda()
    dyb = _rand((2, 9, 33, 128), 14).bfloat16().cuda()
    dw = torch.empty((64, 128, 3, 3), device='cuda')
    ops.conv3x3_wgrad(xb[..., 64:], dyb[..., 64:], dw)
    _, ref = O.conv3x3_bwd(_np(xb[..., 64:]), np.zeros((64, 128, 3, 3)), _np(dyb[..., 64:]))
    assert rel_err(_np(dw), ref) < 1e-4


# --------------------------------------------------------------------------------------- first layer
@pytest.mark.parametrize('dtype', BOTH)
@pytest.mark.parametrize('n,h,w,cin,cout', [(2, 16, 23, 3, 64), (1, 9, 40, 1, 64), (1, 8, 32, 4, 128), (1, 1, 1, 3, 64)])
def test_first_conv_input_gradient(dtype, n, h, w, cin, cout):
    """unetk_conv3x3_first_dgrad against the NumPy oracle's conv3x3 input gradient."""
    from oracle import ops_numpy as O
    from unetk import ops
    dy = _rand((n, h, w, cout), 25).to(dtype)
    wt = _rand((cout, cin, 3, 3), 26, 0.2)
    dx = ops.conv3x3_first_dgrad(dy.cuda(), wt.cuda(), cin)
    assert dx.dtype == torch.float32 and tuple(dx.shape) == (n, h, w, cin)
    ref, _ = O.conv3x3_bwd(np.zeros((n, h, w, cin)), _np(wt), _np(dy))
    assert rel_err(_np(dx), ref) < 1e-5


@pytest.mark.parametrize('dtype', BOTH)
@pytest.mark.parametrize('n,h,w,cin', [(2, 16, 23, 3), (1, 9, 40, 1), (1, 8, 32, 4)])
def test_first_conv_fprop(dtype, n, h, w, cin):
    from oracle import ops_numpy as O
    from unetk import ops
    x = _rand((n, h, w, cin), 21, 0.5, 0.5)
    wt = _rand((64, cin, 3, 3), 22, 0.3)
    stats = ops.new_stats(64, 'cuda')
    y = ops.conv3x3_first_fprop(x.cuda(), wt.cuda(), 64, dtype, stats=stats)
    xq, wq = x.to(dtype).float(), wt.to(dtype).float()
    ref = O.conv3x3_fwd(_np(xq), _np(wq))
    got = _np(y)
    assert rel_err(got, ref) < TOL[dtype]
    s = stats.sum(0).cpu().numpy()
    assert np.allclose(s[:, 0], got.reshape(-1, 64).sum(0), rtol=1e-5, atol=1e-4)
    assert np.allclose(s[:, 1], (got * got).reshape(-1, 64).sum(0), rtol=1e-5, atol=1e-4)


@pytest.mark.parametrize('dtype', BOTH)
@pytest.mark.parametrize('n,h,w,cin,cout', [(2, 16, 23, 3, 64), (1, 9, 40, 1, 64), (2, 24, 64, 4, 128), (1, 1, 1, 3, 64)])
def test_first_conv_two_pass_training_forward(dtype, n, h, w, cin, cout):
    """Statistics-only pass + unetk_conv3x3_first_fprop_bn_relu_train against the sequence they replace
    (unetk_conv3x3_first_fprop with statistics, unetk_bn_relu_apply): y and the activation bit for bit, the sums to fp64
    rounding (their atomics land in any order)."""
    from unetk import ops
    x = _rand((n, h, w, cin), 21, 0.5, 0.5).cuda()
    wt = _rand((cout, cin, 3, 3), 22, 0.3).cuda()
    bn = _BN(cout)
    s0 = ops.new_stats(cout, 'cuda')
    y0 = ops.conv3x3_first_fprop(x, wt, cout, dtype, stats=s0)
    st = ops.BNState(cout, 'cuda')
    ops.bn_train_finalize(s0, n * h * w, bn, st, 0.1, 1e-5)
    act0 = ops.bn_relu_apply(y0, st)
    s1 = ops.new_stats(cout, 'cuda')
    ops.conv3x3_first_stats(x, wt, cout, dtype, s1)
    assert torch.allclose(s1.sum(0), s0.sum(0), rtol=1e-12, atol=1e-9)
    wide = torch.full((n, h, w, cout + 64), 3.0, dtype=dtype, device='cuda')      # activation into a channel slice
    y1, act1 = ops.conv3x3_first_fprop_bn_relu_train(x, wt, cout, dtype, st, out=wide[..., 64:])
    assert torch.equal(y1, y0) and torch.equal(act1, act0)
    assert bool((wide[..., :64] == 3.0).all())


@pytest.mark.parametrize('n,h,w,cin', [(2, 16, 23, 3), (2, 11, 70, 1)])
@pytest.mark.parametrize('dt16', H16)
def test_first_conv_wgrad_bf16(n, h, w, cin, dt16):
    from oracle import ops_numpy as O
    from unetk import ops
    x = _rand((n, h, w, cin), 23, 0.5, 0.5)
    dy = _rand((n, h, w, 64), 24).to(dt16)
    dw = torch.empty((64, cin, 3, 3), device='cuda')
    ops.conv3x3_first_wgrad(x.cuda(), dy.cuda(), dw)
    _, ref = O.conv3x3_bwd(_np(x.to(dt16)), np.zeros((64, cin, 3, 3)), _np(dy))
    assert rel_err(_np(dw), ref) < 1e-4


@pytest.mark.parametrize('n,h,w,cin,cout', [(2, 16, 23, 3, 64), (2, 11, 70, 1, 64), (1, 8, 32, 4, 128), (3, 9, 33, 3, 64)])
@pytest.mark.parametrize('dt16', H16)
def test_first_conv_wgrad_with_bn_backward(n, h, w, cin, cout, dt16):
    """unetk_conv3x3_first_wgrad_bn (the conv-output gradient formed on the fly, never stored) against the two-kernel
    sequence it replaces -- bit for bit -- and against the oracle's BatchNorm+ReLU backward + conv3x3 weight gradient."""
    from oracle import ops_numpy as O
    from unetk import ops
    x = _rand((n, h, w, cin), 23, 0.5, 0.5).cuda()
    z = _rand((n, h, w, cout), 27, 1.0, 0.3).to(dt16)
    dout = _rand((n, h, w, cout), 28, 0.1).to(dt16)
    bn = _BN(cout)
    z64 = _np(z)
    stats = ops.new_stats(cout, 'cuda')
    stats[0, :, 0] = torch.from_numpy(z64.reshape(-1, cout).sum(0)).cuda()
    stats[0, :, 1] = torch.from_numpy((z64 * z64).reshape(-1, cout).sum(0)).cuda()
    st = ops.BNState(cout, 'cuda')
    ops.bn_train_finalize(stats, n * h * w, bn, st, 0.1, 1e-5)
    # the reduction pass once, for both (its fp64 atomics land in any order: two runs may differ in the last bit)
    from unetk import _native as nat
    sums = ops.new_stats(cout, 'cuda')
    zc, dc_ = z.cuda(), dout.cuda()
    nat.check(nat.fn['unetk_bn_relu_bwd_reduce'](dc_.data_ptr(), cout, zc.data_ptr(), cout, st.scale.data_ptr(),
                                                 st.shift.data_ptr(), st.mean.data_ptr(), st.invstd.data_ptr(),
                                                 sums.data_ptr(), n * h * w, cout, nat.dtype_code(dt16), nat.stream_ptr()),
              'unetk_bn_relu_bwd_reduce')
    # the two-kernel sequence
    dg0, db0 = torch.empty(cout, device='cuda'), torch.empty(cout, device='cuda')
    dy = ops.bn_relu_bwd(dc_, zc, st, bn.weight, dg0, db0, sums=sums.clone())
    dw0 = torch.empty((cout, cin, 3, 3), device='cuda')
    ops.conv3x3_first_wgrad(x, dy, dw0)
    # fused, on channel slices of wider buffers (pitches differ from the channel count)
    wide_d = torch.zeros((n, h, w, cout + 64), dtype=dt16, device='cuda')
    wide_z = torch.zeros((n, h, w, cout + 128), dtype=dt16, device='cuda')
    wide_d[..., 64:] = dout.cuda()
    wide_z[..., 128:] = z.cuda()
    dg1, db1 = torch.empty(cout, device='cuda'), torch.empty(cout, device='cuda')
    dw1 = torch.full((cout, cin, 3, 3), 7.0, device='cuda')
    ops.bn_relu_bwd_first_wgrad(wide_d[..., 64:], wide_z[..., 128:], st, bn.weight, dg1, db1, x, dw1, sums=sums.clone())
    assert torch.equal(dw1, dw0) and torch.equal(dg1, dg0) and torch.equal(db1, db0)
    # ... and with its own reduction pass
    dw2 = torch.empty_like(dw1)
    ops.bn_relu_bwd_first_wgrad(wide_d[..., 64:], wide_z[..., 128:], st, bn.weight, dg1, db1, x, dw2)
    assert rel_err(_np(dw2), _np(dw0)) < 1e-3 and rel_err(_np(dg1), _np(dg0)) < 1e-5
    # oracle
    zn, mean, var, invstd = O.bn_train_fwd(z64, _np(bn.weight), _np(bn.bias))
    dy_ref, _, _ = O.bn_relu_train_bwd(z64, _np(bn.weight), mean, invstd, O.relu(zn), _np(dout))
    _, ref = O.conv3x3_bwd(_np(x.to(dt16)), np.zeros((cout, cin, 3, 3)), dy_ref)
    assert rel_err(_np(dw1), ref) < TOL[dt16]          # the kernel rounds dy to the 16-bit type, as the stored one was
    with pytest.raises(RuntimeError):
        ops.bn_relu_bwd_first_wgrad(dout.float().cuda(), z.float().cuda(), st, bn.weight, dg1, db1, x, dw1)


# --------------------------------------------------------------------------------------- transposed conv
@pytest.mark.parametrize('dtype', BOTH)
@pytest.mark.parametrize('n,h,w,cin,hs,ws', [(2, 8, 11, 128, 17, 23), (1, 4, 5, 128, 11, 13), (1, 9, 33, 256, 18, 66)])
def test_convT_fwd_pad_cat(dtype, n, h, w, cin, hs, ws):
    from oracle import ops_numpy as O
    from unetk import ops
    co = cin // 2
    x = _rand((n, h, w, cin), 31).to(dtype)
    wt = _rand((cin, co, 2, 2), 32, 0.1).to(dtype).float()
    b = _rand((co,), 33, 0.1)
    skip = _rand((n, hs, ws, co), 34).to(dtype)
    cat = torch.full((n, hs, ws, cin), 5.0, dtype=dtype, device='cuda')
    cat[..., :co] = skip.cuda()
    pad_t, pad_l = (hs - 2 * h) // 2, (ws - 2 * w) // 2
    ops.convT2x2_fwd(x.cuda(), ops.pack_weight(2, wt.cuda(), co, cin, dtype), b.cuda(), cat, co, co, pad_t, pad_l)
    ref = O.pad_cat(_np(skip), O.convT2x2_fwd(_np(x), _np(wt), _np(b)))
    assert rel_err(_np(cat), ref) < TOL[dtype]
    assert np.array_equal(_np(cat)[..., :co], _np(skip))          # skip half untouched


@pytest.mark.parametrize('dtype', BOTH)
def test_convT_dgrad(dtype):
    from oracle import ops_numpy as O
    from unetk import ops
    n, h, w, cin, hs, ws = 2, 8, 11, 128, 17, 23
    co = cin // 2
    wt = _rand((cin, co, 2, 2), 35, 0.1).to(dtype).float()
    dcat = _rand((n, hs, ws, cin), 36).to(dtype)
    dx = ops.convT2x2_dgrad(dcat.cuda(), co, ops.pack_weight(3, wt.cuda(), co, cin, dtype), cin, co, h, w, 0, 0)
    _, dup = O.pad_cat_bwd((n, hs, ws, co), (n, 2 * h, 2 * w, co), _np(dcat))
    ref, _, _ = O.convT2x2_bwd(np.zeros((n, h, w, cin)), _np(wt), dup)
    assert rel_err(_np(dx), ref) < TOL[dtype]


@pytest.mark.parametrize('n,h,w,cin,co,hs,ws', [
    (2, 8, 11, 128, 64, 17, 23),       # the pipelined kernel (c_in % 128 == 0): one 64 x 128 channel pair, ragged width
    (1, 9, 70, 256, 128, 20, 143),     # odd low-res height, three column tiles, 2 x 2 channel pairs, F.pad offsets (1, 1)
    (3, 5, 33, 128, 128, 10, 66),      # more tiles than a look-ahead of three, c_out = c_in
    (2, 8, 11, 192, 64, 17, 23),       # c_in % 128 != 0: the first-generation kernel
])
@pytest.mark.parametrize('dt16', H16)
def test_convT_wgrad_bf16(n, h, w, cin, co, hs, ws, dt16):
    from oracle import ops_numpy as O
    from unetk import ops
    x = _rand((n, h, w, cin), 37).to(dt16)
    cs = 64                                                    # skip channels in front of the up-sampled half
    dcat = _rand((n, hs, ws, cs + co), 38).to(dt16)
    dw = torch.empty((cin, co, 2, 2), device='cuda')
    db = torch.empty((co,), device='cuda')
    ops.convT2x2_wgrad(x.cuda(), dcat.cuda(), cs, dw, db, (hs - 2 * h) // 2, (ws - 2 * w) // 2)
    _, dup = O.pad_cat_bwd((n, hs, ws, cs), (n, 2 * h, 2 * w, co), _np(dcat))
    _, rw, rb = O.convT2x2_bwd(_np(x), np.zeros((cin, co, 2, 2)), dup)
    assert rel_err(_np(dw), rw) < 1e-4
    assert rel_err(_np(db), rb) < 1e-4


# --------------------------------------------------------------------------------------- BN + ReLU
class _BN:
    def __init__(self, c):
        self.weight = _rand((c,), 41, 0.25, 1.0).cuda()
        self.bias = _rand((c,), 42, 0.1).cuda()
        self.running_mean = _rand((c,), 43, 0.1).cuda()
        self.running_var = _rand((c,), 44, 0.2, 1.0).cuda()
        self.num_batches_tracked = torch.zeros((), dtype=torch.int64, device='cuda')


@pytest.mark.parametrize('dtype', BOTH)
@pytest.mark.parametrize('c', [64, 256, 1024])
def test_bn_relu_train_fwd_bwd(dtype, c):
    from oracle import ops_numpy as O
    from unetk import ops
    n, h, w = 2, 7, 9
    y = _rand((n, h, w, c), 45, 1.0, 0.3).to(dtype)
    dout = _rand((n, h, w, c), 46, 0.1).to(dtype)
    bn = _BN(c)
    rm0, rv0 = _np(bn.running_mean), _np(bn.running_var)
    y64 = _np(y)
    # statistics as the conv epilogue would deliver them
    stats = ops.new_stats(c, 'cuda')
    stats[0, :, 0] = torch.from_numpy(y64.reshape(-1, c).sum(0)).cuda()
    stats[0, :, 1] = torch.from_numpy((y64 * y64).reshape(-1, c).sum(0)).cuda()
    st = ops.BNState(c, 'cuda')
    ops.bn_train_finalize(stats, n * h * w, bn, st, 0.1, 1e-5)
    act = ops.bn_relu_apply(y.cuda(), st)
    z, mean, var, invstd = O.bn_train_fwd(y64, _np(bn.weight), _np(bn.bias))
    ref_act = O.relu(z)
    assert rel_err(_np(act), ref_act) < TOL[dtype]
    rm, rv = O.bn_running_update(rm0, rv0, mean, var, n * h * w)
    assert rel_err(_np(bn.running_mean), rm) < 1e-5 and rel_err(_np(bn.running_var), rv) < 1e-5
    assert int(bn.num_batches_tracked) == 1
    dgamma = torch.empty(c, device='cuda')
    dbeta = torch.empty(c, device='cuda')
    dy = ops.bn_relu_bwd(dout.cuda(), y.cuda(), st, bn.weight, dgamma, dbeta)
    # the oracle takes the ReLU mask from the fp64 activation; identical except where bn(y) is within rounding of 0
    rdy, rdg, rdb = O.bn_relu_train_bwd(y64, _np(bn.weight), mean, invstd, ref_act, _np(dout))
    assert rel_err(_np(dgamma), rdg) < 1e-3 and rel_err(_np(dbeta), rdb) < 1e-3
    assert rel_err(_np(dy), rdy) < tol(dtype, 2e-2, 1e-3)


@pytest.mark.parametrize('dtype', BOTH)
@pytest.mark.parametrize('pooled', [False, True])
def test_bn_relu_bwd_sums_with_offset_channels(dtype, pooled):
    """The reductions accumulate their second sum as sum dz*y in 16-bit tensors and subtract mean * sum dz once per
    thread (elementwise.hip: s2_step / s2_final).  Channels whose mean is many standard deviations from zero are where
    that cancels: dgamma / dbeta must still match the fp64 oracle on the same stored values."""
    from oracle import ops_numpy as O
    from unetk import ops
    n, h, w, c = 2, 40, 50, 64
    y = _rand((n, h, w, c), 145, 0.25, 6.0).to(dtype)             # mean 6, half-width 0.25: |mean| / std about 40
    y[..., ::2] -= 12.0                                           # every other channel sits at -6
    dout = _rand((n, h, w, c), 146, 0.1).to(dtype)
    bn = _BN(c)
    y64 = _np(y)
    stats = ops.new_stats(c, 'cuda')
    stats[0, :, 0] = torch.from_numpy(y64.reshape(-1, c).sum(0)).cuda()
    stats[0, :, 1] = torch.from_numpy((y64 * y64).reshape(-1, c).sum(0)).cuda()
    st = ops.BNState(c, 'cuda')
    ops.bn_train_finalize(stats, n * h * w, bn, st, 0.1, 1e-5)
    z, mean, var, invstd = O.bn_train_fwd(y64, _np(bn.weight), _np(bn.bias))
    dgamma, dbeta = torch.empty(c, device='cuda'), torch.empty(c, device='cuda')
    if pooled:
        dpool = torch.zeros((n, h // 2, w // 2, c), dtype=dtype, device='cuda')     # no pooled gradient: same sums
        ops.bn_relu_bwd(dout.cuda(), y.cuda(), st, bn.weight, dgamma, dbeta, dpool=dpool)
    else:
        ops.bn_relu_bwd(dout.cuda(), y.cuda(), st, bn.weight, dgamma, dbeta)
    _, rdg, rdb = O.bn_relu_train_bwd(y64, _np(bn.weight), mean, invstd, O.relu(z), _np(dout))
    assert rel_err(_np(dgamma), rdg) < 1e-3 and rel_err(_np(dbeta), rdb) < 1e-3


@pytest.mark.parametrize('dtype', BOTH)
def test_bn_eval(dtype):
    from oracle import ops_numpy as O
    from unetk import ops
    c = 128
    y = _rand((1, 5, 6, c), 47).to(dtype)
    bn = _BN(c)
    st = ops.BNState(c, 'cuda')
    ops.bn_eval_params(bn, st, 1e-5)
    act = ops.bn_relu_apply(y.cuda(), st)
    ref = O.relu(O.bn_eval_fwd(_np(y), _np(bn.weight), _np(bn.bias), _np(bn.running_mean), _np(bn.running_var)))
    assert rel_err(_np(act), ref) < TOL[dtype]


# --------------------------------------------------------------------------------------- pool / upsample
@pytest.mark.parametrize('dtype', BOTH)
@pytest.mark.parametrize('h,w', [(16, 22), (17, 23)])
def test_maxpool_fwd_bwd(dtype, h, w):
    from oracle import ops_numpy as O
    from unetk import ops
    x = _rand((2, h, w, 64), 51).to(dtype)
    p = ops.maxpool_fwd(x.cuda())
    assert np.array_equal(_np(p), O.maxpool2_fwd(_np(x)))          # bit exact
    dp = _rand((2, h // 2, w // 2, 64), 52).to(dtype)
    dskip = _rand((2, h, w, 64), 53).to(dtype)
    dx = ops.maxpool_bwd(x.cuda(), dp.cuda(), dskip.cuda())
    ref = O.maxpool2_bwd(_np(x), _np(dp)) + _np(dskip)
    assert rel_err(_np(dx), ref) < tol(dtype, 1e-2, 1e-6)   # one bf16 rounding of the sum
    dx0 = ops.maxpool_bwd(x.cuda(), dp.cuda())
    assert np.array_equal(_np(dx0), O.maxpool2_bwd(_np(x), _np(dp)))


def test_maxpool_ties_route_to_first_max():
    from unetk import ops
    x = torch.zeros((1, 2, 2, 64), dtype=torch.float32, device='cuda')      # all four equal
    dx = ops.maxpool_bwd(x, torch.ones((1, 1, 1, 64), device='cuda'))
    assert torch.all(dx[0, 0, 0] == 1) and dx.sum() == 64


@pytest.mark.parametrize('dtype', BOTH)
@pytest.mark.parametrize('h,w,hs,ws', [(8, 11, 17, 23), (4, 5, 11, 13), (1, 3, 2, 6)])
def test_bilinear_fwd_bwd(dtype, h, w, hs, ws):
    from oracle import ops_numpy as O
    from unetk import ops
    n, c = 2, 64
    x = _rand((n, h, w, c), 61).to(dtype)
    skip = _rand((n, hs, ws, c), 62).to(dtype)
    cat = torch.full((n, hs, ws, 2 * c), 3.0, dtype=dtype, device='cuda')
    cat[..., :c] = skip.cuda()
    pad_t, pad_l = (hs - 2 * h) // 2, (ws - 2 * w) // 2
    ops.bilinear_fwd(x.cuda(), cat, c, pad_t, pad_l)
    ref = O.pad_cat(_np(skip), O.bilinear2x_ac_fwd(_np(x)))
    assert rel_err(_np(cat), ref) < tol(dtype, 1e-2, 1e-5)
    dcat = _rand((n, hs, ws, 2 * c), 63).to(dtype)
    dx = ops.bilinear_bwd(dcat.cuda(), c, c, h, w, pad_t, pad_l)
    _, dup = O.pad_cat_bwd((n, hs, ws, c), (n, 2 * h, 2 * w, c), _np(dcat))
    assert rel_err(_np(dx), O.bilinear2x_ac_bwd((n, h, w, c), dup)) < tol(dtype, 1e-2, 1e-5)


@pytest.mark.parametrize('dtype', BOTH)
def test_copy_channels(dtype):
    from unetk import ops
    src = _rand((2, 5, 7, 128), 64).to(dtype).cuda()
    dst = torch.zeros((2, 5, 7, 192), dtype=dtype, device='cuda')
    ops.copy_channels(src[..., 64:], dst[..., :64])
    assert torch.equal(dst[..., :64], src[..., 64:]) and torch.all(dst[..., 64:] == 0)


# --------------------------------------------------------------------------------------- head
@pytest.mark.parametrize('dtype', BOTH)
@pytest.mark.parametrize('ncls', [1, 2, 5, 21, 32, 33])
def test_conv1x1_fwd_bwd(dtype, ncls):
    from oracle import ops_numpy as O
    from unetk import ops
    n, h, w, c = 2, 9, 13, 64
    x = _rand((n, h, w, c), 71).to(dtype)
    wt = _rand((ncls, c, 1, 1), 72, 0.2)
    b = _rand((ncls,), 73, 0.1)
    y = ops.conv1x1_fwd(x.cuda(), wt.cuda(), b.cuda(), ncls)
    assert rel_err(_np(y), O.conv1x1_fwd(_np(x), _np(wt), _np(b))) < TOL[dtype]
    dl = _rand((n, h, w, ncls), 74, 0.1).to(dtype)
    dw = torch.empty((ncls, c, 1, 1), device='cuda')
    db = torch.empty((ncls,), device='cuda')
    dx = ops.conv1x1_bwd(x.cuda(), dl.cuda(), wt.cuda(), dw, db)
    rdx, rdw, rdb = O.conv1x1_bwd(_np(x), _np(wt), _np(dl))
    assert rel_err(_np(dx), rdx) < TOL[dtype]
    assert rel_err(_np(dw), rdw) < 1e-4 and rel_err(_np(db), rdb) < 1e-4


# --------------------------------------------------------------------------------------- loss + dice
@pytest.mark.parametrize('dtype', BOTH)
@pytest.mark.parametrize('ncls', [1, 2, 3, 8, 21])
@pytest.mark.parametrize('empty', [False, True])
def test_seg_loss_fwd_bwd(dtype, ncls, empty):
    from oracle import ops_numpy as O
    from unetk import ops
    n, h, w = 2, 12, 19
    z = _rand((n, h, w, ncls), 81, 2.0).to(dtype)
    g = torch.Generator().manual_seed(82)
    mask = torch.zeros((n, h, w), dtype=torch.int64) if empty else torch.randint(0, max(ncls, 2), (n, h, w), generator=g)
    out4 = ops.seg_loss_fwd(z.cuda(), mask.cuda())
    dl = ops.seg_loss_bwd(z.cuda(), mask.cuda(), out4)
    rloss, rdz = O.seg_loss_fwd_bwd(_np(z), mask.numpy())
    assert abs(float(out4[0]) - rloss) < 1e-5 * max(1.0, abs(rloss))
    assert rel_err(_np(dl), rdz) < tol(dtype, 2e-2, 1e-4)
    go = torch.tensor(0.5, device='cuda')
    dl2 = ops.seg_loss_bwd(z.cuda(), mask.cuda(), out4, go)
    assert rel_err(_np(dl2), 0.5 * rdz) < tol(dtype, 2e-2, 1e-4)


@pytest.mark.parametrize('dtype', BOTH)
@pytest.mark.parametrize('n,h,w', [(1, 5, 7), (3, 33, 41), (1, 1, 2)])
def test_seg_loss_two_classes_vector_path_with_tail(dtype, n, h, w):
    """The packed two-class forward reads four pixels per thread; pixel counts that are not multiples of four (and
    out-of-range labels, which count for no class) must match the oracle like any other."""
    from oracle import ops_numpy as O
    from unetk import ops
    z = _rand((n, h, w, 2), 83, 3.0).to(dtype)
    g = torch.Generator().manual_seed(84)
    mask = torch.randint(0, 2, (n, h, w), generator=g)
    out4 = ops.seg_loss_fwd(z.cuda(), mask.cuda())
    rloss, _ = O.seg_loss_fwd_bwd(_np(z), mask.numpy())
    assert abs(float(out4[0]) - rloss) < 1e-5 * max(1.0, abs(rloss))
    # the same logits as a channel slice of a wider buffer (pitch 4) take the generic kernel: same loss
    wide = torch.zeros((n, h, w, 4), dtype=dtype)
    wide[..., :2] = z
    out4w = ops.seg_loss_fwd(wide.cuda()[..., :2], mask.cuda())
    assert abs(float(out4w[0]) - float(out4[0])) < 1e-6 * max(1.0, abs(rloss))


def test_loss_matches_reference_golden():
    """The reference-generated fixture itself (tests/golden/loss_*.npz), not just the restatement."""
    from conftest import golden
    from oracle import weights as W
    from unetk import ops
    for ncls in (1, 2, 21):
        g = golden(f'loss_{ncls}')
        z = torch.from_numpy(W.wave(f'loss{ncls}.z', (2, ncls, 12, 19), 2.0)).permute(0, 2, 3, 1).contiguous().cuda()
        mask = torch.from_numpy(g['mask']).cuda()
        out4 = ops.seg_loss_fwd(z, mask)
        dl = ops.seg_loss_bwd(z, mask, out4)
        assert abs(float(out4[0]) - float(g['loss'])) < 1e-5
        assert rel_err(_np(dl.permute(0, 3, 1, 2)), g['dlogits']) < 1e-4


def test_dice_kernels():
    from conftest import golden
    from unetk import ops
    g = golden('dice')
    a = torch.from_numpy(g['a']).cuda().contiguous()
    b = torch.from_numpy(g['b']).cuda().contiguous()
    # multiclass, reduce_batch_first=True: one group of everything
    d, coef = ops.dice_fwd(a, b, 1, a.numel(), 1e-6, True)
    assert abs(float(d) - float(g['mc_rbf1'])) < 1e-6
    dx = ops.dice_bwd(b, coef, None, -1.0, 1, a.numel()).view_as(a)       # d(1 - dice)
    assert rel_err(_np(dx), g['dloss_mc_da']) < 1e-5
    # reduce_batch_first=False on [3,9,7]: three groups
    a3, b3 = a[:, 0].contiguous(), b[:, 0].contiguous()
    d3, coef3 = ops.dice_fwd(a3, b3, 3, 63, 1e-6, True)
    assert abs(float(d3) - float(g['dice_3d_rbf0'])) < 1e-6
    dx3 = ops.dice_bwd(b3, coef3, None, 1.0, 3, 63).view_as(a3)
    assert rel_err(_np(dx3), g['ddice_rbf0_da']) < 1e-5
    z = torch.zeros((2, 5, 6), device='cuda')
    dz, _ = ops.dice_fwd(z, z, 1, 60, 1e-6, False)
    assert float(dz) == 1.0


# ------------------------------------------------------------------------------------- optimizer step (8f row 2)
@pytest.mark.parametrize('momentum,weight_decay,max_norm', [(0.999, 1e-8, 1.0), (0.0, 0.0, 1.0), (0.9, 1e-2, None)])
def test_clip_rmsprop_matches_torch(momentum, weight_decay, max_norm):
    """unetk_clip_rmsprop_step against torch.nn.utils.clip_grad_norm_ + torch.optim.RMSprop (what train.py:77-78,
    115-116 run), three steps, on a flat buffer with an odd length (vector body + scalar tail)."""
    import ctypes
    from unetk._native import fn, check, stream_ptr
    n = 4 * 50_000 + 3
    g = torch.Generator().manual_seed(21)
    p0 = torch.randn(n, generator=g)
    ref_p = torch.nn.Parameter(p0.clone())
    opt = torch.optim.RMSprop([ref_p], lr=1e-3, weight_decay=weight_decay, momentum=momentum)
    p = p0.clone().cuda()
    sq = torch.zeros(n, device='cuda')
    buf = torch.zeros(n, device='cuda')
    ws = torch.empty(int(fn['unetk_clip_rmsprop_workspace_bytes']()), dtype=torch.uint8, device='cuda')
    norm = torch.zeros((), dtype=torch.float64, device='cuda')
    for step in range(3):
        grad = torch.randn(n, generator=g) * (3.0 if step == 0 else 1e-3)     # first step clips, later ones do not
        ref_p.grad = grad.clone()
        ref_norm = torch.nn.utils.clip_grad_norm_([ref_p], max_norm) if max_norm is not None else grad.norm()
        opt.step()
        gd = grad.cuda()
        check(fn['unetk_clip_rmsprop_step'](p.data_ptr(), gd.data_ptr(), sq.data_ptr(), buf.data_ptr(), n, ws.data_ptr(),
                                            ws.numel(), norm.data_ptr(), float(max_norm or 0.0), 1e-3, 0.99, 1e-8,
                                            weight_decay, momentum, stream_ptr()), 'unetk_clip_rmsprop_step')
        assert abs(float(norm) - float(ref_norm)) <= 1e-5 * float(ref_norm)
        assert rel_err(gd.cpu().numpy(), ref_p.grad.numpy()) < 1e-5            # clipped gradient written back
        assert rel_err(p.cpu().numpy(), ref_p.detach().numpy()) < 1e-5


@pytest.mark.parametrize('dtype', BOTH)
@pytest.mark.parametrize('n,h,w,c', [(2, 9, 13, 64), (1, 16, 22, 128), (1, 7, 2, 64)])
def test_bn_relu_apply_pool_equals_apply_then_pool(dtype, n, h, w, c):
    """The fused BN+ReLU+MaxPool2d(2) pass is bit-identical to the two separate passes, writes into a channel slice,
    and leaves the dropped odd row/column un-pooled."""
    from unetk import ops
    y = _rand((n, h, w, c), 31).to(dtype).cuda()
    st = ops.BNState(c, 'cuda')
    st.scale.copy_(_rand((c,), 32) + 0.2)
    st.shift.copy_(_rand((c,), 33) * 0.3)
    ref = ops.bn_relu_apply(y, st)
    ref_pool = ops.maxpool_fwd(ref)
    cat = torch.full((n, h, w, 2 * c), 7.0, dtype=dtype, device='cuda')
    pool = torch.empty((n, h // 2, w // 2, c), dtype=dtype, device='cuda')
    out = ops.bn_relu_apply(y, st, out=cat[..., :c], pool_out=pool)
    assert torch.equal(out, ref) and torch.equal(pool, ref_pool)
    assert torch.equal(cat[..., c:], torch.full_like(cat[..., c:], 7.0))


@pytest.mark.parametrize('dt16', H16)
@pytest.mark.parametrize('cin,cout', [(64, 64), (128, 64), (64, 128), (256, 256)])
def test_conv3x3_fprop_bn_relu_equals_conv_then_eval_bn(cin, cout, dt16):
    """Inference epilogue (eval-mode BatchNorm2d + ReLU folded into the conv, unet_parts.py:16-18 under net.eval()):
    against torch's fp32 conv -> batch_norm(running stats) -> relu on the same bf16-rounded operands (bar: one bf16
    rounding of the result, 2^-8 relative + a small absolute term for cancellation), and against the two-pass path
    (which additionally rounds the pre-BN tensor to bf16)."""
    import torch.nn.functional as F
    from unetk import ops
    torch.manual_seed(cin * 1000 + cout)
    n, h, w = 2, 21, 37                                  # ragged against the 16x32 tiles
    x = torch.randn((n, h, w, cin), device='cuda').to(dt16)
    wt = (torch.randn((cout, cin, 3, 3), device='cuda') * (2.0 / (9 * cin)) ** 0.5)
    bn = torch.nn.BatchNorm2d(cout).cuda().eval()
    with torch.no_grad():
        bn.weight.uniform_(0.5, 1.5)
        bn.bias.uniform_(-0.5, 0.5)
        bn.running_mean.uniform_(-0.3, 0.3)
        bn.running_var.uniform_(0.5, 2.0)
    wpk = ops.pack_weight(0, wt, cout, cin, dt16)
    st = ops.BNState(cout, 'cuda')
    ops.bn_eval_params(bn, st, bn.eps)
    cat = torch.full((n, h, w, cout + 64), 7.0, dtype=dt16, device='cuda')   # output = channel slice
    pooled = torch.full((n, h // 2, w // 2, cout), -1.0, dtype=dt16, device='cuda')
    got = ops.conv3x3_fprop_bn_relu(x, wpk, cout, st, out=cat[..., :cout], pool_out=pooled)
    # the MaxPool2d(2) of the stored activation comes out of the same epilogue (floor mode: odd last row/column dropped)
    assert torch.equal(pooled, ops.maxpool_fwd(got.contiguous()))
    assert torch.equal(cat[..., cout:].float(), torch.full((n, h, w, 64), 7.0, device='cuda')), 'wrote past its slice'
    with torch.no_grad():
        y = F.conv2d(x.float().permute(0, 3, 1, 2), wt.to(dt16).float(), padding=1)
        ref = F.relu(bn(y)).permute(0, 2, 3, 1)
    ulp = 2.0 ** -8 if dt16 == torch.bfloat16 else 2.0 ** -11       # one rounding of the result
    err = (got.float() - ref).abs()
    assert float((err - ulp * ref.abs()).max()) < 2e-3
    two = ops.bn_relu_apply(ops.conv3x3_fprop(x, wpk, cout), st)
    assert float(((two.float() - got.float()).abs() - 4 * ulp * got.float().abs()).max()) < 2e-2 * ulp * 256
    assert float((two.float() - got.float()).abs().mean()) < 2e-3 * ulp * 256


@pytest.mark.parametrize('dtype', BOTH)
def test_first_conv_bn_relu_equals_torch_eval_chain(dtype):
    import torch.nn.functional as F
    from unetk import ops
    torch.manual_seed(5)
    n, h, w, cout = 2, 19, 45, 64
    x = torch.rand((n, h, w, 3), device='cuda')
    wt = torch.randn((cout, 3, 3, 3), device='cuda') * 0.3
    bn = torch.nn.BatchNorm2d(cout).cuda().eval()
    with torch.no_grad():
        bn.weight.uniform_(0.5, 1.5)
        bn.bias.uniform_(-0.5, 0.5)
        bn.running_mean.uniform_(-0.3, 0.3)
        bn.running_var.uniform_(0.5, 2.0)
    st = ops.BNState(cout, 'cuda')
    ops.bn_eval_params(bn, st, bn.eps)
    got = ops.conv3x3_first_fprop_bn_relu(x, wt, cout, dtype, st)
    xr, wr = x.to(dtype).float(), wt.to(dtype).float()
    with torch.no_grad():
        ref = F.relu(bn(F.conv2d(xr.permute(0, 3, 1, 2), wr, padding=1))).permute(0, 2, 3, 1)
    err = (got.float() - ref).abs()
    if dtype != torch.float32:
        ulp = 2.0 ** -8 if dtype == torch.bfloat16 else 2.0 ** -11
        assert float((err - ulp * ref.abs()).max()) < 1e-3
    else:
        assert float(err.max()) < 2e-5


@pytest.mark.parametrize('dtype', BOTH)
@pytest.mark.parametrize('h,w,c', [(10, 13, 64), (7, 6, 128), (9, 9, 512)])
def test_bn_relu_bwd_with_pool_scatter_equals_maxpool_bwd_then_bn_bwd(dtype, h, w, c):
    """Encoder blocks (unet_model.py:26-30): the BatchNorm backward that forms dskip + max-pool scatter on the fly
    against the two-kernel path (unetk_maxpool2x2_bwd, then the plain BatchNorm backward).  fp32: the element-wise
    result is the same arithmetic -> equal to 1e-6 (only the channel sums are accumulated in another order); bf16: the
    two-kernel path rounds dskip + dpool to bf16 in between."""
    from unetk import ops
    torch.manual_seed(h * 100 + w + c)
    n = 2
    y = torch.randn((n, h, w, c), device='cuda').to(dtype)
    bn = torch.nn.BatchNorm2d(c).cuda().train()
    with torch.no_grad():
        bn.weight.uniform_(0.5, 1.5)
        bn.bias.uniform_(-0.3, 0.3)
    stats = ops.new_stats(c, 'cuda')
    yf = y.float()
    stats[0, :, 0] = yf.double().sum((0, 1, 2))
    stats[0, :, 1] = (yf.double() ** 2).sum((0, 1, 2))
    st = ops.BNState(c, 'cuda')

    class B:
        weight, bias, running_mean, running_var, num_batches_tracked = bn.weight, bn.bias, None, None, None
    ops.bn_train_finalize(stats, n * h * w, B, st, 0.1, 1e-5)
    cat = torch.randn((n, h, w, c + 64), device='cuda').to(dtype)        # dskip = channel slice of the concat gradient
    dskip = cat[..., :c]
    dpool = torch.randn((n, h // 2, w // 2, c), device='cuda').to(dtype)
    act = ops.bn_relu_apply(y, st)
    dg1, db1, dg2, db2 = (torch.empty(c, device='cuda') for _ in range(4))
    fused = ops.bn_relu_bwd(dskip, y, st, bn.weight, dg1, db1, dpool=dpool)
    dout = ops.maxpool_bwd(act, dpool, dskip)
    plain = ops.bn_relu_bwd(dout, y, st, bn.weight, dg2, db2)
    tol = 1e-5 if dtype == torch.float32 else 3e-2
    scale = float(plain.float().abs().max())
    assert float((fused.float() - plain.float()).abs().max()) <= tol * scale
    assert float((dg1 - dg2).abs().max()) <= tol * float(dg2.abs().max())
    assert float((db1 - db2).abs().max()) <= tol * float(db2.abs().max())
    assert torch.equal(cat[..., c:], cat[..., c:]) and torch.isfinite(fused.float()).all()


@pytest.mark.parametrize('dtype', BOTH)
@pytest.mark.parametrize('ncls', [1, 2, 4])
def test_bn_relu_bwd_with_head_gradient_equals_conv1x1_dgrad_then_bn_bwd(dtype, ncls):
    """Last block (unet_model.py:35): the BatchNorm backward that forms OutConv's input gradient on the fly against
    unetk_conv1x1_dgrad followed by the plain BatchNorm backward.  fp32: same fp32 chain -> 1e-5; bf16: the two-kernel
    path rounds the input gradient to bf16 in between."""
    from unetk import ops
    torch.manual_seed(ncls)
    n, h, w, c = 2, 11, 17, 64
    y = torch.randn((n, h, w, c), device='cuda').to(dtype)
    gamma = torch.rand(c, device='cuda') + 0.5
    stats = ops.new_stats(c, 'cuda')
    stats[0, :, 0] = y.double().sum((0, 1, 2))
    stats[0, :, 1] = (y.double() ** 2).sum((0, 1, 2))
    st = ops.BNState(c, 'cuda')

    class B:
        weight, bias, running_mean, running_var, num_batches_tracked = gamma, torch.zeros(c, device='cuda'), None, None, None
    ops.bn_train_finalize(stats, n * h * w, B, st, 0.1, 1e-5)
    dl = (torch.randn((n, h, w, ncls), device='cuda') * 0.1).to(dtype)
    wh = torch.randn((ncls, c), device='cuda') * 0.2
    x = torch.randn((n, h, w, c), device='cuda').to(dtype)
    dg1, db1, dg2, db2 = (torch.empty(c, device='cuda') for _ in range(4))
    # OutConv's own gradients come out of the same pass: its input is this layer's activation
    x = ops.bn_relu_apply(y, st)
    hw1, hb1 = torch.empty((ncls, c, 1, 1), device='cuda'), torch.empty(ncls, device='cuda')
    hw2, hb2 = torch.empty((ncls, c, 1, 1), device='cuda'), torch.empty(ncls, device='cuda')
    fused = ops.bn_relu_bwd(None, y, st, gamma, dg1, db1, dhead=(dl, wh, hw1, hb1))
    dx = ops.conv1x1_bwd(x, dl, wh, hw2, hb2, True)
    plain = ops.bn_relu_bwd(dx, y, st, gamma, dg2, db2)
    assert float((hw1 - hw2).abs().max()) <= 1e-4 * float(hw2.abs().max()) + 1e-6
    assert float((hb1 - hb2).abs().max()) <= 1e-4 * float(hb2.abs().max()) + 1e-6
    tol = 1e-5 if dtype == torch.float32 else 3e-2
    assert float((fused.float() - plain.float()).abs().max()) <= tol * float(plain.float().abs().max())
    assert float((dg1 - dg2).abs().max()) <= tol * float(dg2.abs().max())
    assert float((db1 - db2).abs().max()) <= tol * float(db2.abs().max())


@pytest.mark.parametrize('dtype', BOTH)
@pytest.mark.parametrize('ncls,npix_shape', [(1, (2, 9, 13)), (2, (1, 16, 16)), (2, (2, 7, 11)), (3, (1, 5, 9)), (4, (2, 6, 7)),
                                             (2, (2, 96, 131))])     # many pixels: the DPP pixel sum of the 16-bit form
def test_bn_relu_apply_head_equals_apply_then_conv1x1(dtype, ncls, npix_shape):
    """Last block forward (unet_parts.py:20-21 then :74): the fused kernel's activation must equal bn_relu_apply's bit
    for bit, its logits conv1x1_fwd's on that activation (same per-lane fma chain and shuffle order: bit for bit)."""
    from unetk import ops
    torch.manual_seed(ncls * 7 + npix_shape[2])
    n, h, w = npix_shape
    c = 64
    y = torch.randn((n, h, w, c), device='cuda').to(dtype)
    st = ops.BNState(c, 'cuda')
    st.buf.uniform_(-1.0, 1.0)
    wh = torch.randn((ncls, c), device='cuda') * 0.2
    bias = torch.randn(ncls, device='cuda')
    cat = torch.full((n, h, w, c + 64), 5.0, device='cuda').to(dtype)
    act, logits = ops.bn_relu_apply_head(y, st, wh, bias, out=cat[..., :c])
    ref_act = ops.bn_relu_apply(y, st)
    ref_logits = ops.conv1x1_fwd(ref_act, wh.reshape(ncls, c, 1, 1), bias, ncls)
    assert torch.equal(act, ref_act)
    assert torch.equal(cat[..., c:].float(), torch.full((n, h, w, 64), 5.0, device='cuda'))
    assert logits.shape == ref_logits.shape and torch.equal(logits, ref_logits)
    # logits only (what the whole-net forward asks for: nothing reads the activation): the same logits, nothing stored
    none, logits2 = ops.bn_relu_apply_head(y, st, wh, bias, store_act=False)
    assert none is None and torch.equal(logits2, ref_logits)


def test_scratch_buffers_are_per_stream():
    """The zeroed statistics arena and the grow-only workspace are reused in stream order, so every stream has its own."""
    from unetk import ops
    dev = torch.device('cuda', torch.cuda.current_device())
    ops.begin_stats_arena(dev, 64)
    a = ops.new_stats(64, dev)
    wa = ops.workspace(4096, dev)
    side = torch.cuda.Stream()
    with torch.cuda.stream(side):
        ops.begin_stats_arena(dev, 64)
        b = ops.new_stats(64, dev)
        wb = ops.workspace(4096, dev)
    assert a.data_ptr() != b.data_ptr() and wa.data_ptr() != wb.data_ptr()
    ops.begin_stats_arena(dev, 64)
    assert ops.new_stats(64, dev).data_ptr() == a.data_ptr()         # same stream: the same slice again
    assert ops.workspace(4096, dev).data_ptr() == wa.data_ptr()
    torch.cuda.synchronize()
