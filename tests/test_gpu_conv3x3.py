"""GPU parity of the MFMA conv3x3 kernels (fprop, fused BN statistics, dgrad) through the C ABI,
against the fp64 NumPy oracle (oracle/ops_numpy.py) on the same seeded inputs."""
import numpy as np
import pytest
import torch

from conftest import rel_err

pytestmark = pytest.mark.gpu

# fp32: BASELINE.json's 1e-4 relative; bf16: 2e-2 relative (inputs are pre-rounded to bf16, so the
# only error left is accumulation order + one output rounding, far inside that budget).
TOL = {torch.float32: 1e-4, torch.bfloat16: 2e-2}


def _rand(shape, seed, scale=1.0):
    g = torch.Generator().manual_seed(seed)
    return (torch.rand(shape, generator=g) * 2 - 1) * scale


@pytest.mark.parametrize('dtype', [torch.float32, torch.bfloat16])
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


@pytest.mark.parametrize('dtype', [torch.float32, torch.bfloat16])
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


@pytest.mark.parametrize('dtype', [torch.float32, torch.bfloat16])
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
