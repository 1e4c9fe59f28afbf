"""Timing of the fused (skip + max-pool scatter) BatchNorm backward kernels at the encoder shapes of configs[1]."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'pytorch-unet_amd'))
import torch
from unetk import ops

def timeit(f, reps=10):
    f(); torch.cuda.synchronize()
    a, b = torch.cuda.Event(True), torch.cuda.Event(True)
    a.record()
    for _ in range(reps):
        f()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / reps

for (h, w, c) in [(640, 959, 64), (320, 479, 128), (160, 239, 256), (80, 119, 512)]:
    n = 16
    y = torch.randn((n, h, w, c), device='cuda').bfloat16()
    cat = torch.randn((n, h, w, 2 * c), device='cuda').bfloat16()
    dskip = cat[..., :c]
    dpool = torch.randn((n, h // 2, w // 2, c), device='cuda').bfloat16()
    st = ops.BNState(c, 'cuda'); st.buf.uniform_(0.5, 1.5)
    gamma = torch.ones(c, device='cuda'); dg = torch.empty(c, device='cuda'); db = torch.empty(c, device='cuda')
    out = torch.empty_like(y)
    S = y.numel() * 2 / 1e9
    sums = ops.new_stats(c, 'cuda'); coef = torch.rand((3, c), device='cuda')
    fn, sp = ops.fn, ops.stream_ptr
    t_r = timeit(lambda: fn['unetk_bn_relu_bwd_reduce_pool'](dskip.data_ptr(), 2 * c, dpool.data_ptr(), c, y.data_ptr(), c, st.scale.data_ptr(), st.shift.data_ptr(), st.mean.data_ptr(), st.invstd.data_ptr(), sums.data_ptr(), n, h, w, c, 0, sp()))
    t_a = timeit(lambda: fn['unetk_bn_relu_bwd_apply_pool'](dskip.data_ptr(), 2 * c, dpool.data_ptr(), c, y.data_ptr(), c, st.scale.data_ptr(), st.shift.data_ptr(), st.mean.data_ptr(), coef.data_ptr(), out.data_ptr(), c, n, h, w, c, 0, sp()))
    t_r0 = timeit(lambda: fn['unetk_bn_relu_bwd_reduce'](dskip.data_ptr(), 2 * c, y.data_ptr(), c, st.scale.data_ptr(), st.shift.data_ptr(), st.mean.data_ptr(), st.invstd.data_ptr(), sums.data_ptr(), n * h * w, c, 0, sp()))
    t_a0 = timeit(lambda: fn['unetk_bn_relu_bwd_apply'](dskip.data_ptr(), 2 * c, y.data_ptr(), c, st.scale.data_ptr(), st.shift.data_ptr(), st.mean.data_ptr(), coef.data_ptr(), out.data_ptr(), c, n * h * w, c, 0, sp()))
    t_m = timeit(lambda: ops.maxpool_bwd(y, dpool, dskip, out=dskip))
    print(f'{h}x{w}x{c}: reduce_pool {t_r:.3f} ms ({2.25 * S / t_r:.2f} TB/s)  apply_pool {t_a:.3f} ms ({3.25 * S / t_a:.2f} TB/s) | plain reduce {t_r0:.3f} apply {t_a0:.3f} maxpool_bwd {t_m:.3f}  -> fused {t_r + t_a:.3f} vs {t_r0 + t_a0 + t_m:.3f}')
