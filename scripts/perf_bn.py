"""Ad-hoc timing of the BatchNorm+ReLU passes on the config-2 activation sizes."""
import os, sys, time
import torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'pytorch-unet_amd'))
from unetk import ops
from unetk._native import fn, check, stream_ptr

def timeit(f, reps=20):
    for _ in range(3): f()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(reps): f()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / reps

tot = [0, 0, 0]
for (h, w, c) in [(640, 959, 64), (320, 479, 128), (160, 239, 256), (80, 119, 512), (40, 59, 1024)]:
    npix = 16 * h * w
    y = (torch.rand((16, h, w, c), device='cuda') - 0.5).bfloat16()
    g = (torch.rand((16, h, w, c), device='cuda') - 0.5).bfloat16()
    out = torch.empty_like(y)
    sc = torch.rand(c, device='cuda') + 0.5; sh = torch.rand(c, device='cuda') - 0.5
    mean = torch.zeros(c, device='cuda'); inv = torch.ones(c, device='cuda'); coef = torch.rand(3 * c, device='cuda')
    sums = torch.zeros(32 * c * 2, device='cuda', dtype=torch.float64)
    ta = timeit(lambda: fn['unetk_bn_relu_apply'](y.data_ptr(), c, sc.data_ptr(), sh.data_ptr(), out.data_ptr(), c, npix, c, 0, stream_ptr()))
    tr = timeit(lambda: fn['unetk_bn_relu_bwd_reduce'](g.data_ptr(), c, y.data_ptr(), c, sc.data_ptr(), sh.data_ptr(), mean.data_ptr(), inv.data_ptr(), sums.data_ptr(), npix, c, 0, stream_ptr()))
    tb = timeit(lambda: fn['unetk_bn_relu_bwd_apply'](g.data_ptr(), c, y.data_ptr(), c, sc.data_ptr(), sh.data_ptr(), mean.data_ptr(), coef.data_ptr(), out.data_ptr(), c, npix, c, 0, stream_ptr()))
    b = y.numel() * 2
    print(f'{h}x{w}x{c}: apply {ta*1e3:.3f} ms {2*b/ta/1e12:.2f} TB/s | reduce {tr*1e3:.3f} ms {2*b/tr/1e12:.2f} TB/s | bwd_apply {tb*1e3:.3f} ms {3*b/tb/1e12:.2f} TB/s', flush=True)
    tot[0] += ta; tot[1] += tr; tot[2] += tb
print('totals ms', [round(t * 1e3, 3) for t in tot])
