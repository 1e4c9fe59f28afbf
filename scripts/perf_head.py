"""Timing of the three OutConv-fused BatchNorm kernels at the benchmark's last layer (16 x 640 x 959 x 64, 2 classes)
against their HBM floors:  python scripts/perf_head.py"""
import os, sys, time
import torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'pytorch-unet_amd'))
from unetk._native import fn, check, dtype_code

dt = {'bf16': torch.bfloat16, 'fp16': torch.float16, 'fp32': torch.float32}[os.environ.get('DT', 'bf16')]
N, H, W, C, K = int(os.environ.get('N', 16)), 640, 959, 64, 2
npix = N * H * W
dev = 'cuda'
y = (torch.randn((npix, C), device=dev)).to(dt)
dl = (torch.randn((npix, K), device=dev) * 0.01).to(dt)
logits = torch.empty((npix, K), device=dev, dtype=dt)
dy = torch.empty_like(y)
scale = torch.rand(C, device=dev) + 0.5; shift = torch.randn(C, device=dev) * 0.1
mean = torch.randn(C, device=dev) * 0.1; invstd = torch.rand(C, device=dev) + 0.5
coef = torch.rand(3 * C, device=dev)
w = torch.randn((K, C), device=dev) * 0.1; bias = torch.zeros(K, device=dev)
sums = torch.zeros(32 * C * 2, device=dev, dtype=torch.float64)
hs = torch.zeros(32 * K * (C + 1), device=dev, dtype=torch.float64)
dc = dtype_code(dt)
st = torch.cuda.current_stream().cuda_stream
es = y.element_size()
calls = {
    'bn_relu_apply_head': (lambda: fn['unetk_bn_relu_apply_head'](y.data_ptr(), C, scale.data_ptr(), shift.data_ptr(), None, 0,
                                                                  w.data_ptr(), bias.data_ptr(), K, logits.data_ptr(), K, npix, C, dc, st),
                           npix * (C + K) * es),
    'bn_relu_bwd_reduce_head': (lambda: fn['unetk_bn_relu_bwd_reduce_head'](dl.data_ptr(), K, w.data_ptr(), K, y.data_ptr(), C,
                                                                            scale.data_ptr(), shift.data_ptr(), mean.data_ptr(),
                                                                            invstd.data_ptr(), sums.data_ptr(), hs.data_ptr(), npix, C, dc, st),
                                npix * (C + K) * es),
    'bn_relu_bwd_apply_head': (lambda: fn['unetk_bn_relu_bwd_apply_head'](dl.data_ptr(), K, w.data_ptr(), K, y.data_ptr(), C,
                                                                          scale.data_ptr(), shift.data_ptr(), mean.data_ptr(),
                                                                          coef.data_ptr(), dy.data_ptr(), C, npix, C, dc, st),
                               npix * (2 * C + K) * es),
}
only = os.environ.get('ONLY')
for name, (f, nbytes) in calls.items():
    if only and only not in name:
        continue
    check(f(), name)
    torch.cuda.synchronize()
    reps = 20
    t0 = time.perf_counter()
    for _ in range(reps):
        f()
    torch.cuda.synchronize()
    t = (time.perf_counter() - t0) / reps
    print(f'{name:26s} {t * 1e3:.3f} ms  {nbytes / t / 1e12:.2f} TB/s', flush=True)
