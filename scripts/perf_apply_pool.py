"""Timing of the forward BatchNorm + ReLU + MaxPool2d(2) kernel at the encoder shapes of configs[1]."""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'pytorch-unet_amd'))
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

N = 16
for h, w, c in [(640, 959, 64), (320, 479, 128), (160, 239, 256), (80, 119, 512)]:
    y = torch.randn((N, h, w, c), device='cuda').to(torch.bfloat16)
    st = ops.BNState(c, 'cuda'); st.buf.uniform_(-1.0, 1.0)
    out = torch.empty_like(y)
    pool = torch.empty((N, h // 2, w // 2, c), device='cuda', dtype=torch.bfloat16)
    t = timeit(lambda: ops.bn_relu_apply(y, st, out=out, pool_out=pool))
    t0 = timeit(lambda: ops.bn_relu_apply(y, st, out=out))
    nb = (2 * y.numel() + pool.numel()) * 2
    print(f'{h}x{w}x{c}: apply_pool {t:.3f} ms ({nb / t / 1e9:.2f} TB/s) | plain apply {t0:.3f} ms ({2 * y.numel() * 2 / t0 / 1e9:.2f} TB/s)', flush=True)
