"""Times the first layer's BatchNorm-backward apply + weight gradient, two kernels vs the fused one, at the benchmark's
shape (16 x 640 x 959, 3 -> 64).  python scripts/perf_first_bwd.py [bf16|fp16]"""
import os
import sys
import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', 'pytorch-unet_amd'))
from unetk import ops  # noqa: E402

dt = torch.float16 if (len(sys.argv) > 1 and sys.argv[1] == 'fp16') else torch.bfloat16
n, h, w, cin, c = 16, 640, 959, 3, 64
g = torch.Generator(device='cuda').manual_seed(1)
x = torch.rand((n, h, w, cin), device='cuda', generator=g)
z = (torch.randn((n, h, w, c), device='cuda', generator=g)).to(dt)
dout = (torch.randn((n, h, w, c), device='cuda', generator=g) * 0.1).to(dt)


class BN:
    weight = torch.rand(c, device='cuda') + 0.5
    bias = torch.zeros(c, device='cuda')
    running_mean = None
    running_var = None
    num_batches_tracked = None


st = ops.BNState(c, 'cuda')
stats = ops.new_stats(c, 'cuda')
zf = z.float().reshape(-1, c)
stats[0, :, 0] = zf.sum(0).double()
stats[0, :, 1] = (zf * zf).sum(0).double()
ops.bn_train_finalize(stats, n * h * w, BN, st, 0.0, 1e-5)
dg, db = torch.empty(c, device='cuda'), torch.empty(c, device='cuda')
dw0 = torch.empty((c, cin, 3, 3), device='cuda')
dw1 = torch.empty((c, cin, 3, 3), device='cuda')
sums = ops.new_stats(c, 'cuda')
out = torch.empty_like(z)


def two():
    dy = ops.bn_relu_bwd(dout, z, st, BN.weight, dg, db, out=out, sums=sums)
    ops.conv3x3_first_wgrad(x, dy, dw0)


def fused():
    ops.bn_relu_bwd_first_wgrad(dout, z, st, BN.weight, dg, db, x, dw1, sums=sums)


def timeit(f, reps=20):
    for _ in range(3):
        f()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        f()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps


t2, t1 = timeit(two), timeit(fused)
print(f'{dt}: apply + wgrad {t2:.3f} ms, fused {t1:.3f} ms, equal: {torch.equal(dw0, dw1)}', flush=True)
