"""Timing of the first layer's kernels (3 -> 64 channels at 16 x 640 x 959) against their HBM floors."""
import os
import sys

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', 'pytorch-unet_amd'))
from unetk import ops  # noqa: E402

n, h, w = 16, 640, 959
x = torch.rand(n, h, w, 3, device='cuda')
wt = (torch.rand(64, 3, 3, 3, device='cuda') - 0.5) * 0.3
dy = (torch.rand(n, h, w, 64, device='cuda') - 0.5).bfloat16()
dw = torch.empty_like(wt)


def timeit(f, reps=20):
    for _ in range(3):
        f()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(reps):
        f()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / reps


stats = ops.new_stats(64, 'cuda')
y = ops.conv3x3_first_fprop(x, wt, 64, torch.bfloat16, stats=stats)
t = timeit(lambda: ops.conv3x3_first_fprop(x, wt, 64, torch.bfloat16, stats=stats))
gb = (x.numel() * 4 + y.numel() * 2) / 1e9
print(f'first fprop: {t:.3f} ms  {gb / t:.2f} TB/s of {gb:.2f} GB')
t = timeit(lambda: ops.conv3x3_first_wgrad(x, dy, dw))
print(f'first wgrad: {t:.3f} ms  {gb / t:.2f} TB/s of {gb:.2f} GB')

# training forward: conv (+ statistics) then the apply pass, against statistics-only pass + conv with BN + ReLU
st = ops.BNState(64, 'cuda')
st.scale.fill_(0.7); st.shift.fill_(0.1)
act = torch.empty_like(y)
out = torch.empty_like(y)
t1 = timeit(lambda: ops.conv3x3_first_fprop(x, wt, 64, torch.bfloat16, out=y, stats=stats))
t2 = timeit(lambda: ops.bn_relu_apply(y, st, out=act))
t3 = timeit(lambda: ops.conv3x3_first_stats(x, wt, 64, torch.bfloat16, stats))
t4 = timeit(lambda: ops.conv3x3_first_fprop_bn_relu_train(x, wt, 64, torch.bfloat16, st, out=out))
print(f'forward: conv+stats {t1:.3f} + apply {t2:.3f} = {t1 + t2:.3f} ms | stats only {t3:.3f} + conv+BN+ReLU {t4:.3f} = {t3 + t4:.3f} ms')
