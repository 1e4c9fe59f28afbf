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
