"""Does an MFMA-bound kernel run faster between HBM-bound kernels than back to back?  (the launch
order of the backward.)  Times conv3x3 fprop launches alone, BatchNorm+ReLU apply launches alone, and the two
alternating; HIP events around every conv launch give its duration in both settings."""
import os
import sys

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'pytorch-unet_amd'))
from unetk import ops  # noqa: E402
from unetk._native import fn, stream_ptr  # noqa: E402

N = 16
for (h, w, cin, cout) in [(640, 959, 64, 64), (320, 479, 128, 128), (160, 239, 256, 256), (80, 119, 512, 512)]:
    dt = torch.bfloat16
    x = (torch.rand((N, h, w, cin), device='cuda') - 0.5).to(dt)
    wt = (torch.rand((cout, cin, 3, 3), device='cuda') - 0.5) * 0.05
    wpk = ops.pack_weight(0, wt, cout, cin, dt)
    stats = ops.new_stats(cout, 'cuda')
    y = ops.conv3x3_fprop(x, wpk, cout, stats=stats)
    a = torch.empty_like(y)
    sc = torch.rand(cout, device='cuda') + 0.5
    sh = torch.rand(cout, device='cuda') - 0.5
    npix = N * h * w

    def conv():
        ops.conv3x3_fprop(x, wpk, cout, out=y, stats=stats)

    def apply():
        fn['unetk_bn_relu_apply'](y.data_ptr(), cout, sc.data_ptr(), sh.data_ptr(), a.data_ptr(), cout, npix, cout, 0,
                                  stream_ptr())

    def conv_ms(between, reps=30):
        evs = []
        for _ in range(3):
            conv()
            between()
        torch.cuda.synchronize()
        for _ in range(reps):
            s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            s.record()
            conv()
            e.record()
            between()
            evs.append((s, e))
        torch.cuda.synchronize()
        return sum(s.elapsed_time(e) for s, e in evs) / reps

    alone = conv_ms(lambda: None)
    mixed = conv_ms(apply)
    mixed2 = conv_ms(lambda: (apply(), apply()))
    print(f'{h}x{w} {cin}->{cout}: conv back to back {alone:.3f} ms | with one BN apply between {mixed:.3f} ms | with two {mixed2:.3f} ms',
          flush=True)
