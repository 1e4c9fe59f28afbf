"""dgrad alone, the stand-alone BN-backward reduce, and the fused dgrad+reduce at the configs[1] DoubleConv shapes."""
import os
import sys

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', 'pytorch-unet_amd'))
from unetk import ops  # noqa: E402
from unetk._native import fn, dtype_code, stream_ptr  # noqa: E402

SHAPES = [(16, 640, 959, 64), (16, 320, 479, 128), (16, 160, 239, 256), (16, 80, 119, 512), (16, 40, 59, 1024)]
only = os.environ.get('ONLY')
if only:
    SHAPES = [SHAPES[int(i)] for i in only.split(',')]


class St:
    pass


def timeit(f, reps=10):
    f()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(reps):
        f()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / reps


for n, h, w, c in SHAPES:
    dt = torch.bfloat16
    dy = (torch.randn(n, h, w, c, device='cuda') * 0.1).to(dt)
    z = torch.randn(n, h, w, c, device='cuda').to(dt)
    wq = (torch.randn(c, c, 3, 3, device='cuda') * 0.05)
    wpk = ops.pack_weight(1, wq, c, c, dt)
    st = St()
    st.mean = torch.zeros(c, device='cuda'); st.invstd = torch.ones(c, device='cuda')
    st.scale = torch.ones(c, device='cuda'); st.shift = torch.zeros(c, device='cuda')
    out = torch.empty_like(dy)
    sums = torch.zeros(32 * c * 2, dtype=torch.float64, device='cuda')
    t0 = timeit(lambda: ops.conv3x3_dgrad(dy, wpk, c, out=out))
    t1 = timeit(lambda: fn['unetk_bn_relu_bwd_reduce'](out.data_ptr(), c, z.data_ptr(), c, st.scale.data_ptr(),
                                                       st.shift.data_ptr(), st.mean.data_ptr(), st.invstd.data_ptr(),
                                                       sums.data_ptr(), n * h * w, c, dtype_code(dt), stream_ptr()))
    t2 = timeit(lambda: fn['unetk_conv3x3_dgrad_bn_reduce'](dy.data_ptr(), c, 0, wpk.data_ptr(), out.data_ptr(), c, 0,
                                                            z.data_ptr(), c, st.scale.data_ptr(), st.shift.data_ptr(),
                                                            st.mean.data_ptr(), st.invstd.data_ptr(), sums.data_ptr(),
                                                            n, h, w, c, c, dtype_code(dt), stream_ptr()))
    print(f'{h}x{w} {c}->{c}: dgrad {t0:.3f} ms  reduce {t1:.3f} ms  fused {t2:.3f} ms  (saves {t0 + t1 - t2:+.3f})', flush=True)
