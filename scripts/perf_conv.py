"""Ad-hoc timing of the conv kernels at the BASELINE config-2 layer shapes (batch 16, 959x640)."""
import os, sys, time
import torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'pytorch-unet_amd'))
from unetk import ops

N = int(os.environ.get('N', 16))
layers = [(640, 959, 64, 64), (640, 959, 128, 64), (320, 479, 64, 128), (320, 479, 128, 128), (320, 479, 256, 128),
          (160, 239, 128, 256), (160, 239, 256, 256), (160, 239, 512, 256), (80, 119, 256, 512), (80, 119, 512, 512),
          (80, 119, 1024, 512), (40, 59, 512, 1024), (40, 59, 1024, 1024),
          (640, 959, 64, 128)]          # the shape of up4's first conv as a dgrad (GEMM-K 64, GEMM-N 128)
if os.environ.get('ONLY'):
    layers = [layers[int(i)] for i in os.environ['ONLY'].split(',')]
dt = {'bf16': torch.bfloat16, 'fp16': torch.float16, 'fp32': torch.float32}[os.environ.get('DT', 'bf16')]
for (h, w, cin, cout) in layers:
    x = (torch.rand((N, h, w, cin), device='cuda') - 0.5).to(dt)
    wt = (torch.rand((cout, cin, 3, 3), device='cuda') - 0.5) * 0.05
    wpk = ops.pack_weight(0, wt, cout, cin, dt)
    stats = ops.new_stats(cout, 'cuda') if os.environ.get('STATS', '1') == '1' else None
    y = ops.conv3x3_fprop(x, wpk, cout, stats=stats)
    torch.cuda.synchronize()
    reps = 20
    t0 = time.perf_counter()
    for _ in range(reps):
        ops.conv3x3_fprop(x, wpk, cout, out=y, stats=stats)
    torch.cuda.synchronize()
    dtm = (time.perf_counter() - t0) / reps
    fl = 2.0 * N * h * w * 9 * cin * cout
    print(f'{h}x{w} {cin}->{cout}: {dtm*1e3:.3f} ms  {fl/dtm/1e12:.1f} TFLOP/s', flush=True)
