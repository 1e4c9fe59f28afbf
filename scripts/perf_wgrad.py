"""Ad-hoc timing of conv3x3 wgrad at the BASELINE config-2 layer shapes (batch 16, 959x640)."""
import os, sys, time
import torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'pytorch-unet_amd'))
from unetk import ops

N = int(os.environ.get('N', 16))
layers = [(640, 959, 64, 64), (640, 959, 128, 64), (320, 479, 64, 128), (320, 479, 128, 128), (320, 479, 256, 128),
          (160, 239, 128, 256), (160, 239, 256, 256), (160, 239, 512, 256), (80, 119, 256, 512), (80, 119, 512, 512),
          (80, 119, 1024, 512), (40, 59, 512, 1024), (40, 59, 1024, 1024)]
dt = torch.bfloat16
if os.environ.get('ONLY'):
    layers = [layers[int(i)] for i in os.environ['ONLY'].split(',')]
tot = 0.0
for (h, w, cin, cout) in layers:
    x = (torch.rand((N, h, w, cin), device='cuda') - 0.5).to(dt)
    dy = (torch.rand((N, h, w, cout), device='cuda') - 0.5).to(dt)
    dw = torch.empty((cout, cin, 3, 3), device='cuda')
    for _ in range(5):
        ops.conv3x3_wgrad(x, dy, dw)
    torch.cuda.synchronize()
    reps = 20
    t0 = time.perf_counter()
    for _ in range(reps):
        ops.conv3x3_wgrad(x, dy, dw)
    torch.cuda.synchronize()
    dtm = (time.perf_counter() - t0) / reps
    tot += dtm
    fl = 2.0 * N * h * w * 9 * cin * cout
    print(f'{h}x{w} {cin}->{cout}: {dtm*1e3:.3f} ms  {fl/dtm/1e12:.1f} TFLOP/s', flush=True)
print(f'total {tot*1e3:.2f} ms')
