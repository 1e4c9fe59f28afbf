"""Kernel-time table of one BASELINE configs[2] batch (bilinear eval inference, bf16, batch 64)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'pytorch-unet_amd'))
import torch
from unet import UNet
from unetk import _native
dev = torch.device('cuda', 0)
torch.manual_seed(0)
net = UNet(3, 2, True).to(dev).to(memory_format=torch.channels_last).eval()
img = torch.rand((64, 3, 640, 959), device=dev).contiguous(memory_format=torch.channels_last)
def fn():
    with torch.inference_mode(), torch.autocast('cuda', dtype=torch.bfloat16):
        return net(img)
for _ in range(2): fn()
torch.cuda.synchronize()
with _native.KernelTimer() as kt:
    fn(); torch.cuda.synchronize()
t = kt.summary()
tot = sum(v[1] for v in t.values())
for name, (n, m, f) in sorted(t.items(), key=lambda kv: -kv[1][1]):
    tf = f' {f / (m * 1e-3) / 1e12:7.1f} TFLOP/s' if f > 0 and m > 0 else ''
    print(f'  {name:34s} x{n:3d} {m:9.3f} ms {100 * m / tot:5.1f}%{tf}')
print('total', tot)
