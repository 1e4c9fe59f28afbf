import os, sys
ROOT = '/root/repo' if os.path.isdir('/root/repo/pytorch-unet_amd') else os.getcwd()
sys.path.insert(0, os.path.join(ROOT, 'pytorch-unet_amd'))
import torch
from unet import UNet
from unetk import _native
from unetk.trainer import Trainer
dev = torch.device('cuda', 0)
torch.manual_seed(0)
net = UNet(3, 21, False).to(dev).to(memory_format=torch.channels_last).train()
tr = Trainer(net, amp=False)
img = torch.rand((16, 3, 512, 512), device=dev).contiguous(memory_format=torch.channels_last)
mask = torch.randint(0, 21, (16, 512, 512), device=dev)
for _ in range(2): tr.step(img, mask)
torch.cuda.synchronize()
with _native.KernelTimer() as kt:
    tr.step(img, mask); torch.cuda.synchronize()
t = kt.summary()
tot = sum(v[1] for v in t.values())
for name, (n, m, f) in sorted(t.items(), key=lambda kv: -kv[1][1])[:22]:
    tf = f' {f / (m * 1e-3) / 1e12:7.1f} TFLOP/s' if f > 0 and m > 0 else ''
    print(f'  {name:34s} x{n:3d} {m:9.3f} ms {100 * m / tot:5.1f}%{tf}')
print('total', tot)
