import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, 'pytorch-unet_amd')]
import torch
from unet import UNet
from unetk.trainer import Trainer
net = UNet(3, 2, False).cuda().to(memory_format=torch.channels_last).train()
tr = Trainer(net)
img = torch.rand((2, 3, 128, 160), device='cuda').contiguous(memory_format=torch.channels_last)
mask = (torch.rand((2, 128, 160), device='cuda') < 0.2).long()
for _ in range(3): tr.step(img, mask)
torch.cuda.synchronize()
flat = tr.sync.flat
def t(f, reps=20):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(reps): f()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / reps * 1e3
def clip():
    n = torch.linalg.vector_norm(flat); flat.mul_(torch.clamp(1.0 / (n + 1e-6), max=1.0))
print('clip ms', t(clip)); print('opt.step ms', t(tr.optimizer.step)); print('zeros ms', t(lambda: torch.zeros(tr.sync.total, device='cuda')))
