"""Host time to queue one training step against the GPU time it takes (is the launch thread ever the bottleneck?)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'pytorch-unet_amd'))
import torch
from unet import UNet
from unetk.trainer import Trainer

dev = torch.device('cuda', 0)
torch.manual_seed(0)
net = UNet(3, 2, False).to(dev).to(memory_format=torch.channels_last).train()
images = torch.rand((16, 3, 640, 959), device=dev).contiguous(memory_format=torch.channels_last)
masks = (torch.rand((16, 640, 959), device=dev) < 0.2).long()
tr = Trainer(net, amp=True)
for _ in range(3):
    tr.step(images, masks)
torch.cuda.synchronize()
K = 10
t0 = time.perf_counter()
for _ in range(K):
    tr.step(images, masks)
t1 = time.perf_counter()
torch.cuda.synchronize()
t2 = time.perf_counter()
print(f'host queues a step in {(t1 - t0) / K * 1e3:.1f} ms (loop returned), GPU finishes it in {(t2 - t0) / K * 1e3:.1f} ms per step')
