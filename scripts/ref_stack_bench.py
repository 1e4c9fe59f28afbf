"""The reference's layer stack in STOCK PyTorch on the same GPU: the number this build is compared with in DESIGN.md 4.

    python scripts/ref_stack_bench.py [--steps 10] [--warmup 3] [--batch 16]  > gpurun_out/ref_stack.log

A U-Net of plain torch.nn modules with the reference's architecture (unet/unet_model.py:6-36, unet_parts.py:8-77:
DoubleConv = (Conv2d 3x3 no bias, BatchNorm2d, ReLU) x 2; Down = MaxPool2d(2) + DoubleConv; Up = ConvTranspose2d k2 s2,
F.pad, cat, DoubleConv; OutConv = Conv2d k1), trained exactly as train.py:97-117 trains it -- channels_last, bf16
autocast, CrossEntropyLoss + dice loss, clip_grad_norm_(1.0), RMSprop(lr 1e-5, weight_decay 1e-8, momentum 0.999,
foreach) -- at BASELINE.json configs[1] (batch 16, 3 x 640 x 959, synthetic data).  PyTorch 2.10-ROCm runs its convolutions
through MIOpen, whose first call per shape searches for a solver (minutes); a heartbeat line is printed meanwhile.
Nothing of this repository's library is loaded.
"""
import argparse
import json
import sys
import threading
import time

import torch
import torch.nn as nn
import torch.nn.functional as F


def double_conv(cin, cout):
    return nn.Sequential(nn.Conv2d(cin, cout, 3, padding=1, bias=False), nn.BatchNorm2d(cout), nn.ReLU(inplace=True),
                         nn.Conv2d(cout, cout, 3, padding=1, bias=False), nn.BatchNorm2d(cout), nn.ReLU(inplace=True))


class StockUNet(nn.Module):
    def __init__(self, n_channels=3, n_classes=2):
        super().__init__()
        self.n_classes = n_classes
        self.inc = double_conv(n_channels, 64)
        self.downs = nn.ModuleList([nn.Sequential(nn.MaxPool2d(2), double_conv(c, 2 * c)) for c in (64, 128, 256, 512)])
        self.ups = nn.ModuleList([nn.ConvTranspose2d(c, c // 2, 2, stride=2) for c in (1024, 512, 256, 128)])
        self.up_convs = nn.ModuleList([double_conv(c, c // 2) for c in (1024, 512, 256, 128)])
        self.outc = nn.Conv2d(64, n_classes, 1)

    def forward(self, x):
        skips = [self.inc(x)]
        for d in self.downs:
            skips.append(d(skips[-1]))
        y = skips.pop()
        for up, conv in zip(self.ups, self.up_convs):
            s = skips.pop()
            y = up(y)
            dy, dx = s.size(2) - y.size(2), s.size(3) - y.size(3)
            y = F.pad(y, [dx // 2, dx - dx // 2, dy // 2, dy - dy // 2])
            y = conv(torch.cat([s, y], dim=1))
        return self.outc(y)


def dice_loss(probs, onehot, eps=1e-6):                      # utils/dice_score.py:5-28, multiclass=True
    p, t = probs.flatten(0, 1), onehot.flatten(0, 1)
    inter = 2 * (p * t).sum((-1, -2))
    sets = p.sum((-1, -2)) + t.sum((-1, -2))
    sets = torch.where(sets == 0, inter, sets)
    return 1 - ((inter + eps) / (sets + eps)).mean()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--steps', type=int, default=10)
    ap.add_argument('--warmup', type=int, default=3)
    ap.add_argument('--batch', type=int, default=16)
    a = ap.parse_args()
    stop = threading.Event()

    def heartbeat():
        t0 = time.time()
        while not stop.wait(45):
            print(f'[heartbeat] {time.time() - t0:.0f} s', flush=True)
    threading.Thread(target=heartbeat, daemon=True).start()

    torch.manual_seed(0)
    dev = torch.device('cuda')
    net = StockUNet().to(dev).to(memory_format=torch.channels_last).train()
    opt = torch.optim.RMSprop(net.parameters(), lr=1e-5, weight_decay=1e-8, momentum=0.999, foreach=True)
    crit = nn.CrossEntropyLoss()
    img = torch.rand(a.batch, 3, 640, 959, device=dev).contiguous(memory_format=torch.channels_last)
    mask = torch.randint(0, 2, (a.batch, 640, 959), device=dev)

    def step():
        with torch.autocast('cuda', dtype=torch.bfloat16):
            pred = net(img)
            loss = crit(pred, mask)
            loss = loss + dice_loss(F.softmax(pred, dim=1).float(), F.one_hot(mask, 2).permute(0, 3, 1, 2).float())
        opt.zero_grad(set_to_none=True)
        loss.backward()
        torch.nn.utils.clip_grad_norm_(net.parameters(), 1.0)
        opt.step()
        return loss

    t0 = time.time()
    step()
    torch.cuda.synchronize()
    first = time.time() - t0
    print(f'first step (includes MIOpen solver search): {first:.1f} s', flush=True)
    for _ in range(a.warmup):
        step()
    torch.cuda.synchronize()
    t0 = time.time()
    for _ in range(a.steps):
        step()
    torch.cuda.synchronize()
    dt = (time.time() - t0) / a.steps
    stop.set()
    print(json.dumps({'stack': f'torch {torch.__version__} stock nn modules (MIOpen convolutions)', 'config': 'train959 bf16',
                      'batch': a.batch, 'ms_per_step': dt * 1e3, 'images_per_sec': a.batch / dt,
                      'first_step_seconds': first, 'steps': a.steps, 'warmup': a.warmup}), flush=True)


if __name__ == '__main__':
    sys.exit(main())
