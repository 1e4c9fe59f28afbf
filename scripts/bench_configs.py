#!/usr/bin/env python
"""The OTHER BASELINE.json configurations on one GPU (bench.py keeps the headline, configs[1]):

  cfg3  UNet(3,2,bilinear=True).eval(), bf16 autocast, batch 64 @959x640 inference     (746.49 GFLOP/img)
  cfg4  UNet(3,2,False) bf16 train step, batch 8 @1918x1280 (the per-GPU share)         (10801.3 GFLOP/img)
  cfg5  UNet(3,21,False) fp32 train step, batch 16 @512x512, CE+dice multiclass         (1157.1 GFLOP/img)

  python scripts/bench_configs.py [cfg3 cfg4 cfg5] [--steps K --warmup W]
One JSON line per configuration (same keys as bench.py's line, no roofline / cpu_baseline legs).
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for _p in (ROOT, os.path.join(ROOT, 'pytorch-unet_amd')):
    if _p not in sys.path:
        sys.path.insert(0, _p)

import torch  # noqa: E402


def timed(fn, steps, warmup):
    for _ in range(warmup):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / steps


def line(metric, ips, dt, steps, warmup, dtype, workload, gflop_per_img, batch):
    return {'metric': metric, 'value': ips, 'unit': 'images/sec', 'n_gpus': 1, 'steps': steps, 'warmup': warmup,
            'ms_per_step': 1e3 * dt, 'higher_is_better': True, 'scaling': 'weak', 'vs_baseline': None, 'dtype': dtype,
            'data': 'synthetic', 'config': {'workload': workload, 'global_batch': batch, 'parallelism': 'dp1'},
            'model_tflops': ips * gflop_per_img / 1e3}


def cfg3(steps, warmup):
    from unet import UNet
    dev = torch.device('cuda', 0)
    torch.manual_seed(0)
    net = UNet(3, 2, True).to(dev).to(memory_format=torch.channels_last).eval()
    B = 64
    img = torch.rand((B, 3, 640, 959), device=dev).contiguous(memory_format=torch.channels_last)

    def fn():
        with torch.inference_mode(), torch.autocast('cuda', dtype=torch.bfloat16):
            return net(img)

    dt = timed(fn, steps, warmup)
    return line('inference images/sec @959x640 bf16 bilinear', B / dt, dt, steps, warmup, 'bf16',
                'UNet(3,2,bilinear=True).eval() bf16 inference, batch 64 (BASELINE.json configs[2])', 746.49, B)


def train_cfg(n_classes, amp, B, hh, ww, steps, warmup):
    from unet import UNet
    from unetk.trainer import Trainer
    dev = torch.device('cuda', 0)
    torch.manual_seed(0)
    net = UNet(3, n_classes, False).to(dev).to(memory_format=torch.channels_last).train()
    trainer = Trainer(net, amp=amp)
    g = torch.Generator(device=dev).manual_seed(1234)
    img = torch.rand((B, 3, hh, ww), device=dev, generator=g).contiguous(memory_format=torch.channels_last)
    if n_classes == 2:
        mask = (torch.rand((B, hh, ww), device=dev, generator=g) < 0.2).long()
    else:
        mask = torch.randint(0, n_classes, (B, hh, ww), device=dev, generator=g)
    return timed(lambda: trainer.step(img, mask), steps, warmup)


def cfg4(steps, warmup):
    B = 8
    dt = train_cfg(2, True, B, 1280, 1918, steps, warmup)
    return line('train images/sec @1918x1280 bf16', B / dt, dt, steps, warmup, 'bf16',
                'UNet(3,2,False) Carvana scale=1.0 bf16 training step, batch 8 per GPU (BASELINE.json configs[3], '
                'one rank of it)', 10801.3, B)


def cfg5(steps, warmup):
    B = 16
    dt = train_cfg(21, False, B, 512, 512, steps, warmup)
    return line('train images/sec @512x512 fp32 21 classes', B / dt, dt, steps, warmup, 'f32',
                'UNet(3,21,False) fp32 training step, CE+dice multiclass, batch 16 per GPU (BASELINE.json configs[4]; '
                'batch size not given there)', 1157.1, B)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('which', nargs='*', default=['cfg3', 'cfg4', 'cfg5'])
    ap.add_argument('--steps', type=int, default=5)
    ap.add_argument('--warmup', type=int, default=2)
    args = ap.parse_args()
    torch.cuda.set_device(0)
    for name in args.which:
        out = {'cfg3': cfg3, 'cfg4': cfg4, 'cfg5': cfg5}[name](args.steps, args.warmup)
        print(json.dumps(out), flush=True)
        torch.cuda.empty_cache()


if __name__ == '__main__':
    main()
