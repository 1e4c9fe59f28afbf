#!/usr/bin/env python
"""Headline benchmark: U-Net training images/sec at 959x640, bf16, batch 16 per GPU (BASELINE.json configs[1]).

  python bench.py --gpus N --steps K --warmup W
  (N > 1: launched by torch.distributed.run, one rank per GPU, gradients all-reduced on RCCL)

A step = the reference's inner training loop body (train.py:97-117) on one batch of synthetic Carvana-shaped
tensors already resident in HBM: bf16-autocast forward, CE+dice loss, backward, clip_grad_norm_, RMSprop step.
Rank 0 prints ONE JSON line with the whole-job images/sec plus
  roofline     -- the dominant kernel (NHWC implicit-GEMM conv3x3 on MFMA, fprop+dgrad launches of one step):
                  algorithmic FLOPs / HIP-event time on the launch stream, against the 2.5 PFLOP/s dense bf16 peak
  cpu_baseline -- the reference's PyTorch CPU path (oracle/unet_torch.py, a stock-torch.nn restatement) timed on
                  the box's host cores on a bounded sample (batch 1), kind "port".
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
for _p in (ROOT, os.path.join(ROOT, 'pytorch-unet_amd')):
    if _p not in sys.path:
        sys.path.insert(0, _p)

import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

H, W = 640, 959
TRAIN_GFLOP_PER_IMG = 2693.7        # fwd + dgrad + wgrad, no dgrad for the first conv


def cpu_baseline(seconds_budget=25.0):
    """Reference CPU path (stock torch.nn restatement), fp32, channels_last, batch 1, full train step."""
    from oracle.unet_torch import TorchUNet, train_step
    cores = os.cpu_count() or 1
    threads = min(cores, int(os.environ.get('OMP_NUM_THREADS', cores)))
    torch.set_num_threads(threads)
    torch.manual_seed(0)
    net = TorchUNet(3, 2, False).to(memory_format=torch.channels_last).train()
    opt = torch.optim.RMSprop(net.parameters(), lr=1e-5, weight_decay=1e-8, momentum=0.999, foreach=True)
    img = torch.rand(1, 3, H, W).contiguous(memory_format=torch.channels_last)
    mask = (torch.rand(1, H, W) < 0.2).long()
    train_step(net, opt, img, mask)                     # warm-up
    t0 = time.perf_counter()
    n = 0
    while n < 3 and (n == 0 or time.perf_counter() - t0 < seconds_budget):
        train_step(net, opt, img, mask)
        n += 1
    dt = (time.perf_counter() - t0) / n
    return {'value': 1.0 / dt, 'unit': 'images/sec', 'cores': threads, 'kind': 'port',
            'sample': f'{n} fp32 train steps (fwd+CE+dice+bwd+clip+RMSprop) of batch 1 at 959x640 after 1 warm-up, '
                      f'torch {torch.__version__} CPU, {threads} threads'}


def pmc_traffic():
    """HBM bytes per conv3x3 fprop/dgrad launch from the committed rocprofv3 --pmc passes of THIS command
    (profiles/*_conv_traffic.json, made by scripts/pmc_summary.py: 2 x FETCH_SIZE + WRITE_SIZE, the gfx950
    correction of MI355X_MICROARCH.md 'HBM'); counters cannot be read from inside the timed process."""
    import glob
    files = sorted(glob.glob(os.path.join(ROOT, 'profiles', '*_conv_traffic.json')))
    if not files:
        return None
    try:
        with open(files[-1]) as f:
            return json.load(f).get('hbm_bytes_per_launch')
    except (OSError, ValueError):
        return None


def pmc_mfma_busy():
    """Matrix-pipe busy fraction of the conv3x3 fprop/dgrad launches from the committed counter pass of THIS command
    (profiles/*_mfma_util.json, scripts/pmc_mfma_util.py): SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE/8 * 256 * 4)."""
    import glob
    files = sorted(glob.glob(os.path.join(ROOT, 'profiles', '*_mfma_util.json')))
    if not files:
        return None
    try:
        with open(files[-1]) as f:
            return json.load(f).get('conv3x3_db_kernel', {}).get('mfma_busy_fraction')
    except (OSError, ValueError):
        return None


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=10)
    ap.add_argument('--warmup', type=int, default=3)
    ap.add_argument('--batch-size', '-b', type=int, default=16, help='images per GPU (weak scaling)')
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--height', type=int, default=H, help='sanity runs of other BASELINE configs; the metric is quoted at 640')
    ap.add_argument('--width', type=int, default=W)
    args = ap.parse_args()

    rank = int(os.environ.get('RANK', 0))
    world = int(os.environ.get('WORLD_SIZE', 1))
    local = int(os.environ.get('LOCAL_RANK', 0))
    torch.cuda.set_device(local)
    dev = torch.device('cuda', local)
    if world > 1:
        dist.init_process_group('nccl', device_id=dev)

    from unet import UNet
    from unetk import _native
    from unetk.trainer import Trainer

    torch.manual_seed(0)
    net = UNet(3, 2, False).to(dev).to(memory_format=torch.channels_last).train()
    if world > 1:                                    # identical replicas: broadcast rank 0's random init
        for t in list(net.parameters()) + list(net.buffers()):
            dist.broadcast(t.data, 0)
    trainer = Trainer(net, amp=True)
    g = torch.Generator(device=dev).manual_seed(1234 + rank)
    B = args.batch_size
    hh, ww = args.height, args.width
    images = torch.rand((B, 3, hh, ww), device=dev, generator=g).contiguous(memory_format=torch.channels_last)
    masks = (torch.rand((B, hh, ww), device=dev, generator=g) < 0.2).long()

    for _ in range(args.warmup):
        trainer.step(images, masks)

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        loss = trainer.step(images, masks)
    barrier()
    elapsed = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([elapsed], device=dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t)
    loss_val = float(loss)

    roofline = None
    table = {}
    # kernel timing pass: one more step (on EVERY rank: it contains the gradient collectives) with, on rank 0,
    # every C-ABI launch bracketed by HIP events on its stream
    if rank != 0:
        trainer.step(images, masks)
        torch.cuda.synchronize()
    else:
        with _native.KernelTimer() as kt:
            trainer.step(images, masks)
            torch.cuda.synchronize()
        table = kt.summary()
        n_l, ms, fl = 0, 0.0, 0.0
        for name in ('unetk_conv3x3_fprop', 'unetk_conv3x3_dgrad', 'unetk_conv3x3_dgrad_sums'):
            if name in table:
                n_l += table[name][0]
                ms += table[name][1]
                fl += table[name][2]
        if ms > 0:
            achieved = fl / (ms * 1e-3) / 1e12
            roofline = {'bound': 'mfma',
                        'kernel': 'conv3x3_db_kernel<bf16> (conv3x3 fprop + dgrad launches of one step; '
                                  '128-channel NB=4 and 64-channel NB=2 instantiations)',
                        'launches_per_step': n_l, 'avg_launch_ms': ms / n_l, 'achieved': achieved, 'peak': 2500.0,
                        'unit': 'TFLOP/s', 'frac': achieved / 2500.0, 'traffic': pmc_traffic(),
                        'mfma_busy_counters': pmc_mfma_busy()}
        total_ms = sum(v[1] for v in table.values())
        sys.stderr.write('kernel-time table of one step (HIP events, serialised by the event brackets):\n')
        for name, (n, m, f) in sorted(table.items(), key=lambda kv: -kv[1][1]):
            tf = f' {f / (m * 1e-3) / 1e12:7.1f} TFLOP/s' if f > 0 and m > 0 else ''
            sys.stderr.write(f'  {name:34s} x{n:3d} {m:9.3f} ms {100 * m / max(total_ms, 1e-9):5.1f}%{tf}\n')

    cpu = None
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        cpu = cpu_baseline()

    if rank == 0:
        ips = world * B * args.steps / elapsed
        out = {
            'metric': f'train images/sec @{ww}x{hh} bf16', 'value': ips, 'unit': 'images/sec', 'n_gpus': world,
            'steps': args.steps, 'warmup': args.warmup, 'ms_per_step': 1e3 * elapsed / args.steps,
            'higher_is_better': True, 'scaling': 'weak', 'vs_baseline': None, 'dtype': 'bf16', 'data': 'synthetic',
            'config': {'workload': 'UNet(3,2,bilinear=False) Carvana scale=0.5 (959x640) bf16 training step, '
                                   f'batch {B} per GPU (BASELINE.json configs[1])' if (hh, ww) == (H, W) else
                                   f'UNet(3,2,bilinear=False) {ww}x{hh} bf16 training step, batch {B} per GPU (sanity run)',
                       'global_batch': world * B, 'parallelism': f'dp{world}', 'optimizer': 'RMSprop+clip_grad_norm',
                       'loss': 'CE+dice'},
            'model_tflops': ips * TRAIN_GFLOP_PER_IMG * (hh * ww) / (H * W) / 1e3, 'final_loss': loss_val,
            'roofline': roofline, 'cpu_baseline': cpu,
        }
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.destroy_process_group()


if __name__ == '__main__':
    main()
