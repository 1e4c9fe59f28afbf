#!/usr/bin/env python
"""Headline benchmark: U-Net training images/sec at 959x640, bf16, batch 16 per GPU (BASELINE.json configs[1]).

  python bench.py --gpus N --steps K --warmup W [--config NAME]

N > 1: one rank per GPU, gradients bucket-all-reduced on RCCL.  Either the caller starts the ranks
(`python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N`: RANK/WORLD_SIZE are then in the
environment) or, when RANK is NOT set, this process starts them itself -- N fresh children through
torch.distributed.run, before anything here has touched a GPU -- and relays rank 0's JSON line.

A step = the reference's inner training loop body (train.py:97-117) on one batch of synthetic Carvana-shaped
tensors already resident in HBM: bf16-autocast forward, CE+dice loss, backward, clip_grad_norm_, RMSprop step.
Rank 0 prints ONE JSON line with the whole-job images/sec plus
  roofline     -- the dominant kernel (NHWC implicit-GEMM conv3x3 on MFMA, fprop+dgrad launches of one step):
                  algorithmic FLOPs / HIP-event time on the launch stream, against the dense MFMA peak of the dtype
  roofline_hbm -- the HBM-bound kernel families of the same step (BatchNorm+ReLU apply / backward reduce / backward
                  apply incl. their pool and head variants, ConvTranspose2d, first layer, loss, optimizer):
                  algorithmic bytes from the call arguments / HIP-event time, against 8 TB/s
  cpu_baseline -- the reference's PyTorch CPU path (oracle/unet_torch.py, a stock-torch.nn restatement) timed on
                  the box's host cores on a bounded sample (batch 1), kind "port" (N = 1, default config only): the
                  train step, and beside it BASELINE.json configs[0] (fp32 eval forward, N = 1).
  other_configs -- (default command only) BASELINE.json configs[2..4] run through this same harness afterwards, one child
                  process each (5 steps, 2 warm-up): value, ms/step and the roofline fraction of each; context beside
                  the headline value, skipped with --no-other-configs or --no-cpu-baseline (the quick forms)
--force-collectives: run the gradient buckets through a 1-rank RCCL group (what one GPU can say about the
data-parallel path: launch + stream hand-over + CU contention of RCCL's kernels against the whole-CU conv kernels).

--config selects the other BASELINE.json configurations (same JSON, own roofline leg); the default is the headline:
  train959            configs[1]  UNet(3,2,False) bf16 train step, batch 16 @959x640
  infer959_bilinear   configs[2]  UNet(3,2,True).eval() bf16 inference, batch 64 @959x640
  train1918           configs[3]  UNet(3,2,False) bf16 train step, batch 8 per GPU @1918x1280
  train512_c21        configs[4]  UNet(3,21,False) fp32 train step, CE+dice, batch 16 per GPU @512x512

--backend gloo --dry: CPU rehearsal of the multi-rank plumbing (launcher, rank environment, flat gradient buffer,
bucketed all-reduce, max-over-ranks timing) with NO kernels -- the product has no CPU path; the line says "dry": true
and its value is not a measurement.
"""
import argparse
import json
import os
import socket
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
for _p in (ROOT, os.path.join(ROOT, 'pytorch-unet_amd')):
    if _p not in sys.path:
        sys.path.insert(0, _p)

H, W = 640, 959

# name -> (n_classes, bilinear, train, amp, batch per GPU, H, W, GFLOP per image (BASELINE.md / SURVEY 8d), label)
CONFIGS = {
    'train959': dict(n_classes=2, bilinear=False, train=True, amp=True, batch=16, h=640, w=959, gflop=2693.7,
                     metric='train images/sec @959x640 bf16',
                     workload='UNet(3,2,bilinear=False) Carvana scale=0.5 (959x640) bf16 training step, '
                              'batch {B} per GPU (BASELINE.json configs[1])'),
    'infer959_bilinear': dict(n_classes=2, bilinear=True, train=False, amp=True, batch=64, h=640, w=959, gflop=746.49,
                              metric='inference images/sec @959x640 bf16 bilinear',
                              workload='UNet(3,2,bilinear=True).eval() bf16 inference, batch {B} per GPU '
                                       '(BASELINE.json configs[2])'),
    'train1918': dict(n_classes=2, bilinear=False, train=True, amp=True, batch=8, h=1280, w=1918, gflop=10801.3,
                      metric='train images/sec @1918x1280 bf16',
                      workload='UNet(3,2,bilinear=False) Carvana scale=1.0 (1918x1280) bf16 training step, '
                               'batch {B} per GPU (BASELINE.json configs[3])'),
    'train512_c21': dict(n_classes=21, bilinear=False, train=True, amp=False, batch=16, h=512, w=512, gflop=1157.1,
                         metric='train images/sec @512x512 fp32 21 classes',
                         workload='UNet(3,21,bilinear=False) fp32 training step, CE+dice multiclass, batch {B} per GPU '
                                  '(BASELINE.json configs[4]; batch size not given there)'),
}
PEAK_TFLOPS = {'bf16': 2500.0, 'fp16': 2500.0, 'f32': 157.3}      # MI355X_MICROARCH.md: dense bf16 / fp16 MFMA; fp32-input MFMA


def parse_args(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=10)
    ap.add_argument('--warmup', type=int, default=3)
    ap.add_argument('--config', choices=sorted(CONFIGS), default='train959')
    ap.add_argument('--batch-size', '-b', type=int, default=None, help='images per GPU (weak scaling); default per config')
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--no-other-configs', action='store_true',
                    help='default line only: do not run BASELINE.json configs[2..4] as child processes afterwards')
    ap.add_argument('--height', type=int, default=None, help='sanity runs at other sizes; the metric is quoted at the config size')
    ap.add_argument('--width', type=int, default=None)
    ap.add_argument('--backend', choices=['nccl', 'gloo'], default='nccl')
    ap.add_argument('--dry', action='store_true', help='CPU rehearsal of the rank plumbing; no kernels, no measurement')
    ap.add_argument('--amp-dtype', choices=['bf16', 'fp16'], default='bf16',
                    help="autocast dtype of the amp configurations: bf16 (BASELINE.json) or fp16 (the reference's --amp "
                         "default on a GPU, run with its GradScaler); not the headline line")
    ap.add_argument('--force-collectives', action='store_true',
                    help='N = 1: all-reduce the gradient buckets in a 1-rank RCCL group anyway (overhead measurement)')
    return ap.parse_args(argv)


def free_port():
    with socket.socket() as s:
        s.bind(('127.0.0.1', 0))
        return s.getsockname()[1]


def launch_ranks(args, argv):
    """--gpus N without a rank environment: start N FRESH ranks (this process has not touched a GPU and never will)
    and relay rank 0's JSON line.  Returns the children's exit code."""
    cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', f'--nproc-per-node={args.gpus}',
           '--master-addr', '127.0.0.1', '--master-port', str(free_port()), os.path.abspath(__file__)] + list(argv)
    env = dict(os.environ)
    env.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')      # dmabuf IPC: RCCL across processes needs it on this image
    proc = subprocess.Popen(cmd, stdout=subprocess.PIPE, env=env, text=True)
    json_line = None
    for line in proc.stdout:
        s = line.strip()
        if s.startswith('{') and '"metric"' in s:
            json_line = s
        else:
            sys.stderr.write(line)
    rc = proc.wait()
    if json_line is not None:
        print(json_line, flush=True)
    elif rc == 0:
        sys.stderr.write('bench.py: the ranks exited cleanly but printed no result line\n')
        rc = 1
    return rc


def other_configs(steps=5, warmup=2, limit_s=150):
    """BASELINE.json configs[2..4] through this same harness, one fresh child process each (sequentially, after the
    headline measurement; this process is idle meanwhile), so that the driver's record of the default command carries
    them too.  Context beside the headline value, never part of it; a child that fails or overruns leaves an error
    entry instead of a number."""
    res = {}
    for name in ('infer959_bilinear', 'train1918', 'train512_c21'):
        cmd = [sys.executable, os.path.abspath(__file__), '--config', name, '--steps', str(steps), '--warmup', str(warmup),
               '--no-cpu-baseline', '--no-other-configs']
        t0 = time.perf_counter()
        try:
            p = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.DEVNULL, text=True, timeout=limit_s)
            line = next((l for l in reversed(p.stdout.splitlines()) if l.startswith('{') and '"metric"' in l), None)
            if p.returncode != 0 or line is None:
                res[name] = {'error': f'exit code {p.returncode}, no result line' if line is None else f'exit code {p.returncode}'}
                continue
            d = json.loads(line)
            r = d.get('roofline') or {}
            res[name] = {'metric': d['metric'], 'value': d['value'], 'unit': d['unit'], 'ms_per_step': d['ms_per_step'],
                         'steps': d['steps'], 'warmup': d['warmup'], 'dtype': d['dtype'], 'workload': d['config']['workload'],
                         'global_batch': d['config']['global_batch'], 'model_tflops': d['model_tflops'],
                         'roofline': {k: r.get(k) for k in ('kernel', 'launches_per_step', 'avg_launch_ms', 'achieved', 'peak',
                                                            'unit', 'frac')},
                         'wall_s': round(time.perf_counter() - t0, 1)}
        except subprocess.TimeoutExpired:
            res[name] = {'error': f'no result within {limit_s} s'}
    return res


def cpu_baseline(seconds_budget=25.0):
    """Reference CPU path (stock torch.nn restatement), fp32, channels_last, batch 1, full train step."""
    import torch
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
    # BASELINE.json configs[0] beside it: UNet(3,2,False) fp32 eval forward, N = 1 (unet_model.py:25-36)
    net.eval()
    with torch.no_grad():
        net(img)                                        # warm-up
        t1 = time.perf_counter()
        m = 0
        while m < 3 and (m == 0 or time.perf_counter() - t1 < 8.0):
            net(img)
            m += 1
        dtf = (time.perf_counter() - t1) / m
    return {'value': 1.0 / dt, 'unit': 'images/sec', 'cores': threads, 'kind': 'port',
            'sample': f'{n} fp32 train steps (fwd+CE+dice+bwd+clip+RMSprop) of batch 1 at 959x640 after 1 warm-up, '
                      f'torch {torch.__version__} CPU, {threads} threads',
            'forward_n1': {'value': 1.0 / dtf, 'unit': 'images/sec', 'ms': 1e3 * dtf,
                           'sample': f'{m} fp32 eval forwards of batch 1 at 959x640 (BASELINE.json configs[0]) after '
                                     f'1 warm-up, same threads'}}


def committed_profile(pattern, getter):
    """A number from the newest committed rocprofv3 --pmc summary under profiles/ (counters cannot be read from
    inside the timed process).  Returns {'value', 'source'} so the line says where and when it was taken."""
    import glob
    files = sorted(glob.glob(os.path.join(ROOT, 'profiles', pattern)))
    if not files:
        return None
    try:
        with open(files[-1]) as f:
            doc = json.load(f)
        v = getter(doc)
    except (OSError, ValueError, AttributeError):
        return None
    if v is None:
        return None
    return {'value': v, 'source': os.path.relpath(files[-1], ROOT), 'profiled_at': doc.get('commit') or doc.get('bench_sha')}


def dry_run(args, rank, world):
    """CPU/gloo rehearsal: the flat completion-ordered gradient buffer and its bucketed all-reduce, no kernels."""
    import torch
    import torch.distributed as dist
    from unet import UNet
    from unetk.ddp import FlatGrads
    torch.manual_seed(0)
    net = UNet(3, 2, False)
    sync = FlatGrads(net, bucket_bytes=25 << 20)

    def step():
        grads = sync.initial_grads()
        for blk in FlatGrads._completion_layers(net):
            for p in blk:
                grads[p].fill_(float(rank + 1))
            sync.ready(blk, grads)
        sync.finish()
        sync.consume()                                  # the optimizer step of the real loop
        return grads

    for _ in range(args.warmup):
        step()
    if world > 1:
        dist.barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        grads = step()
    if world > 1:
        dist.barrier()
    elapsed = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t)
    want = sum(range(1, world + 1)) / world
    ok = all(bool(torch.all(g == want)) for g in grads.values())
    if not ok:
        raise RuntimeError(f'rank {rank}: averaged gradients are wrong')
    return elapsed


def main(argv=None):
    argv = sys.argv[1:] if argv is None else argv
    args = parse_args(argv)
    if args.gpus < 1:
        sys.exit('bench.py: --gpus must be >= 1')
    if 'RANK' not in os.environ and args.gpus > 1:
        sys.exit(launch_ranks(args, argv))

    import torch
    import torch.distributed as dist

    rank = int(os.environ.get('RANK', 0))
    world = int(os.environ.get('WORLD_SIZE', 1))
    local = int(os.environ.get('LOCAL_RANK', 0))
    if world != args.gpus:
        sys.exit(f'bench.py: --gpus {args.gpus} but the launcher started WORLD_SIZE={world} ranks')
    cfg = CONFIGS[args.config]
    B = args.batch_size or cfg['batch']
    hh, ww = args.height or cfg['h'], args.width or cfg['w']
    dtype_name = ('fp16' if args.amp_dtype == 'fp16' else 'bf16') if cfg['amp'] else 'f32'
    amp_dtype = torch.float16 if dtype_name == 'fp16' else torch.bfloat16

    if args.dry:
        if args.backend != 'gloo':
            sys.exit('bench.py: --dry is the CPU rehearsal, use --backend gloo')
        if world > 1:
            dist.init_process_group('gloo')
        elapsed = dry_run(args, rank, world)
        if rank == 0:
            print(json.dumps({'metric': cfg['metric'], 'value': None, 'unit': 'images/sec', 'n_gpus': world,
                              'steps': args.steps, 'warmup': args.warmup, 'ms_per_step': 1e3 * elapsed / args.steps,
                              'higher_is_better': True, 'scaling': 'weak', 'vs_baseline': None, 'dtype': dtype_name,
                              'data': 'none', 'dry': True, 'backend': 'gloo',
                              'config': {'workload': 'DRY RUN: gradient all-reduce plumbing only, no kernels',
                                         'global_batch': world * B, 'parallelism': f'dp{world}'}}), flush=True)
        if world > 1:
            dist.destroy_process_group()
        return

    torch.cuda.set_device(local)
    dev = torch.device('cuda', local)
    if world > 1:
        dist.init_process_group(args.backend, device_id=dev if args.backend == 'nccl' else None)
    elif args.force_collectives:
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        os.environ.setdefault('MASTER_PORT', str(free_port()))
        dist.init_process_group('nccl', rank=0, world_size=1, device_id=dev)

    from unet import UNet
    from unetk import _native
    from unetk.trainer import Trainer

    torch.manual_seed(0)
    net = UNet(3, cfg['n_classes'], cfg['bilinear']).to(dev)
    if world > 1:                                    # identical replicas: broadcast rank 0's random init (while the
        for t in list(net.parameters()) + list(net.buffers()):   # tensors are still plainly contiguous)
            dist.broadcast(t.data, 0)
    net = net.to(memory_format=torch.channels_last)
    net = net.train() if cfg['train'] else net.eval()
    g = torch.Generator(device=dev).manual_seed(1234 + rank)
    images = torch.rand((B, 3, hh, ww), device=dev, generator=g).contiguous(memory_format=torch.channels_last)
    if cfg['n_classes'] == 2:
        masks = (torch.rand((B, hh, ww), device=dev, generator=g) < 0.2).long()
    else:
        masks = torch.randint(0, cfg['n_classes'], (B, hh, ww), device=dev, generator=g)

    if cfg['train']:
        trainer = Trainer(net, amp=cfg['amp'], force_collectives=args.force_collectives, amp_dtype=amp_dtype)

        def step():
            return trainer.step(images, masks)
    else:
        def step():
            with torch.inference_mode(), torch.autocast('cuda', dtype=amp_dtype, enabled=cfg['amp']):
                return net(images)

    for _ in range(args.warmup):
        step()

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        res = step()
    barrier()
    elapsed = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([elapsed], device=dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t)
    loss_val = float(res) if cfg['train'] else None

    roofline = roofline_hbm = None
    # kernel timing passes: TIMING_PASSES more steps (on EVERY rank: they contain the gradient collectives) with, on rank
    # 0, every C-ABI launch bracketed by HIP events on its stream; per launch the MEDIAN of the passes counts (an event
    # pair also spans the time the GPU waits for the host to submit the launch: one hiccup of the launching thread
    # would otherwise be booked as kernel time)
    TIMING_PASSES = 3
    if rank != 0:
        for _ in range(TIMING_PASSES):
            step()
        torch.cuda.synchronize()
    else:
        passes, pass_ms = [], []
        for _ in range(TIMING_PASSES):
            tp = time.perf_counter()
            with _native.KernelTimer() as one:
                step()
                torch.cuda.synchronize()
            pass_ms.append(1e3 * (time.perf_counter() - tp))
            passes.append(one)
        kt = passes[0]
        kt.take_median(passes[1:])
        table = kt.summary()
        # the conv3x3 fprop/dgrad launches of the step by the device kernel that served them; the roofline object is the
        # DOMINANT kernel's (bf16: conv3x3_k32_kernel, the 128-channel n-tiles -- its rocprofv3 --kernel-trace row must
        # show the same average duration; fp32: conv3x3_db_kernel), the whole family rides along as `conv3x3_family`
        per_kernel = kt.by_device_kernel()
        dom = max(per_kernel, key=lambda k: per_kernel[k][1]) if per_kernel else None
        n_l, ms, fl, alg_bytes = per_kernel.get(dom, (0, 0.0, 0.0, 0.0))
        fam = [sum(v[i] for v in per_kernel.values()) for i in range(3)]
        if ms > 0:
            achieved = fl / (ms * 1e-3) / 1e12
            peak = PEAK_TFLOPS[dtype_name]
            what = {'conv3x3_k32_kernel': 'GEMM-N a multiple of 128, v_mfma_f32_16x16x32_bf16',
                    'conv3x3_k32n64_kernel': 'GEMM-N = 64, v_mfma_f32_16x16x32_bf16',
                    'conv3x3_db_kernel': 'fp32: every conv3x3 launch, v_mfma_f32_32x32x2_f32'}[dom]
            roofline = {'bound': 'mfma',
                        'kernel': f'{dom} (conv3x3 fprop{" + dgrad" if cfg["train"] else ""} launches of one step: {what})',
                        'launches_per_step': n_l, 'avg_launch_ms': ms / n_l, 'achieved': achieved, 'peak': peak,
                        'timing': f'per-launch median of {TIMING_PASSES} event-bracketed steps after the timed region',
                        'timing_pass_ms': [round(v, 3) for v in pass_ms],
                        'unit': 'TFLOP/s', 'frac': achieved / peak, 'traffic': None,
                        'algorithmic_bytes_per_launch': alg_bytes / n_l,
                        'conv3x3_family': {
                            'kernels': {k: {'launches': v[0], 'avg_launch_ms': v[1] / v[0],
                                            'achieved': v[2] / (v[1] * 1e-3) / 1e12,
                                            'frac': v[2] / (v[1] * 1e-3) / 1e12 / peak} for k, v in per_kernel.items()},
                            'launches_per_step': fam[0], 'achieved': fam[2] / (fam[1] * 1e-3) / 1e12,
                            'frac': fam[2] / (fam[1] * 1e-3) / 1e12 / peak,
                            'note': 'launches with a fused BN-backward reduction are counted with their conv FLOPs only'}}
            if args.config == 'train959' and (hh, ww, B) == (H, W, 16):
                # counters of THIS command from committed --pmc passes; only quoted for the configuration they were
                # taken on, with their source
                tr = committed_profile('*_conv_traffic.json',
                                       lambda d: d.get('hbm_bytes_per_launch') if d.get('kernel', dom) == dom else None)
                mb = committed_profile('*_mfma_util.json', lambda d: d.get(dom, {}).get('mfma_busy_fraction'))
                if tr:
                    roofline['traffic'] = tr['value']
                    roofline['traffic_source'] = tr['source']
                if mb:
                    roofline['mfma_busy_counters'] = mb['value']
                    roofline['mfma_busy_source'] = mb['source']
            if dtype_name == 'bf16' and dom != 'conv3x3_db_kernel':
                # a measured-on-box matrix-pipe peak beside the nominal one -- the instruction this kernel
                # issues, back to back from registers on random operands (scripts/microbench/mfma_peak.hip)
                pk = committed_profile('*_mfma_peak.json', lambda d: d.get('v_mfma_f32_16x16x32_bf16', {})
                                       .get('random', {}).get('steady_median_tflops'))
                if pk:
                    roofline['peak_measured'] = pk['value']
                    roofline['frac_of_measured'] = achieved / pk['value']
                    roofline['peak_measured_source'] = pk['source']
        fams = kt.by_hbm_family()
        if fams:
            peak_gbs = 8000.0                              # MI355X_MICROARCH.md: HBM3E ~8 TB/s
            def leg(n, ms_, b):
                gbs = b / (ms_ * 1e-3) / 1e9 if ms_ > 0 else 0.0
                return {'launches': n, 'ms_per_step': ms_, 'algorithmic_bytes': b, 'achieved': gbs, 'frac': gbs / peak_gbs}
            roofline_hbm = {'bound': 'hbm', 'peak': peak_gbs, 'unit': 'GB/s',
                            'families': {k: leg(*v) for k, v in sorted(fams.items(), key=lambda kv: -kv[1][1])}}
            bn = [v for k, v in fams.items() if k.startswith('bn_')]
            if bn:
                roofline_hbm['batchnorm_family'] = leg(*[sum(v[i] for v in bn) for i in range(3)])
            roofline_hbm['all'] = leg(*[sum(v[i] for v in fams.values()) for i in range(3)])
            roofline_hbm['note'] = ('algorithmic bytes = every tensor a pass must read or write, once, from the call '
                                    'arguments; time = HIP events around each launch on its stream (one step; per-launch median of the '
                                    'timing passes)')
        total_ms = sum(v[1] for v in table.values())
        sys.stderr.write('kernel-time table of one step (HIP events, serialised by the event brackets):\n')
        for name, (n, m, f) in sorted(table.items(), key=lambda kv: -kv[1][1]):
            tf = f' {f / (m * 1e-3) / 1e12:7.1f} TFLOP/s' if f > 0 and m > 0 else ''
            sys.stderr.write(f'  {name:34s} x{n:3d} {m:9.3f} ms {100 * m / max(total_ms, 1e-9):5.1f}%{tf}\n')

    cpu = None
    if rank == 0 and world == 1 and not args.no_cpu_baseline and args.config == 'train959':
        cpu = cpu_baseline()
    others = None
    if rank == 0 and world == 1 and not (args.no_other_configs or args.no_cpu_baseline) and args.config == 'train959' and not args.force_collectives \
            and (hh, ww) == (cfg['h'], cfg['w']) and args.amp_dtype == 'bf16' and args.batch_size is None:
        others = other_configs()

    if rank == 0:
        ips = world * B * args.steps / elapsed
        sized = (hh, ww) == (cfg['h'], cfg['w']) and dtype_name != 'fp16'     # the metric is quoted in bf16 / fp32
        out = {
            'metric': cfg['metric'] if sized else f'images/sec @{ww}x{hh} {dtype_name}', 'value': ips,
            'unit': 'images/sec', 'n_gpus': world,
            'steps': args.steps, 'warmup': args.warmup, 'ms_per_step': 1e3 * elapsed / args.steps,
            'higher_is_better': True, 'scaling': 'weak', 'vs_baseline': None, 'dtype': dtype_name, 'data': 'synthetic',
            'config': {'workload': cfg['workload'].format(B=B) if sized else
                       f'{args.config} at {ww}x{hh} {dtype_name}, batch {B} per GPU (not the headline configuration)',
                       'global_batch': world * B, 'parallelism': f'dp{world}',
                       'optimizer': 'RMSprop+clip_grad_norm' if cfg['train'] else None,
                       'loss': 'CE+dice' if cfg['train'] else None},
            'model_tflops': ips * cfg['gflop'] * (hh * ww) / (cfg['h'] * cfg['w']) / 1e3, 'final_loss': loss_val,
            'roofline': roofline, 'roofline_hbm': roofline_hbm, 'cpu_baseline': cpu,
        }
        if others is not None:
            out['other_configs'] = others
        if args.force_collectives:
            out['force_collectives'] = {'backend': 'nccl', 'world': world,
                                        'collectives_per_step': len(trainer.sync.launched) if cfg['train'] else 0,
                                        'largest_bytes': 4 * max((b - a for a, b in trainer.sync.launched), default=0),
                                        'last_bytes': 4 * (trainer.sync.launched[-1][1] - trainer.sync.launched[-1][0])
                                        if cfg['train'] and trainer.sync.launched else 0}
        print(json.dumps(out), flush=True)
    if world > 1 or args.force_collectives:
        dist.destroy_process_group()


if __name__ == '__main__':
    main()
