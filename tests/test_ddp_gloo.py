"""Data-parallel gradient bucketing (unetk/ddp.py) with world_size 2 on CPU/gloo: the flat buffer is laid out in
backward-completion order, buckets fire as blocks complete, and every rank ends with the average."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def _free_port():
    with socket.socket() as s:
        s.bind(('127.0.0.1', 0))
        return s.getsockname()[1]


def _worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group('gloo', rank=rank, world_size=world)
    try:
        from unet import UNet
        from unetk.ddp import FlatGrads
        torch.manual_seed(0)
        # parameter holders only; no kernel runs on CPU.  channels_last like train.py:197 / bench.py: autograd then
        # clones 4-D weight gradients instead of keeping the flat views, so the trainer must bind them itself
        net = UNet(3, 2, False).to(memory_format=torch.channels_last)
        sync = FlatGrads(net, bucket_bytes=4 << 20)
        assert sync.world == 2 and not sync.native_avg
        grads = sync.initial_grads()
        launched = []
        for blk in FlatGrads._completion_blocks(net):        # what the whole-net backward reports, in order
            for p in blk:
                grads[p].fill_(float(rank + 1))
                grads[p].view(-1)[0] = float(p.numel() % 97) * (rank + 1)
            sync.ready(blk, grads)
            launched.append(len(sync._works))
        assert launched[-1] >= 3 and launched[0] == 0        # several buckets fired before backward "ended"
        sync.finish()
        ok = True
        for p in net.parameters():
            g = grads[p]
            ok &= bool(torch.all(g.view(-1)[1:] == 1.5))
            ok &= abs(float(g.view(-1)[0]) - 1.5 * float(p.numel() % 97)) < 1e-6
        ok &= sync.total >= sum(p.numel() for p in net.parameters())
        # what a stock optimizer steps from: p.grad must BE the averaged flat view (Trainer's non-fused path)
        sync.bind_param_grads()
        lo, hi = sync.flat.data_ptr(), sync.flat.data_ptr() + 4 * sync.flat.numel()
        for p in net.parameters():
            ok &= lo <= p.grad.data_ptr() < hi and p.grad.shape == p.shape
            ok &= bool(torch.all(p.grad.reshape(-1)[1:] == 1.5))
        q.put((rank, ok))
    finally:
        dist.destroy_process_group()


def test_bucketed_allreduce_world2():
    port = _free_port()
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(180)
        assert p.exitcode == 0
    res = dict(q.get(timeout=5) for _ in range(2))
    assert res == {0: True, 1: True}


def test_flat_layout_is_completion_ordered():
    from unet import UNet
    from unetk.ddp import FlatGrads
    net = UNet(3, 2, True)
    sync = FlatGrads(net)
    offs = [sync.offsets[id(p)] for blk in FlatGrads._completion_blocks(net) for p in blk]
    assert offs == sorted(offs) and offs[0] == 0
    assert all(o % 4 == 0 for o in offs)                     # 16-byte aligned views
    assert {id(p) for p in net.parameters()} == set(sync.offsets)
