"""Data-parallel gradient bucketing (unetk/ddp.py) with world_size 2 on CPU/gloo: the flat buffer is laid out in
backward-completion order, buckets fire as layers complete, and every rank ends with the average."""
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
        for blk in FlatGrads._completion_layers(net):        # what the whole-net backward reports, in order
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
    offs = [sync.offsets[id(p)] for blk in FlatGrads._completion_layers(net) for p in blk]
    assert offs == sorted(offs) and offs[0] == 0
    assert all(o % 4 == 0 for o in offs)                     # 16-byte aligned views
    assert {id(p) for p in net.parameters()} == set(sync.offsets)
    # block granularity is the same order, coarser
    offs_b = [sync.offsets[id(p)] for blk in FlatGrads._completion_blocks(net) for p in blk]
    assert offs_b == offs


def _bucket_schedule(net, **kw):
    """The collectives one backward pass would launch (ranges in elements), without a process group."""
    from unetk.ddp import FlatGrads
    sync = FlatGrads(net, **kw)
    sync.active = True                                       # decide as a multi-rank job would ...
    fired = []
    sync._launch = lambda: (fired.append(tuple(sync._pending)), sync._pending.__setitem__(slice(None), [sync._pending[1]] * 2))
    sync.flat = torch.zeros(1)
    for layer in FlatGrads._completion_layers(net):
        sync.ready(layer, None)
    tail = tuple(sync._pending)                              # ... what finish() would flush after backward
    return sync, fired, tail


def test_bucket_schedule_of_the_reference_unet():
    """Headline model, default 25 MB buckets: no collective above ~30 MB (the 37.7 MB 1024->1024 layer is split), and the
    bucket that is flushed AFTER backward -- fully exposed -- is inc + down1 at most (<= 2 MB)."""
    from unet import UNet
    from unetk.ddp import FlatGrads
    net = UNet(3, 2, False)
    sync, fired, tail = _bucket_schedule(net)
    assert fired and fired[0][0] == 0
    for (a, b), (c, d) in zip(fired, fired[1:] + [tail]):
        assert b == c                                        # contiguous, nothing skipped
    assert tail[1] == sync.total
    assert 4 * (tail[1] - tail[0]) <= 2 << 20
    # the pieces a range is cut into (FlatGrads._launch): equal, <= bucket_bytes
    for lo, hi in fired:
        pieces = -(-(hi - lo) // sync.bucket_elems)
        step = (-(-(hi - lo) // pieces) + 3) // 4 * 4
        assert 4 * step <= 30 << 20
    # a handful of collectives, not one per layer: the latency of each is paid on RCCL's stream
    n_coll = sum(-(-(hi - lo) // sync.bucket_elems) for lo, hi in fired) + 1
    assert 6 <= n_coll <= 16, n_coll
    # the real _launch cuts the same pieces
    sync2 = FlatGrads(net)
    sync2.flat = torch.zeros(sync2.total)
    sync2.active, sync2.native_avg = True, False
    import torch.distributed as dist
    calls = []
    orig = dist.all_reduce
    dist.all_reduce = lambda buf, **kw: calls.append(buf.numel()) or None
    try:
        for layer in FlatGrads._completion_layers(net):
            sync2.ready(layer, None)
        sync2._launch()
    finally:
        dist.all_reduce = orig
    assert sum(calls) == sync2.total and max(calls) * 4 <= 30 << 20 and calls[-1] * 4 <= 2 << 20
    assert [b - a for a, b in sync2.launched] == calls
