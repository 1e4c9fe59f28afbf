"""One rank of tests/test_gpu_ddp_two_ranks.py (started as a child process; not collected by pytest).

    python tests/ddp_worker.py <rank> <world> <port> <out dir> [bf16]

Runs ONE training step (Trainer.step's sequence) of the real whole-net backward on this rank's own images -- the
gradient buckets go out over `gloo` while the backward is still running (FlatGrads._launch stages each bucket through
the host for gloo; RCCL reduces the device buffer in place) -- and writes the flat gradient buffer, the loss and two
updated parameters."""
import os
import sys

import numpy as np
import torch
import torch.distributed as dist

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
for p in (ROOT, os.path.join(ROOT, 'pytorch-unet_amd'), HERE):
    sys.path.insert(0, p)

from oracle import weights as W  # noqa: E402

SHAPE = (2, 3, 64, 80)


def closed_form(mod):
    sd = mod.state_dict()
    mod.load_state_dict({k: torch.from_numpy(W.tensor_for_key(k, tuple(v.shape), str(v.dtype))).to(v.dtype)
                         for k, v in sd.items()}, strict=True)
    return mod


def rank_batch(rank):
    img = torch.from_numpy(W.wave(f'ddp2.img.{rank}', SHAPE, 0.5, 0.5)).cuda().contiguous(memory_format=torch.channels_last)
    mask = torch.from_numpy(W.int_wave(f'ddp2.mask.{rank}', (SHAPE[0],) + SHAPE[2:], 2)).cuda()
    return img, mask


def run(rank, world, port, out, amp=False):
    from unet import UNet
    from unetk.trainer import Trainer
    print(f'rank {rank}/{world}: imports done', flush=True)
    if world > 1:
        dist.init_process_group('gloo', init_method=f'tcp://127.0.0.1:{port}', world_size=world, rank=rank)
    net = closed_form(UNet(3, 2, False)).cuda().to(memory_format=torch.channels_last).train()
    tr = Trainer(net, amp=amp, learning_rate=1e-3, gradient_clipping=1.0, bucket_bytes=4 << 20)
    assert tr.sync.world == world and tr.sync.active == (world > 1)
    launches = [0]
    orig = tr.sync._launch

    def counted():
        lo, hi = tr.sync._pending
        launches[0] += hi > lo
        orig()
    tr.sync._launch = counted
    img, mask = rank_batch(rank)
    print(f'rank {rank}: group up, stepping', flush=True)
    # Trainer.step's body (unetk/trainer.py), with a copy of the flat buffer taken between the all-reduce wait and the
    # optimizer (whose fused kernel scales the gradients by the clip coefficient in place)
    from unetk.loss import segmentation_loss
    with torch.autocast('cuda', dtype=torch.bfloat16, enabled=amp):
        loss = segmentation_loss(net(img), mask)
    tr.optimizer.zero_grad(set_to_none=True)
    loss.backward()
    tr.sync.finish()
    flat = tr.sync.flat.clone()
    tr.optimizer.step()
    loss = loss.detach()
    torch.cuda.synchronize()
    print(f'rank {rank}: step done, {launches[0]} buckets', flush=True)
    np.savez(os.path.join(out, f'rank{rank}.npz'), flat=flat.cpu().numpy(), loss=float(loss),
             buckets=launches[0], total_norm=float(tr.optimizer.total_norm),
             outc=net.outc.conv.weight.detach().cpu().numpy(), inc=net.inc.double_conv[0].weight.detach().cpu().numpy())
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == '__main__':
    import faulthandler
    faulthandler.dump_traceback_later(90, exit=True)          # a hung collective ends the rank with its stacks in the log
    run(int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3]), sys.argv[4], amp=len(sys.argv) > 5 and sys.argv[5] == 'bf16')
