"""The RCCL leg of the data-parallel step on ONE GPU: a 1-rank `nccl` process group with the collectives forced on.
What this can and cannot show: the average over one rank is the identity written in place, so the test proves that
the bucketed asynchronous path RUNS on RCCL (ReduceOp.AVG on fp32 ranges of the flat gradient buffer, work handles,
>= 5 buckets per step, hand-over between the stream the C ABI launches on and RCCL's) and leaves the training
trajectory where the local trainer's is -- to the run-to-run noise of two LOCAL trainers (fp64 BatchNorm atomics
-> last-bit changes -> RMSprop's sign-like normalisation; measured 1.5e-5 on the loss after one step): loss within
1e-3 relative, parameters within the measured local-vs-local noise.
It cannot expose a missing stream dependency (a 1-rank all-reduce rewrites what it read).  World sizes > 1 are
covered on CPU by tests/test_ddp_gloo.py; the 8-GPU run is the driver's."""
import pytest
import torch

pytestmark = pytest.mark.gpu


def test_single_rank_rccl_step_equals_local_step():
    import torch.distributed as dist
    from unet import UNet
    from unetk.trainer import Trainer
    dev = torch.device('cuda', 0)
    dist.init_process_group('nccl', init_method='tcp://127.0.0.1:29541', world_size=1, rank=0, device_id=dev)
    try:
        torch.manual_seed(0)
        a = UNet(3, 2, False).to(dev).to(memory_format=torch.channels_last).train()
        b = UNet(3, 2, False).to(dev).to(memory_format=torch.channels_last).train()
        b.load_state_dict(a.state_dict())
        c = UNet(3, 2, False).to(dev).to(memory_format=torch.channels_last).train()
        c.load_state_dict(a.state_dict())                      # a second LOCAL trainer: the noise yardstick
        p0 = {k: v.detach().clone() for k, v in a.named_parameters()}
        ta = Trainer(a, amp=True)
        tc = Trainer(c, amp=True)
        tb = Trainer(b, amp=True, bucket_bytes=4 << 20, force_collectives=True)
        assert tb.sync.active and tb.sync.native_avg and not ta.sync.active
        g = torch.Generator(device=dev).manual_seed(3)
        img = torch.rand((2, 3, 96, 143), device=dev, generator=g).contiguous(memory_format=torch.channels_last)
        mask = (torch.rand((2, 96, 143), device=dev, generator=g) < 0.2).long()
        n_buckets = 0
        for _ in range(3):
            la = ta.step(img, mask)
            tc.step(img, mask)
            orig = tb.sync._launch
            count = [0]

            def counted():
                lo, hi = tb.sync._pending
                count[0] += hi > lo
                orig()
            tb.sync._launch = counted
            lb = tb.step(img, mask)
            tb.sync._launch = orig
            n_buckets = count[0]
            assert abs(float(la) - float(lb)) <= 1e-3 * abs(float(la))
        assert n_buckets >= 5                      # 124 MB of gradients in >= 4 MB buckets
        # three RMSprop steps move every element by ~10 lr per step (sign-like normalisation).  Two LOCAL trainers
        # (a, c) differ where a gradient is at the rounding noise of the fp64-atomic orderings and steps the other
        # way -- measured here, not assumed.  The RCCL trainer must stay within that noise (x1.5 + 1 % of the
        # movement); a missing or doubled bucket average would show as a difference of the order of the movement.
        moved = diff_rccl = diff_local = 0.0
        for (k, pa), (_, pb), (_, pc) in zip(a.named_parameters(), b.named_parameters(), c.named_parameters()):
            moved += float((pa - p0[k]).abs().sum())
            diff_rccl += float((pa - pb).abs().sum())
            diff_local += float((pa - pc).abs().sum())
        assert moved > 0
        assert diff_rccl <= 1.5 * diff_local + 0.01 * moved, (diff_rccl, diff_local, moved)
        assert diff_rccl <= 0.5 * moved, (diff_rccl, moved)
    finally:
        dist.destroy_process_group()
