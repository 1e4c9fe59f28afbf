"""Gradient bucketing for the data-parallel train step (the reference itself has no distributed code).

One process per GPU.  The whole-network autograd node (unet/unet_model.py) reports each LAYER's parameters (a
convolution and the BatchNorm behind it; a ConvTranspose2d; OutConv) as soon as their gradients are final, in reverse
execution order (outc, up4 ... up1, down4 ... inc; inside a DoubleConv the second layer first).  FlatGrads lays all
parameter gradients out in ONE fp32 buffer in exactly that order, so "the gradients finished so far" is always a
contiguous range.  That range is all-reduced (average) asynchronously on RCCL's stream, while the remaining backward
kernels keep the compute stream busy, whenever

  * >= bucket_bytes are pending (25 MB: large enough for xGMI's per-link-bound rings to reach their bandwidth), or
  * the pending range is at least as large as everything still to come (and >= min_bucket_bytes): the gradients
    shrink geometrically towards the encoder's first layers, so waiting for a full bucket there would leave tens of
    megabytes to a last, fully exposed collective after backward -- with this rule it is inc + down1, about 1 MB.

A range larger than bucket_bytes (the 1024 -> 1024 convolution alone is 37.7 MB) goes out as equal pieces of at most
bucket_bytes.  BatchNorm buffers are not touched: statistics stay per rank (north_star).
"""
import torch
import torch.distributed as dist


class FlatGrads:
    def __init__(self, net, process_group=None, bucket_bytes=25 << 20, force_collectives=False,
                 min_bucket_bytes=1 << 20, grads_to_autograd=True):
        self.module_order = [p for p in net.parameters()]     # torch.optim's order (checkpoints of a stock optimizer)
        order = []
        for layer in self._completion_layers(net):
            order += layer
        assert len(order) == len(net.hot_path_params()) and len(set(map(id, order))) == len(order)
        self.params = order
        self.offsets, off = {}, 0
        for p in order:
            self.offsets[id(p)] = off
            off += (p.numel() + 3) // 4 * 4                   # keep every view 16-byte aligned
        self.total = off
        self.group = process_group
        self.world = dist.get_world_size(process_group) if (dist.is_available() and dist.is_initialized()) else 1
        self.bucket_elems = max(bucket_bytes // 4, 1)
        self.min_bucket_elems = max(min(min_bucket_bytes, bucket_bytes) // 4, 1)
        self.launched = []            # (lo, hi) element ranges of the collectives of the last backward pass, in order
        # force_collectives: run the bucketed all-reduces even in a 1-rank group (rehearses the RCCL path -- stream
        # hand-over, bucket ranges -- on a single GPU; the average over one rank is the identity)
        self.active = self.world > 1 or (force_collectives and dist.is_available() and dist.is_initialized())
        # RCCL averages in the collective; gloo (CPU tests) has no AVG, so sum and divide after the wait
        self.native_avg = self.active and dist.get_backend(process_group) == 'nccl'
        self.flat = None
        self.views = {}
        self._pending = [0, 0]
        self._works = []
        self._unconsumed = False      # a backward pass filled `flat` and no optimizer step / zero_grad has taken it yet
        net._grad_ready_hook = self.ready
        net._initial_grads = self.initial_grads
        # grads_to_autograd=False: the whole-net backward hands NO parameter gradients back to autograd -- they live in
        # `flat` (what the fused optimizer steps from; bind_param_grads() makes them `p.grad` on request).  Handing the
        # flat views back makes AccumulateGrad CLONE every one of them (a view is never "stolen"): 64 small device
        # copies per step, 0.37 ms at 959x640, of values that are neither averaged over ranks nor clipped yet.
        net._grads_via_autograd = bool(grads_to_autograd)

    @staticmethod
    def _completion_layers(net):
        """Parameter groups in the order the whole-net backward reports them (blocks.double_conv_bwd / up_bwd)."""
        from . import blocks
        out = [[net.outc.conv.weight, net.outc.conv.bias]]
        for u in (net.up4, net.up3, net.up2, net.up1):
            out += blocks.up_layers(u)
        for d in (net.down4, net.down3, net.down2, net.down1):
            out += blocks.double_conv_layers(d.maxpool_conv[1])
        out += blocks.double_conv_layers(net.inc)
        return [g for g in out if g]

    @staticmethod
    def _completion_blocks(net):
        """The same order at block granularity (outc, up4 ... up1, down4 ... inc)."""
        from . import blocks
        out = [[net.outc.conv.weight, net.outc.conv.bias]]
        for u in (net.up4, net.up3, net.up2, net.up1):
            out.append([p for g in blocks.up_layers(u) for p in g])
        for d in (net.down4, net.down3, net.down2, net.down1):
            out.append([p for g in blocks.double_conv_layers(d.maxpool_conv[1]) for p in g])
        out.append([p for g in blocks.double_conv_layers(net.inc) for p in g])
        return out

    def initial_grads(self):
        """Fresh flat buffer + per-parameter views for one backward pass."""
        if self._unconsumed:
            # every backward pass WRITES a fresh flat buffer (the reference zeroes its gradients before each backward:
            # train.py:113); a second backward would silently drop the first one's gradients instead of adding to them
            raise RuntimeError('gradient accumulation is not supported on the flat gradient buffer: call '
                               'optimizer.step() or optimizer.zero_grad() between backward passes')
        dev = self.params[0].device
        self.flat = torch.zeros(self.total, dtype=torch.float32, device=dev)   # zero: alignment gaps enter the norm
        self._unconsumed = True
        self._pending = [0, 0]
        self._works = []
        self.launched = []
        self.views = {p: self.flat[self.offsets[id(p)]:self.offsets[id(p)] + p.numel()].view(p.shape)
                      for p in self.params}
        return dict(self.views)

    def consume(self):
        """The gradients of the last backward pass were applied (optimizer.step) or discarded (zero_grad)."""
        self._unconsumed = False

    def bind_param_grads(self):
        """Make every `p.grad` the parameter's view of the flat buffer.  Autograd only *sometimes* keeps the tensor a
        Function returns as `.grad` (AccumulateGrad clones it when its strides differ from the parameter's, e.g. after
        `net.to(memory_format=channels_last)`), so anything done to the flat buffer after backward -- the all-reduce
        average, the clip -- reaches a stock optimizer only through this call."""
        if self.flat is None:
            return
        for p in self.params:
            if p.requires_grad:
                p.grad = self.views[p]

    def ready(self, params, grads):
        if not params:
            return
        lo = min(self.offsets[id(p)] for p in params)
        hi = max(self.offsets[id(p)] + (p.numel() + 3) // 4 * 4 for p in params)
        assert lo == self._pending[1], 'layers must complete in layout order'
        self._pending[1] = hi
        pending = hi - self._pending[0]
        if self.active and (pending >= self.bucket_elems or
                            pending >= max(self.total - hi, self.min_bucket_elems)):
            self._launch()

    def _launch(self):
        lo, hi = self._pending
        if hi > lo:
            op = dist.ReduceOp.AVG if self.native_avg else dist.ReduceOp.SUM
            pieces = -(-(hi - lo) // self.bucket_elems)       # equal pieces of at most bucket_elems, 16-byte aligned
            step = -(-(hi - lo) // pieces)
            step = (step + 3) // 4 * 4
            for a in range(lo, hi, step):
                b = min(a + step, hi)
                buf = self.flat[a:b]
                if not self.native_avg and buf.is_cuda:
                    # gloo (rehearsals without RCCL, e.g. two ranks sharing one GPU): several asynchronous gloo
                    # collectives on device tensors interleaved with queued kernels hang in Work.wait() on this stack
                    # (plain-torch probe: profiles/r03_gloo_cuda_probe.log), so the bucket is staged through the host
                    # -- same bucket ranges, same launch points inside the backward, same wait-before-clip; only the
                    # overlap is lost
                    buf = buf.cpu()
                self._works.append((dist.all_reduce(buf, op=op, group=self.group, async_op=True), a, b, buf))
                self.launched.append((a, b))
        self._pending = [hi, hi]

    def finish(self):
        """Call after backward(): flush the last bucket and make the compute stream wait for every all-reduce."""
        if self.active:
            self._launch()
            for w, lo, hi, buf in self._works:
                w.wait()
                if buf.device != self.flat.device:
                    self.flat[lo:hi].copy_(buf)
                if not self.native_avg:
                    self.flat[lo:hi].div_(self.world)
        self._works = []
