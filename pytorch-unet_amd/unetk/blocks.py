"""Forward/backward composites of the U-Net blocks on top of the HIP kernels (unetk.ops).

Everything here works on NHWC views and explicit `saved` tuples; the nn.Modules in `unet/` wrap these in
`torch.autograd.Function`s (per block for stand-alone use, one Function for the whole UNet so that the
skip-connection buffers, the gradient accumulation of the skips and the gradient bucketing are under our control).

Reference semantics restated (file:line into the reference):
  double_conv_*   unet/unet_parts.py:15-22   conv3x3 -> BN -> ReLU, twice
  pool            unet/unet_parts.py:34      MaxPool2d(2)
  up_*            unet/unet_parts.py:49-67   Upsample|ConvTranspose2d -> F.pad -> cat([skip, up])
  head_*          unet/unet_parts.py:74      Conv2d k=1 + bias
"""
import os

import torch

from . import ops


# set by the autograd glue while a forward runs that a backward will follow (grad mode itself is off inside
# torch.autograd.Function.forward): eval-mode BatchNorm layers then keep their pre-BN tensor instead of fusing it away
SAVING_FOR_BACKWARD = False

# inference: fold eval-mode BatchNorm + ReLU (+ MaxPool2d) into the conv epilogue.  Tests flip it to compare the fused
# kernels with the unfused chain (conv -> bn_relu_apply -> maxpool) on the same net.
FUSE_EVAL_EPILOGUE = True


def activation_dtype(x):
    """The autocast dtype under torch.autocast (the reference's --amp path, train.py:100: torch's default for 'cuda' is
    float16, BASELINE.json's configurations ask for bfloat16), otherwise the input's dtype."""
    if torch.is_autocast_enabled():
        dt = torch.get_autocast_dtype('cuda')
        if dt not in (torch.bfloat16, torch.float16):
            raise RuntimeError(f'the MI355X kernels implement bf16, fp16 and fp32; autocast dtype {dt} is not supported')
        return dt
    if x.dtype in (torch.float32, torch.bfloat16, torch.float16):
        return x.dtype
    raise RuntimeError(f'unsupported activation dtype {x.dtype}')


def packed_weight(mod, kind, dtype, c_out, c_in):
    """Cached MFMA-fragment packing of mod.weight, refreshed when the parameter changes (optimizer step,
    load_state_dict, .to())."""
    w = mod.weight
    cache = mod.__dict__.setdefault('_unetk_pack', {})
    key = (kind, dtype)
    hit = cache.get(key)
    if hit is not None and hit[0] == w._version and hit[1] == w.data_ptr():
        return hit[2]
    t = ops.pack_weight(kind, w, c_out, c_in, dtype)
    cache[key] = (w._version, w.data_ptr(), t)
    return t


def pack_all(net, dtype, need_dgrad):
    """Refresh the packed weights of EVERY MFMA convolution of `net` in one launch (unetk_pack_plan_*), then let
    packed_weight() hit its cache: 42 per-layer pack launches per training step become one.  The plan (device copy
    of the per-layer records) is cached on the net and rebuilt when a parameter or packed buffer moved; layers whose
    weights are not plain contiguous fp32 keep the per-layer path."""
    import ctypes
    import numpy as np
    import torch.nn as nn
    targets = []
    for m in net.modules():
        if isinstance(m, nn.Conv2d) and m.kernel_size == (3, 3) and m.in_channels > 4:
            kinds, c_out, c_in = (0, 1), m.out_channels, m.in_channels
        elif isinstance(m, nn.ConvTranspose2d):
            kinds, c_out, c_in = (2, 3), m.out_channels, m.in_channels
        else:
            continue
        w = m.weight
        if w.dtype != torch.float32 or not w.is_contiguous() or not w.is_cuda:
            continue
        for kind in kinds if need_dgrad else kinds[:1]:
            targets.append((m, kind, c_out, c_in))
    if not targets:
        return

    def stale(t):
        m, kind = t[0], t[1]
        hit = m.__dict__.get('_unetk_pack', {}).get((kind, dtype))
        return hit is None or hit[0] != m.weight._version or hit[1] != m.weight.data_ptr()
    if not any(stale(t) for t in targets):
        return
    dc = ops.dtype_code(dtype)
    plans = net.__dict__.setdefault('_unetk_pack_plans', {})
    plan = plans.get((dtype, need_dgrad))
    srcs = tuple(t[0].weight.data_ptr() for t in targets)
    if plan is None or plan['srcs'] != srcs:
        es = torch.empty(0, dtype=dtype).element_size()
        bufs = []
        for m, kind, c_out, c_in in targets:
            nbytes = ops.fn['unetk_packed_weight_bytes'](kind, c_out, c_in, dc)
            if nbytes < 0:
                ops.check(int(nbytes), 'unetk_packed_weight_bytes')
            bufs.append(torch.empty(nbytes // es, dtype=dtype, device=m.weight.device))

        class Item(ctypes.Structure):
            _fields_ = [('src', ctypes.c_void_p), ('dst', ctypes.c_void_p), ('kind', ctypes.c_int),
                        ('c_out', ctypes.c_int), ('c_in', ctypes.c_int), ('dtype', ctypes.c_int)]
        items = (Item * len(targets))(*[Item(t[0].weight.data_ptr(), b.data_ptr(), t[1], t[2], t[3], dc)
                                        for t, b in zip(targets, bufs)])
        nbytes = int(ops.fn['unetk_pack_plan_bytes'](len(targets)))
        host = np.zeros(nbytes, dtype=np.uint8)
        total = int(ops.fn['unetk_pack_plan_build'](ctypes.addressof(items), len(targets), host.ctypes.data, nbytes))
        if total <= 0:
            ops.check(total, 'unetk_pack_plan_build')
        plan = {'srcs': srcs, 'bufs': bufs, 'dev': torch.from_numpy(host).to(targets[0][0].weight.device),
                'total': total, 'n': len(targets)}
        plans[(dtype, need_dgrad)] = plan
    ops.check(ops.fn['unetk_pack_plan_run'](plan['dev'].data_ptr(), plan['n'], plan['total'], ops.stream_ptr()),
              'unetk_pack_plan_run')
    for (m, kind, _, _), buf in zip(targets, plan['bufs']):
        m.__dict__.setdefault('_unetk_pack', {})[(kind, dtype)] = (m.weight._version, m.weight.data_ptr(), buf)


def _f32c(p):
    t = p.detach()
    return t if (t.dtype == torch.float32 and t.is_contiguous()) else t.float().contiguous()


def _bn_uses_batch_stats(bn):
    return bn.training or bn.running_mean is None


def _check_bn(bn):
    if bn.running_mean is not None and bn.running_mean.dtype != torch.float32:
        raise RuntimeError('BatchNorm buffers must be fp32')
    if bn.weight is not None and bn.weight.dtype != torch.float32:
        raise RuntimeError('BatchNorm parameters must be fp32 (keep the model in fp32 and use autocast for bf16)')


def conv_bn_relu_fwd(conv, bn, x, dtype, out=None, pool_out=None, head=None):
    """One conv3x3 -> BN -> ReLU.  x: NHWC view (fp32 image for the first layer, else `dtype`).
    Returns (activation, saved) with saved = (x, y, state).
    head: dict {'conv': OutConv's Conv2d} when OutConv consumes this layer's output; where the fused kernel applies
    (training-mode BatchNorm, <= 4 classes) the dict receives 'logits' and 'saved' (what head_fwd would return)."""
    c_in, c_out = conv.in_channels, conv.out_channels
    _check_bn(bn)
    batch_stats = _bn_uses_batch_stats(bn)
    n, h, w, _, _ = ops.nhwc_meta(x, 'conv input')
    stats = ops.new_stats(c_out, x.device) if batch_stats else None
    if not batch_stats and FUSE_EVAL_EPILOGUE and not SAVING_FOR_BACKWARD and (c_in <= 4 or dtype != torch.float32):
        # inference (evaluate.py:10, predict.py:20 call net.eval()): BatchNorm with running statistics is a per-channel
        # affine map -> folded with the ReLU into the conv epilogue, no pre-BN tensor, no second pass
        st = ops.BNState(c_out, x.device)
        ops.bn_eval_params(bn, st, float(bn.eps))
        if c_in <= 4:
            x = x if x.dtype == torch.float32 else x.float()
            act = ops.conv3x3_first_fprop_bn_relu(x, _f32c(conv.weight), c_out, dtype, st, out=out)
        else:
            if x.dtype != dtype:
                x = x.to(dtype)
            act = ops.conv3x3_fprop_bn_relu(x, packed_weight(conv, 0, dtype, c_out, c_in), c_out, st, out=out,
                                            pool_out=pool_out)
            pool_out = None                               # came out of the same epilogue
        if pool_out is not None:
            ops.maxpool_into(act, pool_out)
        return act, (x, None, st)
    # the first layer in training: statistics pass, finalize, then the layer again with BatchNorm + ReLU in its epilogue
    # (its 64-channel output is written once and not read back for an apply pass)
    first_fused = c_in <= 4 and FUSE_FIRST_FWD and pool_out is None and head is None
    if c_in <= 4:
        xin = x if x.dtype == torch.float32 else x.float()
        if not first_fused:
            y = ops.conv3x3_first_fprop(xin, _f32c(conv.weight), c_out, dtype, stats=stats)
        elif batch_stats:
            ops.conv3x3_first_stats(xin, _f32c(conv.weight), c_out, dtype, stats)
        x = xin
    else:
        if x.dtype != dtype:
            x = x.to(dtype)
        y = ops.conv3x3_fprop(x, packed_weight(conv, 0, dtype, c_out, c_in), c_out, stats=stats)
    st = ops.BNState(c_out, x.device)
    if batch_stats:
        momentum = bn.momentum
        update = bn.training and bn.running_mean is not None
        if update and momentum is None:      # cumulative moving average (torch semantics)
            momentum = 1.0 / float(int(bn.num_batches_tracked) + 1)

        class _B:          # what the finalize kernel may touch
            weight, bias = bn.weight, bn.bias
            running_mean = bn.running_mean if update else None
            running_var = bn.running_var if update else None
            num_batches_tracked = bn.num_batches_tracked if update else None
        ops.bn_train_finalize(stats, n * h * w, _B, st, float(momentum or 0.0), float(bn.eps))
    else:
        ops.bn_eval_params(bn, st, float(bn.eps), for_backward=SAVING_FOR_BACKWARD)
    if first_fused:
        y, act = ops.conv3x3_first_fprop_bn_relu_train(x, _f32c(conv.weight), c_out, dtype, st, out=out)
        return act, (x, y, st)
    if head is not None and pool_out is None and ops.head_fwd_fusable(c_out, head['conv'].out_channels, dtype) \
            and head['conv'].kernel_size == (1, 1):
        hc = head['conv']
        wh = _f32c(hc.weight)
        # head['logits_only'] (the whole-net forward): the activation itself has no reader -- OutConv's gradients are
        # re-formed from y in the BatchNorm backward (head_bwd_fusable), or there is no backward -- so it is not stored
        skip = bool(head.get('logits_only')) and out is None and ops.head_bwd_fusable(c_out, hc.out_channels, dtype) \
            and not HEAD_STORE_ACT
        act, logits = ops.bn_relu_apply_head(y, st, wh.reshape(hc.out_channels, c_out), _f32c(hc.bias), out=out,
                                             store_act=not skip)
        head['logits'], head['saved'] = logits, (act, wh, dtype)
        return act, (x, y, st)
    act = ops.bn_relu_apply(y, st, out=out, pool_out=pool_out)
    return act, (x, y, st)


# Set by the whole-net backward (unet/unet_model.py) while it runs: conv_bn_relu_bwd launches dgrad first and postpones the
# layer's wgrad until the next layer's BatchNorm-backward reduction is queued, so that MFMA-bound and HBM-bound kernels
# alternate (dgrad, reduce, wgrad, apply, dgrad ...) instead of running in pairs: +0.6 % images/sec in a same-box A/B
# (UNETK_DEFER_WGRAD=0 restores wgrad-before-dgrad).  Stand-alone block Functions keep the plain order.
DEFER_WGRAD = False
DEFER_WGRAD_ENABLED = os.environ.get('UNETK_DEFER_WGRAD', '1') != '0'

# UNETK_HEAD_STORE_ACT=1 (A/B runs): the whole-net forward stores the last activation although nothing reads it
HEAD_STORE_ACT = os.environ.get('UNETK_HEAD_STORE_ACT', '0') != '0'

# UNETK_FUSE_FIRST_FWD=0 (A/B runs): the first layer's forward stays conv (+ statistics), finalize, apply pass
FUSE_FIRST_FWD = os.environ.get('UNETK_FUSE_FIRST_FWD', '1') != '0'

# UNETK_FUSE_FIRST_WGRAD=0 (A/B runs): the first layer's BatchNorm-backward apply pass and weight gradient stay two kernels
FUSE_FIRST_WGRAD = os.environ.get('UNETK_FUSE_FIRST_WGRAD', '1') != '0'

# fold the first layer's BN-backward reduction into the second layer's dgrad epilogue (UNETK_FUSE_BN_REDUCE=0: A/B runs)
FUSE_BN_REDUCE = os.environ.get('UNETK_FUSE_BN_REDUCE', '1') != '0'


def conv_bn_relu_bwd(conv, bn, saved, dact, grads, need_dx, dx_out=None, dpool=None, dhead=None, dx_sums=None,
                     bn_sums=None, bn_prev=None, done=None):
    """Backward of conv_bn_relu_fwd.  grads: dict param -> fp32 tensor to fill (allocated here if missing).
    dpool: gradient of the MaxPool2d(2) that also consumed this layer's output (`dact` is then the skip gradient).
    dhead: (dlogits, OutConv weight) when OutConv consumed this layer's output (`dact` is then None).
    dx_sums: zeroed [REPLICAS, c_in, 2] accumulator for the per-channel sums of the input gradient.
    bn_sums: this layer's BN-backward sums, already formed by the dgrad that wrote `dact`.
    bn_prev: (z, BNState) of the layer that produced this layer's input: returns (dx, that layer's bn_sums).
    done: callback(params) invoked once this layer's parameter gradients are queued (gradient bucketing, unetk.ddp)."""
    x, y, st = saved
    c_in, c_out = conv.in_channels, conv.out_channels
    if y is None:
        raise RuntimeError('this forward ran outside autograd (fused inference epilogue): nothing was saved for backward')
    frozen = not _bn_uses_batch_stats(bn)       # eval-mode BatchNorm: running statistics are constants of the graph
    dgamma = _grad_buf(grads, bn.weight) if bn.weight is not None else None
    dbeta = _grad_buf(grads, bn.bias) if bn.bias is not None else None
    dw = _grad_buf(grads, conv.weight)
    if c_in <= 4 and not need_dx and FUSE_FIRST_WGRAD and dpool is None and dhead is None \
            and ops.first_wgrad_bn_fusable(y.dtype):
        # the training loop's first layer: nothing reads the conv-output gradient but the weight-gradient kernel, which
        # forms it on the fly (one full-resolution tensor neither written nor read back)
        ops.bn_relu_bwd_first_wgrad(dact, y, st, bn.weight, dgamma, dbeta, x, dw, sums=bn_sums, frozen=frozen)
        if done is not None:
            done(conv_bn_params(conv, bn))
        return None
    dy = ops.bn_relu_bwd(dact, y, st, bn.weight, dgamma, dbeta, dpool=dpool, dhead=dhead, sums=bn_sums, frozen=frozen)
    if c_in <= 4:
        ops.conv3x3_first_wgrad(x, dy, dw)
        if done is not None:
            done(conv_bn_params(conv, bn))
        if need_dx:
            return ops.conv3x3_first_dgrad(dy, _f32c(conv.weight), c_in)
        return None
    def wgrad():
        ops.conv3x3_wgrad(x, dy, dw)
        if done is not None:
            done(conv_bn_params(conv, bn))
    if DEFER_WGRAD and need_dx:
        # whole-net backward: dgrad first, and the weight gradient behind the NEXT layer's BatchNorm-backward reduction
        # (ops.bn_relu_bwd runs the postponed launch between its two passes) -- MFMA-bound and HBM-bound kernels then
        # alternate instead of running in pairs
        dx = ops.conv3x3_dgrad(dy, packed_weight(conv, 1, dy.dtype, c_out, c_in), c_in, out=dx_out, sums=dx_sums,
                               bn_prev=bn_prev)
        ops.defer(wgrad)
        return dx
    wgrad()                                  # before the dgrad launch: the collective overlaps it
    if not need_dx:
        return None
    return ops.conv3x3_dgrad(dy, packed_weight(conv, 1, dy.dtype, c_out, c_in), c_in, out=dx_out, sums=dx_sums,
                             bn_prev=bn_prev)


def _grad_buf(grads, p):
    g = grads.get(p)
    if g is None:
        g = torch.empty(p.shape, dtype=torch.float32, device=p.device)
        grads[p] = g
    return g


# ------------------------------------------------------------------------------------ DoubleConv
def double_conv_fwd(dc, x, dtype, out=None, pool_out=None, head=None):
    """pool_out: optional [N,H//2,W//2,C] buffer that receives MaxPool2d(2) of the block's output (fused).
    head: see conv_bn_relu_fwd (OutConv's logits emitted by the block's last kernel)."""
    seq = dc.double_conv
    a1, s1 = conv_bn_relu_fwd(seq[0], seq[1], x, dtype)
    a2, s2 = conv_bn_relu_fwd(seq[3], seq[4], a1, dtype, out=out, pool_out=pool_out, head=head)
    return a2, (s1, s2)


def double_conv_bwd(dc, saved, da2, grads, need_dx, dx_out=None, dpool=None, dhead=None, dx_sums=None, done=None,
                    done_first=None, sums2=None):
    """done: per-layer completion callback (see conv_bn_relu_bwd); done_first: parameters that complete together with
    the block's second layer (OutConv's, when its gradient comes out of this block's BatchNorm backward: `dhead`).
    sums2: BN-backward sums of the block's SECOND layer, already formed by the kernel that wrote `da2` (up_bwd)."""
    seq = dc.double_conv
    s1, s2 = saved
    c_mid = seq[3].in_channels
    done2 = done
    if done is not None and done_first:
        def done2(params):
            done(done_first)
            done(params)
    if FUSE_BN_REDUCE and ops.dgrad_bn_reduce_is_fused(c_mid, seq[3].out_channels, s1[1].dtype):
        # s1 = (x, z1, state1): the second conv consumed relu(bn(z1)), so its dgrad epilogue can reduce for BN 1
        da1, sums1 = conv_bn_relu_bwd(seq[3], seq[4], s2, da2, grads, True, dpool=dpool, dhead=dhead,
                                      bn_prev=(s1[1], s1[2]), done=done2, bn_sums=sums2)
    else:
        da1, sums1 = conv_bn_relu_bwd(seq[3], seq[4], s2, da2, grads, True, dpool=dpool, dhead=dhead, done=done2,
                                      bn_sums=sums2), None
    return conv_bn_relu_bwd(seq[0], seq[1], s1, da1, grads, need_dx, dx_out=dx_out, dx_sums=dx_sums, bn_sums=sums1,
                            done=done)


def conv_bn_params(conv, bn):
    return [p for p in (conv.weight, bn.weight, bn.bias) if p is not None]


def double_conv_params(dc):
    seq = dc.double_conv
    ps = [seq[0].weight, seq[1].weight, seq[1].bias, seq[3].weight, seq[4].weight, seq[4].bias]
    return [p for p in ps if p is not None]


def double_conv_layers(dc):
    """Parameter groups in the order double_conv_bwd completes them: second layer, then first."""
    seq = dc.double_conv
    return [conv_bn_params(seq[3], seq[4]), conv_bn_params(seq[0], seq[1])]


# ------------------------------------------------------------------------------------ Up (both flavours)
def pad_offsets(h, w, hs, ws):
    """F.pad(x1, [dX//2, dX-dX//2, dY//2, dY-dY//2]) (unet_parts.py:59-63): top/left offsets of the window."""
    dy, dx = hs - 2 * h, ws - 2 * w
    if dy < 0 or dx < 0:
        raise RuntimeError('Up: the skip tensor is smaller than the up-sampled one (negative padding is not supported)')
    return dy // 2, dx // 2


def up_into_cat(up_mod, x_low, cat, c_skip, dtype):
    """Writes up(x_low), padded, into cat[..., c_skip:].  Returns saved info for backward."""
    n, h, w, c_low, _ = ops.nhwc_meta(x_low, 'Up low-res input')
    _, hs, ws, ctot, _ = ops.nhwc_meta(cat, 'Up concat buffer')
    pad_t, pad_l = pad_offsets(h, w, hs, ws)
    if x_low.dtype != dtype:
        x_low = x_low.to(dtype)
    if isinstance(up_mod, torch.nn.ConvTranspose2d):
        c_out = up_mod.out_channels
        if c_skip + c_out != ctot:
            raise RuntimeError('Up: channel counts of skip and up-sampled tensors do not add up')
        ops.convT2x2_fwd(x_low, packed_weight(up_mod, 2, dtype, c_out, c_low), _f32c(up_mod.bias), cat, c_skip, c_out,
                         pad_t, pad_l)
    else:
        if c_skip + c_low != ctot:
            raise RuntimeError('Up: channel counts of skip and up-sampled tensors do not add up')
        ops.bilinear_fwd(x_low, cat, c_skip, pad_t, pad_l)
    return (x_low, h, w, c_low, pad_t, pad_l)


# fold the BN-backward reduction of the layer below an Up block into the ConvTranspose2d dgrad epilogue
# (UNETK_FUSE_BN_REDUCE_UP=0: A/B runs)
FUSE_BN_REDUCE_UP = os.environ.get('UNETK_FUSE_BN_REDUCE_UP', '1') != '0'


def up_bwd(up_mod, saved, dcat, c_skip, grads, need_dx=True, dcat_sums=None, done=None, bn_prev=None):
    """dcat_sums: per-channel sums of dcat from the dgrad that wrote it (ConvTranspose2d bias gradient for free).
    bn_prev: (z, BNState) of the conv -> BN -> ReLU layer that produced the block's low-res input: where the
    ConvTranspose2d dgrad kernel can form that layer's BN-backward sums the result is (dx, bn_sums), else (dx, None)."""
    x_low, h, w, c_low, pad_t, pad_l = saved
    ops.run_deferred()                       # parameter groups complete in layout order (unetk.ddp)
    if isinstance(up_mod, torch.nn.ConvTranspose2d):
        c_out = up_mod.out_channels
        ops.convT2x2_wgrad(x_low, dcat, c_skip, _grad_buf(grads, up_mod.weight), _grad_buf(grads, up_mod.bias), pad_t, pad_l,
                           sums=dcat_sums)
        if done is not None:
            done([up_mod.weight, up_mod.bias])
        if not need_dx:
            return None if bn_prev is None else (None, None)
        if bn_prev is not None and FUSE_BN_REDUCE_UP and ops.convT_dgrad_bn_reduce_is_fused(c_low, c_out, dcat.dtype):
            return ops.convT2x2_dgrad(dcat, c_skip, packed_weight(up_mod, 3, dcat.dtype, c_out, c_low), c_low, c_out,
                                      h, w, pad_t, pad_l, bn_prev=bn_prev)
        dx = ops.convT2x2_dgrad(dcat, c_skip, packed_weight(up_mod, 3, dcat.dtype, c_out, c_low), c_low, c_out, h, w,
                                pad_t, pad_l)
        return dx if bn_prev is None else (dx, None)
    if not need_dx:
        return None if bn_prev is None else (None, None)
    dx = ops.bilinear_bwd(dcat, c_skip, c_low, h, w, pad_t, pad_l)
    return dx if bn_prev is None else (dx, None)


def up_params(up_block):
    ps = []
    if isinstance(up_block.up, torch.nn.ConvTranspose2d):
        ps += [up_block.up.weight, up_block.up.bias]
    return ps + double_conv_params(up_block.conv)


def up_layers(up_block):
    """Parameter groups of an Up block in completion order: its DoubleConv's two layers, then the ConvTranspose2d."""
    out = double_conv_layers(up_block.conv)
    if isinstance(up_block.up, torch.nn.ConvTranspose2d):
        out.append([up_block.up.weight, up_block.up.bias])
    return out


# ------------------------------------------------------------------------------------ head
def head_fwd(conv, x, dtype):
    if conv.kernel_size != (1, 1):
        raise RuntimeError('OutConv expects a 1x1 convolution')
    if x.dtype != dtype:
        x = x.to(dtype)
    w = _f32c(conv.weight)
    return ops.conv1x1_fwd(x, w, _f32c(conv.bias), conv.out_channels), (x, w)


def head_bwd(conv, saved, dl, grads, need_dx=True, fuse_dx=False):
    """Returns (input gradient, None), or with fuse_dx -- when the consumer's BatchNorm backward can form the input
    gradient and OutConv's own weight/bias gradient on the fly (ops.head_bwd_fusable) -- (None, dhead): nothing is
    launched here and the caller passes `dhead` on to double_conv_bwd of the block that fed OutConv."""
    x, w = saved[0], saved[1]
    dtype = saved[2] if len(saved) > 2 else x.dtype      # (x is None when the fused forward stored no activation)
    if dl.dtype != dtype:
        dl = dl.to(dtype)
    n_classes, c = conv.out_channels, conv.in_channels
    dw, db = _grad_buf(grads, conv.weight), _grad_buf(grads, conv.bias)
    if need_dx and fuse_dx and ops.head_bwd_fusable(c, n_classes, dtype):
        return None, (dl, w.reshape(n_classes, c), dw, db)
    if x is None:
        raise RuntimeError('OutConv backward: the forward kept no activation and the fused backward does not apply')
    return ops.conv1x1_bwd(x, dl, w, dw, db, need_dx), None
