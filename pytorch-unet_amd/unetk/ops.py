"""Thin Python wrappers over the C ABI (include/unetk.h): shape checks, output allocation, launch.

Tensors here are NHWC *views*: shape [N,H,W,C], stride[-1] == 1, stride[-2] == ld >= C (a channel
slice of a wider buffer is fine), rows and images dense.  Modules convert from/to PyTorch's logical
NCHW with `.permute` -- on a channels_last tensor that is free.
"""
import torch

from . import _native as nat
from ._native import fn, check, dtype_code, stream_ptr, require_cuda


def nhwc_meta(t, what='tensor'):
    """(n, h, w, c, ld) of an NHWC view; strides of size-1 dimensions are don't-cares."""
    if t.dim() != 4:
        raise RuntimeError(f'{what}: expected a 4-D NHWC view, got shape {tuple(t.shape)}')
    n, h, w, c = t.shape
    sn, sh, sw, sc = t.stride()
    ok = c == 1 or sc == 1
    ld = sw if w > 1 else (sh if h > 1 else (sn if n > 1 else c))
    if w == 1 and h > 1:
        ld = sh          # one pixel per row: the row stride is the pixel pitch
    ok = ok and ld >= c and (h == 1 or w == 1 or sh == w * ld) and (n == 1 or sn == h * w * ld)
    if not ok:
        raise RuntimeError(f'{what}: expected a dense NHWC view with unit channel stride, got shape '
                           f'{tuple(t.shape)} strides {t.stride()}')
    return n, h, w, c, ld


def new_nhwc(n, h, w, c, dtype, device):
    return torch.empty((n, h, w, c), dtype=dtype, device=device)


def to_nhwc(x):
    """Logical NCHW (any strides) -> NHWC view.  Free for channels_last tensors and for channel slices of
    channels_last buffers; anything else is made channels_last first (a copy by PyTorch, outside the hot path)."""
    v = x.permute(0, 2, 3, 1)
    try:
        _, _, _, c, ld = nhwc_meta(v)
        es = x.element_size()
        if (ld * es) % 16 == 0 and x.data_ptr() % 16 == 0:
            return v
    except RuntimeError:
        pass
    return x.contiguous(memory_format=torch.channels_last).permute(0, 2, 3, 1)


def to_nchw(x_nhwc):
    """NHWC view -> logical NCHW tensor with channels_last strides (no copy)."""
    return x_nhwc.permute(0, 3, 1, 2)


# ------------------------------------------------------------------------------------ weights
def pack_weight(kind, w_param, c_out, c_in, dtype):
    """kind: 0 conv3x3 fprop, 1 conv3x3 dgrad, 2 convT fwd, 3 convT dgrad (unetk.h)."""
    require_cuda(w_param)
    src = w_param.detach()
    if src.dtype != torch.float32 or not src.is_contiguous():
        src = src.float().contiguous()
    dc = dtype_code(dtype)
    nbytes = fn['unetk_packed_weight_bytes'](kind, c_out, c_in, dc)
    if nbytes < 0:
        check(int(nbytes), 'unetk_packed_weight_bytes')
    out = torch.empty(nbytes // src.new_empty(0, dtype=dtype).element_size(), dtype=dtype, device=src.device)
    check(fn['unetk_pack_weight'](kind, src.data_ptr(), out.data_ptr(), c_out, c_in, dc, stream_ptr()), 'unetk_pack_weight')
    return out


# ------------------------------------------------------------------------------------ conv 3x3
class ZeroArena:
    """One zero-filled fp64 buffer handed out in slices: the 18 BatchNorm layers of a pass each need a zeroed
    [replicas][C][2] accumulator, and 18 tiny fill kernels cost more than one memset of the lot.  `begin()` re-zeroes
    the buffer (one launch); every slice is consumed (reduced + finalised) before the next `begin()` on that stream."""

    def __init__(self):
        self.buf, self.used = None, 0

    def begin(self, device, n_doubles):
        if self.buf is None or self.buf.numel() < n_doubles or self.buf.device != torch.device(device):
            self.buf = torch.zeros(n_doubles, dtype=torch.float64, device=device)
        else:
            self.buf.zero_()
        self.used = 0

    def take(self, n_doubles, device):
        if self.buf is None or self.buf.device != torch.device(device) or self.used + n_doubles > self.buf.numel():
            return None
        t = self.buf[self.used:self.used + n_doubles]
        self.used += n_doubles
        return t


class _ArenaMap:
    """One ZeroArena per (device, stream): reuse of a slice is ordered only by the stream that zeroed and consumed it,
    so two streams (or two GPUs driven from one process) must never share a buffer."""

    def __init__(self):
        self.arenas = {}

    def _current(self, device, create):
        dev = torch.device(device)
        if dev.type != 'cuda':
            return None
        idx = dev.index if dev.index is not None else torch.cuda.current_device()
        key = (idx, torch.cuda.current_stream(idx).cuda_stream)
        a = self.arenas.get(key)
        if a is None and create:
            a = self.arenas[key] = ZeroArena()
        return a

    def begin(self, device, n_doubles):
        self._current(device, True).begin(device, n_doubles)

    def take(self, n_doubles, device):
        a = self._current(device, False)
        return a.take(n_doubles, device) if a is not None else None


_ARENA = _ArenaMap()


def begin_stats_arena(device, total_channels):
    """Zero, in one launch, accumulators for BatchNorm layers with `total_channels` channels in all (whole-net
    forward and backward passes call this once each)."""
    _ARENA.begin(device, nat.STAT_REPLICAS * 2 * int(total_channels))


def new_stats(c, device):
    t = _ARENA.take(nat.STAT_REPLICAS * c * 2, device)
    if t is not None:
        return t.view(nat.STAT_REPLICAS, c, 2)
    return torch.zeros((nat.STAT_REPLICAS, c, 2), dtype=torch.float64, device=device)


def conv3x3_fprop(x, wpk, c_out, out=None, stats=None):
    require_cuda(x, wpk)
    n, h, w, c_in, ldx = nhwc_meta(x, 'conv3x3 input')
    if out is None:
        out = new_nhwc(n, h, w, c_out, x.dtype, x.device)
    on, oh, ow, oc, ldy = nhwc_meta(out, 'conv3x3 output')
    if (on, oh, ow, oc) != (n, h, w, c_out) or out.dtype != x.dtype or wpk.dtype != x.dtype:
        raise RuntimeError('conv3x3: output/weight do not match the input')
    if wpk.numel() != 9 * c_in * c_out:
        raise RuntimeError('conv3x3: packed weight has the wrong size')
    if stats is not None and (stats.dtype != torch.float64 or stats.numel() != nat.STAT_REPLICAS * c_out * 2):
        raise RuntimeError('conv3x3: stats buffer must be float64 [REPLICAS, C, 2]')
    check(fn['unetk_conv3x3_fprop'](x.data_ptr(), ldx, 0, wpk.data_ptr(), out.data_ptr(), ldy, 0,
                                    stats.data_ptr() if stats is not None else None,
                                    n, h, w, c_in, c_out, dtype_code(x.dtype), stream_ptr()), 'unetk_conv3x3_fprop')
    return out


def conv3x3_fprop_bn_relu(x, wpk, c_out, state, out=None, pool_out=None):
    """relu(scale * conv3x3(x) + shift) in one kernel (eval-mode BatchNorm folded into the conv epilogue; bf16);
    pool_out: optional [N,H//2,W//2,C] that receives MaxPool2d(2) of the result from the same epilogue."""
    require_cuda(x, wpk)
    n, h, w, c_in, ldx = nhwc_meta(x, 'conv3x3 input')
    if out is None:
        out = new_nhwc(n, h, w, c_out, x.dtype, x.device)
    on, oh, ow, oc, ldy = nhwc_meta(out, 'conv3x3 output')
    if (on, oh, ow, oc) != (n, h, w, c_out) or out.dtype != x.dtype or wpk.dtype != x.dtype:
        raise RuntimeError('conv3x3: output/weight do not match the input')
    if wpk.numel() != 9 * c_in * c_out or state.scale.numel() != c_out:
        raise RuntimeError('conv3x3: packed weight / BatchNorm coefficients have the wrong size')
    ldp = 0
    if pool_out is not None:
        pn, ph, pw, pc, ldp = nhwc_meta(pool_out, 'pool output')
        if (pn, ph, pw, pc) != (n, h // 2, w // 2, c_out) or pool_out.dtype != x.dtype:
            raise RuntimeError('conv3x3: pool_out must be [N, H//2, W//2, C] of the activation dtype')
    check(fn['unetk_conv3x3_fprop_bn_relu'](x.data_ptr(), ldx, 0, wpk.data_ptr(), state.scale.data_ptr(),
                                            state.shift.data_ptr(), out.data_ptr(), ldy, 0, _ptr(pool_out), ldp,
                                            n, h, w, c_in, c_out, dtype_code(x.dtype), stream_ptr()),
          'unetk_conv3x3_fprop_bn_relu')
    return out


def dgrad_bn_reduce_is_fused(c_in, c_out, dtype):
    """Does conv3x3_dgrad(bn_prev=...) fold the reduction into the dgrad kernel for this layer (else it would run the
    two kernels back to back inside one call: same result, nothing saved)?"""
    return bool(fn['unetk_conv3x3_dgrad_bn_reduce_is_fused'](int(c_in), int(c_out), dtype_code(dtype)))


def conv3x3_dgrad(dy, wpk_d, c_in, out=None, sums=None, bn_prev=None):
    """sums: optional zeroed fp64 [REPLICAS, c_in, 2] that receives the per-channel sums of the written gradient
    (the conv epilogue's fused statistics; used for the ConvTranspose2d bias gradient).
    bn_prev: (z, BNState) of the conv -> BN -> ReLU layer whose activation this convolution consumed: the epilogue
    also forms that layer's BN-backward sums (what unetk_bn_relu_bwd_reduce would compute from the returned gradient);
    returns (dx, bn_sums) then -- pass bn_sums to bn_relu_bwd(sums=...)."""
    require_cuda(dy, wpk_d)
    if bn_prev is not None:
        if sums is not None:
            raise RuntimeError('conv3x3 dgrad: channel sums and BN-backward sums cannot be fused into one launch')
        z, st = bn_prev
        n, h, w, c_out, lddy = nhwc_meta(dy, 'conv3x3 dgrad input')
        zn, zh, zw, zc, ldz = nhwc_meta(z, 'producer layer pre-BN tensor')
        if (zn, zh, zw, zc) != (n, h, w, c_in) or z.dtype != dy.dtype:
            raise RuntimeError('conv3x3 dgrad: producer layer tensor does not match the input gradient')
        if out is None:
            out = new_nhwc(n, h, w, c_in, dy.dtype, dy.device)
        on, oh, ow, oc, lddx = nhwc_meta(out, 'conv3x3 dgrad output')
        if (on, oh, ow, oc) != (n, h, w, c_in) or out.dtype != dy.dtype or wpk_d.dtype != dy.dtype:
            raise RuntimeError('conv3x3 dgrad: output/weight do not match the input')
        if wpk_d.numel() != 9 * c_in * c_out:
            raise RuntimeError('conv3x3 dgrad: packed weight has the wrong size')
        bn_sums = new_stats(c_in, dy.device)
        check(fn['unetk_conv3x3_dgrad_bn_reduce'](dy.data_ptr(), lddy, 0, wpk_d.data_ptr(), out.data_ptr(), lddx, 0,
                                                  z.data_ptr(), ldz, st.scale.data_ptr(), st.shift.data_ptr(),
                                                  st.mean.data_ptr(), st.invstd.data_ptr(), bn_sums.data_ptr(), n, h, w,
                                                  c_in, c_out, dtype_code(dy.dtype), stream_ptr()),
              'unetk_conv3x3_dgrad_bn_reduce')
        return out, bn_sums
    n, h, w, c_out, lddy = nhwc_meta(dy, 'conv3x3 dgrad input')
    if out is None:
        out = new_nhwc(n, h, w, c_in, dy.dtype, dy.device)
    on, oh, ow, oc, lddx = nhwc_meta(out, 'conv3x3 dgrad output')
    if (on, oh, ow, oc) != (n, h, w, c_in) or out.dtype != dy.dtype or wpk_d.dtype != dy.dtype:
        raise RuntimeError('conv3x3 dgrad: output/weight do not match the input')
    if wpk_d.numel() != 9 * c_in * c_out:
        raise RuntimeError('conv3x3 dgrad: packed weight has the wrong size')
    if sums is not None:
        if sums.dtype != torch.float64 or sums.numel() != nat.STAT_REPLICAS * c_in * 2:
            raise RuntimeError('conv3x3 dgrad: sums buffer must be float64 [REPLICAS, c_in, 2]')
        check(fn['unetk_conv3x3_dgrad_sums'](dy.data_ptr(), lddy, 0, wpk_d.data_ptr(), out.data_ptr(), lddx, 0,
                                             sums.data_ptr(), n, h, w, c_in, c_out, dtype_code(dy.dtype), stream_ptr()),
              'unetk_conv3x3_dgrad_sums')
        return out
    check(fn['unetk_conv3x3_dgrad'](dy.data_ptr(), lddy, 0, wpk_d.data_ptr(), out.data_ptr(), lddx, 0,
                                    n, h, w, c_in, c_out, dtype_code(dy.dtype), stream_ptr()), 'unetk_conv3x3_dgrad')
    return out


# ------------------------------------------------------------------------------------ scratch
_WS = {}


def workspace(nbytes, device, tag='ws'):
    """Per-(device, stream, tag) grow-only scratch buffer: reuse is ordered by the stream the kernels run on, so each
    stream gets its own."""
    dev = torch.device(device)
    idx = dev.index if dev.index is not None else torch.cuda.current_device()
    key = (idx, torch.cuda.current_stream(idx).cuda_stream, tag)
    buf = _WS.get(key)
    if buf is None or buf.numel() < nbytes:
        buf = torch.empty(max(int(nbytes), 1 << 20), dtype=torch.uint8, device=device)
        _WS[key] = buf
    return buf


def _ptr(t):
    return t.data_ptr() if t is not None else None


# ------------------------------------------------------------------------------------ wgrad
def conv3x3_wgrad(x, dy, dw):
    """dw [Co,Ci,3,3] fp32 (written) from x [N,H,W,Ci] and dy [N,H,W,Co] (bf16)."""
    require_cuda(x, dy, dw)
    n, h, w, c_in, ldx = nhwc_meta(x, 'wgrad x')
    n2, h2, w2, c_out, lddy = nhwc_meta(dy, 'wgrad dy')
    if (n, h, w) != (n2, h2, w2) or tuple(dw.shape) != (c_out, c_in, 3, 3) or dw.dtype != torch.float32 \
            or not dw.is_contiguous() or x.dtype != dy.dtype:
        raise RuntimeError('conv3x3_wgrad: inconsistent arguments')
    need = fn['unetk_wgrad_workspace_bytes'](0, n, h, w, c_in, c_out)
    if need < 0:
        check(int(need), 'unetk_wgrad_workspace_bytes')
    ws = workspace(need, x.device)
    check(fn['unetk_conv3x3_wgrad'](x.data_ptr(), ldx, dy.data_ptr(), lddy, dw.data_ptr(), ws.data_ptr(), ws.numel(),
                                    n, h, w, c_in, c_out, 0, dtype_code(x.dtype), stream_ptr()), 'unetk_conv3x3_wgrad')
    return dw


# ------------------------------------------------------------------------------------ first layer
def conv3x3_first_fprop(x_f32, w_param, c_out, dtype, out=None, stats=None):
    require_cuda(x_f32, w_param)
    n, h, w, c_in, ldx = nhwc_meta(x_f32, 'first conv input')
    if x_f32.dtype != torch.float32 or w_param.dtype != torch.float32 or not w_param.is_contiguous():
        raise RuntimeError('first conv: image and parameter must be fp32')
    if out is None:
        out = new_nhwc(n, h, w, c_out, dtype, x_f32.device)
    _, _, _, _, ldy = nhwc_meta(out, 'first conv output')
    check(fn['unetk_conv3x3_first_fprop'](x_f32.data_ptr(), ldx, w_param.data_ptr(), out.data_ptr(), ldy, _ptr(stats),
                                          n, h, w, c_in, c_out, dtype_code(dtype), stream_ptr()),
          'unetk_conv3x3_first_fprop')
    return out


def conv3x3_first_stats(x_f32, w_param, c_out, dtype, stats):
    """The first layer's BatchNorm statistics (sums of its output rounded to `dtype`) without storing the output."""
    require_cuda(x_f32, w_param, stats)
    n, h, w, c_in, ldx = nhwc_meta(x_f32, 'first conv input')
    if x_f32.dtype != torch.float32 or w_param.dtype != torch.float32 or not w_param.is_contiguous():
        raise RuntimeError('first conv: image and parameter must be fp32')
    check(fn['unetk_conv3x3_first_fprop'](x_f32.data_ptr(), ldx, w_param.data_ptr(), None, c_out, stats.data_ptr(),
                                          n, h, w, c_in, c_out, dtype_code(dtype), stream_ptr()),
          'unetk_conv3x3_first_fprop')
    return stats


def conv3x3_first_fprop_bn_relu_train(x_f32, w_param, c_out, dtype, state, out=None):
    """The first layer computed again with its BatchNorm + ReLU applied: returns (y, act) -- y is what
    conv3x3_first_fprop stores (saved for backward), act what bn_relu_apply makes of it, bit for bit."""
    require_cuda(x_f32, w_param)
    n, h, w, c_in, ldx = nhwc_meta(x_f32, 'first conv input')
    if x_f32.dtype != torch.float32 or w_param.dtype != torch.float32 or not w_param.is_contiguous():
        raise RuntimeError('first conv: image and parameter must be fp32')
    y = new_nhwc(n, h, w, c_out, dtype, x_f32.device)
    if out is None:
        out = new_nhwc(n, h, w, c_out, dtype, x_f32.device)
    on, oh, ow, oc, lda = nhwc_meta(out, 'first conv activation')
    if (on, oh, ow, oc) != (n, h, w, c_out) or out.dtype != dtype or state.scale.numel() != c_out:
        raise RuntimeError('first conv: output / BatchNorm coefficients do not match')
    check(fn['unetk_conv3x3_first_fprop_bn_relu_train'](x_f32.data_ptr(), ldx, w_param.data_ptr(), state.scale.data_ptr(),
                                                        state.shift.data_ptr(), y.data_ptr(), c_out, out.data_ptr(), lda,
                                                        n, h, w, c_in, c_out, dtype_code(dtype), stream_ptr()),
          'unetk_conv3x3_first_fprop_bn_relu_train')
    return y, out


def conv3x3_first_fprop_bn_relu(x_f32, w_param, c_out, dtype, state, out=None):
    require_cuda(x_f32, w_param)
    n, h, w, c_in, ldx = nhwc_meta(x_f32, 'first conv input')
    if x_f32.dtype != torch.float32 or w_param.dtype != torch.float32 or not w_param.is_contiguous():
        raise RuntimeError('first conv: image and parameter must be fp32')
    if out is None:
        out = new_nhwc(n, h, w, c_out, dtype, x_f32.device)
    on, oh, ow, oc, ldy = nhwc_meta(out, 'first conv output')
    if (on, oh, ow, oc) != (n, h, w, c_out) or out.dtype != dtype or state.scale.numel() != c_out:
        raise RuntimeError('first conv: output / BatchNorm coefficients do not match')
    check(fn['unetk_conv3x3_first_fprop_bn_relu'](x_f32.data_ptr(), ldx, w_param.data_ptr(), state.scale.data_ptr(),
                                                  state.shift.data_ptr(), out.data_ptr(), ldy, n, h, w, c_in, c_out,
                                                  dtype_code(dtype), stream_ptr()), 'unetk_conv3x3_first_fprop_bn_relu')
    return out


def conv3x3_first_wgrad(x_f32, dy, dw):
    require_cuda(x_f32, dy, dw)
    n, h, w, c_in, ldx = nhwc_meta(x_f32, 'first conv input')
    _, _, _, c_out, lddy = nhwc_meta(dy, 'first conv dy')
    need = fn['unetk_conv3x3_first_wgrad_workspace_bytes'](n, h, w, c_out)
    if need < 0:
        check(int(need), 'unetk_conv3x3_first_wgrad_workspace_bytes')
    ws = workspace(need, dy.device)
    check(fn['unetk_conv3x3_first_wgrad'](x_f32.data_ptr(), ldx, dy.data_ptr(), lddy, dw.data_ptr(), ws.data_ptr(),
                                          ws.numel(), n, h, w, c_in, c_out, dtype_code(dy.dtype), stream_ptr()),
          'unetk_conv3x3_first_wgrad')
    return dw


def first_wgrad_bn_fusable(dtype):
    return dtype in (torch.bfloat16, torch.float16)


def bn_relu_bwd_first_wgrad(dout, y, state, gamma, dgamma, dbeta, x_f32, dw, sums=None, frozen=False):
    """BatchNorm+ReLU backward of the FIRST layer when its input needs no gradient: the conv-output gradient has one
    reader, the weight-gradient kernel, which forms it on the fly from (dout, y) -- it is never stored.  Fills
    dgamma/dbeta/dw with what bn_relu_bwd + conv3x3_first_wgrad give (bit for bit: same expression, same rounding).
    sums: as in bn_relu_bwd."""
    require_cuda(dout, y, x_f32, dw)
    n, h, w, c, lddo = nhwc_meta(dout, 'bn grad input')
    yn, yh, yw, yc, ldy = nhwc_meta(y, 'bn saved input')
    xn, xh, xw, c_in, ldx = nhwc_meta(x_f32, 'first conv input')
    if (yn, yh, yw, yc) != (n, h, w, c) or (xn, xh, xw) != (n, h, w) or dout.dtype != y.dtype:
        raise RuntimeError('bn_relu_bwd_first_wgrad: dout, y and the image must cover the same pixels')
    if not first_wgrad_bn_fusable(y.dtype):
        raise RuntimeError('bn_relu_bwd_first_wgrad: 16-bit activations only (use bn_relu_bwd + conv3x3_first_wgrad)')
    dc = dtype_code(y.dtype)
    npix = n * h * w
    have_sums = sums is not None
    if have_sums:
        if sums.dtype != torch.float64 or sums.numel() != nat.STAT_REPLICAS * c * 2:
            raise RuntimeError('bn_relu_bwd_first_wgrad: sums must be float64 [REPLICAS, C, 2]')
    else:
        sums = new_stats(c, y.device)
        check(fn['unetk_bn_relu_bwd_reduce'](dout.data_ptr(), lddo, y.data_ptr(), ldy, state.scale.data_ptr(),
                                             state.shift.data_ptr(), state.mean.data_ptr(), state.invstd.data_ptr(),
                                             sums.data_ptr(), npix, c, dc, stream_ptr()), 'unetk_bn_relu_bwd_reduce')
    coef = torch.empty((3, c), dtype=torch.float32, device=y.device)
    _bn_bwd_finalize(sums, npix, gamma, state, dgamma, dbeta, coef, frozen)
    run_deferred()           # the second layer's weight gradient: parameter gradients complete in layout order (unetk.ddp)
    need = fn['unetk_conv3x3_first_wgrad_workspace_bytes'](n, h, w, c)
    if need < 0:
        check(int(need), 'unetk_conv3x3_first_wgrad_workspace_bytes')
    ws = workspace(need, y.device)
    check(fn['unetk_conv3x3_first_wgrad_bn'](x_f32.data_ptr(), ldx, dout.data_ptr(), lddo, y.data_ptr(), ldy,
                                             state.scale.data_ptr(), state.shift.data_ptr(), state.mean.data_ptr(),
                                             coef.data_ptr(), dw.data_ptr(), ws.data_ptr(), ws.numel(), n, h, w, c_in, c,
                                             dc, stream_ptr()), 'unetk_conv3x3_first_wgrad_bn')
    return dw


def conv3x3_first_dgrad(dy, w_param, c_in):
    """Gradient w.r.t. the fp32 NHWC input image of the first layer: [N,H,W,c_in] fp32."""
    require_cuda(dy, w_param)
    n, h, w, c_out, lddy = nhwc_meta(dy, 'first conv dy')
    dx = new_nhwc(n, h, w, c_in, torch.float32, dy.device)
    check(fn['unetk_conv3x3_first_dgrad'](dy.data_ptr(), lddy, w_param.data_ptr(), dx.data_ptr(), c_in, n, h, w, c_in,
                                          c_out, dtype_code(dy.dtype), stream_ptr()), 'unetk_conv3x3_first_dgrad')
    return dx


# ------------------------------------------------------------------------------------ transposed conv (+pad+cat)
def convT2x2_fwd(x, wpk, bias, cat, choff, c_out, pad_t, pad_l):
    """Writes the up-sampled half into cat[..., choff:choff+c_out] and zeroes the F.pad border."""
    require_cuda(x, wpk, bias, cat)
    n, h, w, c_in, ldx = nhwc_meta(x, 'convT input')
    n2, hs, ws_, ctot, ldc = nhwc_meta(cat, 'concat buffer')
    dc = dtype_code(x.dtype)
    up = cat[..., choff:choff + c_out]
    check(fn['unetk_convT2x2_fwd'](x.data_ptr(), ldx, wpk.data_ptr(), bias.data_ptr(), cat.data_ptr(), ldc, choff,
                                   n, h, w, c_in, c_out, hs, ws_, pad_t, pad_l, dc, stream_ptr()), 'unetk_convT2x2_fwd')
    check(fn['unetk_zero_pad_border'](up.data_ptr(), ldc, n, hs, ws_, c_out, pad_t, pad_l, 2 * h, 2 * w, dc,
                                      stream_ptr()), 'unetk_zero_pad_border')


def convT_dgrad_bn_reduce_is_fused(c_in, c_out, dtype):
    """Does convT2x2_dgrad(bn_prev=...) fold the reduction into the dgrad kernel (else: two kernels, nothing saved)?"""
    return bool(fn['unetk_convT2x2_dgrad_bn_reduce_is_fused'](int(c_in), int(c_out), dtype_code(dtype)))


def convT2x2_dgrad(dcat, choff, wpk_d, c_in, c_out, h, w, pad_t, pad_l, bn_prev=None):
    """bn_prev: (z, BNState) of the conv -> BN -> ReLU layer whose activation the transposed convolution consumed: the
    epilogue also forms that layer's BN-backward sums; returns (dx, bn_sums) then (see conv3x3_dgrad)."""
    require_cuda(dcat, wpk_d)
    n, hs, ws_, ctot, ldd = nhwc_meta(dcat, 'concat gradient')
    dx = new_nhwc(n, h, w, c_in, dcat.dtype, dcat.device)
    if bn_prev is not None:
        z, st = bn_prev
        zn, zh, zw, zc, ldz = nhwc_meta(z, 'producer layer pre-BN tensor')
        if (zn, zh, zw, zc) != (n, h, w, c_in) or z.dtype != dcat.dtype:
            raise RuntimeError('convT dgrad: producer layer tensor does not match the input gradient')
        bn_sums = new_stats(c_in, dcat.device)
        check(fn['unetk_convT2x2_dgrad_bn_reduce'](dcat.data_ptr(), ldd, choff, wpk_d.data_ptr(), dx.data_ptr(), c_in,
                                                   z.data_ptr(), ldz, st.scale.data_ptr(), st.shift.data_ptr(),
                                                   st.mean.data_ptr(), st.invstd.data_ptr(), bn_sums.data_ptr(), n, h, w,
                                                   c_in, c_out, hs, ws_, pad_t, pad_l, dtype_code(dcat.dtype),
                                                   stream_ptr()), 'unetk_convT2x2_dgrad_bn_reduce')
        return dx, bn_sums
    check(fn['unetk_convT2x2_dgrad'](dcat.data_ptr(), ldd, choff, wpk_d.data_ptr(), dx.data_ptr(), c_in, n, h, w, c_in,
                                     c_out, hs, ws_, pad_t, pad_l, dtype_code(dcat.dtype), stream_ptr()),
          'unetk_convT2x2_dgrad')
    return dx


def convT2x2_wgrad(x, dcat, choff, dw, db, pad_t, pad_l, sums=None):
    """sums: per-channel sums of ALL pixels of dcat ([REPLICAS, C_total, 2], from conv3x3_dgrad(sums=...)): the bias
    gradient then needs no pass over dcat (only its pad border is read)."""
    require_cuda(x, dcat, dw, db)
    n, h, w, c_in, ldx = nhwc_meta(x, 'convT input')
    _, hs, ws_, ctot, ldd = nhwc_meta(dcat, 'concat gradient')
    c_out = dw.shape[1]
    dc = dtype_code(x.dtype)
    need = fn['unetk_wgrad_workspace_bytes'](1, n, h, w, c_in, c_out)
    if need < 0:
        check(int(need), 'unetk_wgrad_workspace_bytes')
    ws = workspace(need, x.device)
    check(fn['unetk_convT2x2_wgrad'](x.data_ptr(), ldx, dcat.data_ptr(), ldd, choff, dw.data_ptr(), ws.data_ptr(),
                                     ws.numel(), n, h, w, c_in, c_out, hs, ws_, pad_t, pad_l, 0, dc, stream_ptr()),
          'unetk_convT2x2_wgrad')
    up = dcat[..., choff:choff + c_out]
    scratch = workspace(c_out * 8, x.device, 'chsum')
    if sums is not None:
        if sums.dtype != torch.float64 or sums.numel() != nat.STAT_REPLICAS * ctot * 2:
            raise RuntimeError('convT2x2_wgrad: sums must be float64 [REPLICAS, C_total, 2]')
        check(fn['unetk_convT2x2_bias_grad_from_sums'](sums.data_ptr(), ctot, choff, up.data_ptr(), ldd, n, hs, ws_, c_out,
                                                       pad_t, pad_l, 2 * h, 2 * w, scratch.data_ptr(), db.data_ptr(), dc,
                                                       stream_ptr()), 'unetk_convT2x2_bias_grad_from_sums')
        return
    check(fn['unetk_channel_sum'](up.data_ptr(), ldd, n, hs, ws_, c_out, pad_t, pad_l, 2 * h, 2 * w, scratch.data_ptr(),
                                  db.data_ptr(), dc, stream_ptr()), 'unetk_channel_sum')


def head_fwd_fusable(c, n_classes, dtype):
    """OutConv shapes whose logits the BatchNorm+ReLU apply kernel can emit (a pixel's lanes share a wave)."""
    e = 16 // torch.empty(0, dtype=dtype).element_size()
    ppp = c // e
    return n_classes <= 4 and c % e == 0 and 0 < ppp <= 64 and (ppp & (ppp - 1)) == 0


def bn_relu_apply_head(y, state, w_head, bias, out=None, store_act=True):
    """relu(bn(y)) AND OutConv's logits [N,H,W,n_classes] of it (w_head fp32 [n_classes, C], bias fp32).
    store_act=False: only the logits (returns (None, logits)) -- the activation is one full-resolution tensor that
    nothing reads when the backward pass re-forms it from y (head_bwd_fusable) or does not exist."""
    n, h, w, c, ldy = nhwc_meta(y, 'bn input')
    ncls = w_head.shape[0]
    if w_head.dtype != torch.float32 or bias.dtype != torch.float32 or w_head.numel() != ncls * c \
            or not w_head.is_contiguous():
        raise RuntimeError('bn_relu_apply_head: OutConv weight must be contiguous fp32 [n_classes, C]')
    if not store_act:
        out, out_ptr, ldo = None, None, c
    else:
        if out is None:
            out = new_nhwc(n, h, w, c, y.dtype, y.device)
        _, _, _, _, ldo = nhwc_meta(out, 'bn output')
        out_ptr = out.data_ptr()
    logits = new_nhwc(n, h, w, ncls, y.dtype, y.device)
    check(fn['unetk_bn_relu_apply_head'](y.data_ptr(), ldy, state.scale.data_ptr(), state.shift.data_ptr(),
                                         out_ptr, ldo, w_head.data_ptr(), bias.data_ptr(), ncls,
                                         logits.data_ptr(), ncls, n * h * w, c, dtype_code(y.dtype), stream_ptr()),
          'unetk_bn_relu_apply_head')
    return out, logits


# ------------------------------------------------------------------------------------ BN + ReLU
class BNState:
    """Per-layer scale/shift (+ mean/invstd in train mode): [4, C] fp32."""

    def __init__(self, c, device):
        self.buf = torch.empty((4, c), dtype=torch.float32, device=device)
        self.scale, self.shift, self.mean, self.invstd = self.buf[0], self.buf[1], self.buf[2], self.buf[3]


def bn_train_finalize(stats, count, bn, state, momentum, eps):
    c = state.scale.numel()
    check(fn['unetk_bn_train_finalize'](stats.data_ptr(), count, _ptr(bn.weight), _ptr(bn.bias), _ptr(bn.running_mean),
                                        _ptr(bn.running_var), _ptr(bn.num_batches_tracked), momentum, eps,
                                        state.scale.data_ptr(), state.shift.data_ptr(), state.mean.data_ptr(),
                                        state.invstd.data_ptr(), c, stream_ptr()), 'unetk_bn_train_finalize')


def bn_eval_params(bn, state, eps, for_backward=False):
    """for_backward: also mean / invstd of the running statistics (what the backward passes normalise with)."""
    c = state.scale.numel()
    check(fn['unetk_bn_eval_params'](_ptr(bn.weight), _ptr(bn.bias), bn.running_mean.data_ptr(),
                                     bn.running_var.data_ptr(), eps, state.scale.data_ptr(), state.shift.data_ptr(), c,
                                     stream_ptr()), 'unetk_bn_eval_params')
    if for_backward:
        state.mean.copy_(bn.running_mean)
        state.invstd.copy_(torch.rsqrt(bn.running_var.float() + eps))


def bn_relu_apply(y, state, out=None, pool_out=None):
    """relu(bn(y)); with pool_out also the MaxPool2d(2) of the result (one fused pass)."""
    n, h, w, c, ldy = nhwc_meta(y, 'bn input')
    if out is None:
        out = new_nhwc(n, h, w, c, y.dtype, y.device)
    _, _, _, _, ldo = nhwc_meta(out, 'bn output')
    if pool_out is not None:
        pn, ph, pw, pc, ldp = nhwc_meta(pool_out, 'pool output')
        if (pn, ph, pw, pc) != (n, h // 2, w // 2, c) or pool_out.dtype != y.dtype:
            raise RuntimeError('bn_relu_apply: pool_out must be [N, H//2, W//2, C] of the activation dtype')
        rc = fn['unetk_bn_relu_apply_pool'](y.data_ptr(), ldy, state.scale.data_ptr(), state.shift.data_ptr(),
                                            out.data_ptr(), ldo, pool_out.data_ptr(), ldp, n, h, w, c,
                                            dtype_code(y.dtype), stream_ptr())
        if rc == 0:
            return out
        if rc != -1:                      # -1: channel count the fused kernel does not take -> two passes below
            check(rc, 'unetk_bn_relu_apply_pool')
    check(fn['unetk_bn_relu_apply'](y.data_ptr(), ldy, state.scale.data_ptr(), state.shift.data_ptr(), out.data_ptr(),
                                    ldo, n * h * w, c, dtype_code(y.dtype), stream_ptr()), 'unetk_bn_relu_apply')
    if pool_out is not None:
        _, _, _, _, ldp = nhwc_meta(pool_out, 'pool output')
        check(fn['unetk_maxpool2x2_fwd'](out.data_ptr(), ldo, pool_out.data_ptr(), ldp, n, h, w, c, dtype_code(y.dtype),
                                         stream_ptr()), 'unetk_maxpool2x2_fwd')
    return out


def pool_bwd_fusable(c, dtype):
    """Channel counts the fused (skip + max-pool scatter) BatchNorm backward takes (a 16-byte piece per lane)."""
    e = 16 // torch.empty(0, dtype=dtype).element_size()
    return c % e == 0 and c // e <= 256 and 256 % (c // e) == 0


def head_bwd_fusable(c, n_classes, dtype):
    """OutConv shapes whose input gradient the BatchNorm backward kernels can form on the fly."""
    return n_classes <= 4 and pool_bwd_fusable(c, dtype)


def _bn_bwd_finalize(sums, npix, gamma, state, dgamma, dbeta, coef, frozen):
    """sums -> dgamma, dbeta and the three coefficients of the apply pass (dy = c0*dz - c1 - c2*(y - mean)).
    frozen: BatchNorm ran on its RUNNING statistics (module in eval mode, train.py never does this while training; a
    fine-tuning loop with frozen statistics does): mean and invstd are constants of the graph, so the batch-statistics
    terms vanish -- c0 = gamma*invstd = scale, c1 = c2 = 0 -- and the sums are the parameter gradients themselves."""
    c = state.scale.numel()
    if not frozen:
        check(fn['unetk_bn_relu_bwd_finalize'](sums.data_ptr(), npix, _ptr(gamma), state.invstd.data_ptr(), _ptr(dgamma),
                                               _ptr(dbeta), coef.data_ptr(), c, stream_ptr()), 'unetk_bn_relu_bwd_finalize')
        return
    s = sums.view(nat.STAT_REPLICAS, c, 2).sum(0)
    if dbeta is not None:
        dbeta.copy_(s[:, 0])
    if dgamma is not None:
        dgamma.copy_(s[:, 1])
    coef.zero_()
    coef[0].copy_(state.scale)


# Launches a caller postponed so that they sit BETWEEN HBM-bound passes (unetk.blocks: a layer's weight gradient is
# queued behind the next layer's BatchNorm-backward reduction instead of back to back with its own dgrad).
_DEFERRED = []


def defer(launch):
    _DEFERRED.append(launch)


def run_deferred():
    while _DEFERRED:
        _DEFERRED.pop(0)()


def drop_deferred():
    del _DEFERRED[:]


def bn_relu_bwd(dout, y, state, gamma, dgamma, dbeta, out=None, dpool=None, dhead=None, sums=None, frozen=False):
    """Returns dL/dy (gradient w.r.t. the conv output); writes dgamma/dbeta (fp32 [C]).
    frozen: the forward used the running statistics (see _bn_bwd_finalize).
    dpool: the layer's output also fed MaxPool2d(2) -- `dout` is then only the skip-connection gradient and the
    pooled gradient [N,H//2,W//2,C] is scattered in on the fly (no max-pool backward pass).
    dhead: (dlogits [N,H,W,n_classes], OutConv weight fp32 [n_classes, C], dw, db) -- the layer's output fed OutConv;
    `dout` is None and the 1x1 convolution's input gradient is formed on the fly; dw/db (fp32 gradient buffers of
    OutConv's weight [n_classes,C,1,1] and bias, or None) are filled as a by-product of the reduction pass.
    sums: the reduction pass's result when the dgrad that wrote `dout` already formed it (conv3x3_dgrad(bn_prev=...))."""
    if sums is not None and (dpool is not None or dhead is not None):
        raise RuntimeError('bn_relu_bwd: precomputed sums only apply to the plain (no pool, no head) layer')
    if dhead is not None:
        return _bn_relu_bwd_head(dhead[0], dhead[1], y, state, gamma, dgamma, dbeta, out, *dhead[2:], frozen=frozen)
    n, h, w, c, lddo = nhwc_meta(dout, 'bn grad input')
    _, _, _, _, ldy = nhwc_meta(y, 'bn saved input')
    dc = dtype_code(y.dtype)
    npix = n * h * w
    have_sums = sums is not None
    if have_sums:
        if sums.dtype != torch.float64 or sums.numel() != nat.STAT_REPLICAS * c * 2:
            raise RuntimeError('bn_relu_bwd: sums must be float64 [REPLICAS, C, 2]')
    else:
        sums = new_stats(c, y.device)
    coef = torch.empty((3, c), dtype=torch.float32, device=y.device)
    if out is None:
        out = new_nhwc(n, h, w, c, y.dtype, y.device)
    _, _, _, _, ldo = nhwc_meta(out, 'bn grad output')
    if have_sums:
        pass
    elif dpool is not None:
        pn, ph, pw, pc, lddp = nhwc_meta(dpool, 'pool grad')
        if (pn, ph, pw, pc) != (n, h // 2, w // 2, c) or dpool.dtype != y.dtype or dout.dtype != y.dtype:
            raise RuntimeError('bn_relu_bwd: dpool must be [N, H//2, W//2, C] of the activation dtype')
        check(fn['unetk_bn_relu_bwd_reduce_pool'](dout.data_ptr(), lddo, dpool.data_ptr(), lddp, y.data_ptr(), ldy,
                                                  state.scale.data_ptr(), state.shift.data_ptr(), state.mean.data_ptr(),
                                                  state.invstd.data_ptr(), sums.data_ptr(), n, h, w, c, dc,
                                                  stream_ptr()), 'unetk_bn_relu_bwd_reduce_pool')
    else:
        check(fn['unetk_bn_relu_bwd_reduce'](dout.data_ptr(), lddo, y.data_ptr(), ldy, state.scale.data_ptr(),
                                             state.shift.data_ptr(), state.mean.data_ptr(), state.invstd.data_ptr(),
                                             sums.data_ptr(), npix, c, dc, stream_ptr()), 'unetk_bn_relu_bwd_reduce')
    if not have_sums:
        run_deferred()                       # behind the reduction pass ...
    _bn_bwd_finalize(sums, npix, gamma, state, dgamma, dbeta, coef, frozen)
    if dpool is not None:
        check(fn['unetk_bn_relu_bwd_apply_pool'](dout.data_ptr(), lddo, dpool.data_ptr(), lddp, y.data_ptr(), ldy,
                                                 state.scale.data_ptr(), state.shift.data_ptr(), state.mean.data_ptr(),
                                                 coef.data_ptr(), out.data_ptr(), ldo, n, h, w, c, dc, stream_ptr()),
              'unetk_bn_relu_bwd_apply_pool')
    else:
        check(fn['unetk_bn_relu_bwd_apply'](dout.data_ptr(), lddo, y.data_ptr(), ldy, state.scale.data_ptr(),
                                            state.shift.data_ptr(), state.mean.data_ptr(), coef.data_ptr(),
                                            out.data_ptr(), ldo, npix, c, dc, stream_ptr()), 'unetk_bn_relu_bwd_apply')
    if have_sums:
        run_deferred()                       # ... or, when the producer's dgrad already reduced, behind the apply pass
    return out


def new_zero_doubles(n, device):
    t = _ARENA.take(n, device)
    return t if t is not None else torch.zeros(n, dtype=torch.float64, device=device)


def _bn_relu_bwd_head(dl, w_head, y, state, gamma, dgamma, dbeta, out, dw_head=None, db_head=None, frozen=False):
    n, h, w, c, ldy = nhwc_meta(y, 'bn saved input')
    dn, dh, dw_, ncls, lddl = nhwc_meta(dl, 'head grad')
    if (dn, dh, dw_) != (n, h, w) or dl.dtype != y.dtype or w_head.dtype != torch.float32 or w_head.numel() != ncls * c \
            or not w_head.is_contiguous():
        raise RuntimeError('bn_relu_bwd: dlogits / OutConv weight do not match the layer')
    dc = dtype_code(y.dtype)
    npix = n * h * w
    sums = new_stats(c, y.device)
    coef = torch.empty((3, c), dtype=torch.float32, device=y.device)
    if out is None:
        out = new_nhwc(n, h, w, c, y.dtype, y.device)
    _, _, _, _, ldo = nhwc_meta(out, 'bn grad output')
    hsums = None
    if dw_head is not None:
        if dw_head.dtype != torch.float32 or dw_head.numel() != ncls * c or not dw_head.is_contiguous():
            raise RuntimeError('bn_relu_bwd: OutConv weight-gradient buffer must be contiguous fp32 [n_classes, C, 1, 1]')
        hsums = new_zero_doubles(nat.STAT_REPLICAS * ncls * (c + 1), y.device)
    check(fn['unetk_bn_relu_bwd_reduce_head'](dl.data_ptr(), lddl, w_head.data_ptr(), ncls, y.data_ptr(), ldy,
                                              state.scale.data_ptr(), state.shift.data_ptr(), state.mean.data_ptr(),
                                              state.invstd.data_ptr(), sums.data_ptr(), _ptr(hsums), npix, c, dc,
                                              stream_ptr()), 'unetk_bn_relu_bwd_reduce_head')
    if hsums is not None:
        check(fn['unetk_conv1x1_wgrad_finalize'](hsums.data_ptr(), dw_head.data_ptr(), _ptr(db_head), c, ncls,
                                                 stream_ptr()), 'unetk_conv1x1_wgrad_finalize')
    _bn_bwd_finalize(sums, npix, gamma, state, dgamma, dbeta, coef, frozen)
    check(fn['unetk_bn_relu_bwd_apply_head'](dl.data_ptr(), lddl, w_head.data_ptr(), ncls, y.data_ptr(), ldy,
                                             state.scale.data_ptr(), state.shift.data_ptr(), state.mean.data_ptr(),
                                             coef.data_ptr(), out.data_ptr(), ldo, npix, c, dc, stream_ptr()),
          'unetk_bn_relu_bwd_apply_head')
    return out


# ------------------------------------------------------------------------------------ max pool
def maxpool_into(x, pool_out):
    n, h, w, c, ldx = nhwc_meta(x, 'pool input')
    pn, ph, pw, pc, ldp = nhwc_meta(pool_out, 'pool output')
    if (pn, ph, pw, pc) != (n, h // 2, w // 2, c) or pool_out.dtype != x.dtype:
        raise RuntimeError('maxpool: pool_out must be [N, H//2, W//2, C] of the activation dtype')
    check(fn['unetk_maxpool2x2_fwd'](x.data_ptr(), ldx, pool_out.data_ptr(), ldp, n, h, w, c, dtype_code(x.dtype),
                                     stream_ptr()), 'unetk_maxpool2x2_fwd')
    return pool_out


def maxpool_fwd(x):
    n, h, w, c, ldx = nhwc_meta(x, 'pool input')
    out = new_nhwc(n, h // 2, w // 2, c, x.dtype, x.device)
    check(fn['unetk_maxpool2x2_fwd'](x.data_ptr(), ldx, out.data_ptr(), c, n, h, w, c, dtype_code(x.dtype),
                                     stream_ptr()), 'unetk_maxpool2x2_fwd')
    return out


def maxpool_bwd(x, dpool, dskip=None, out=None):
    n, h, w, c, ldx = nhwc_meta(x, 'pool input')
    _, _, _, _, lddp = nhwc_meta(dpool, 'pool grad')
    ldds = 0
    if dskip is not None:
        _, _, _, _, ldds = nhwc_meta(dskip, 'skip grad')
    if out is None:
        out = new_nhwc(n, h, w, c, x.dtype, x.device)
    _, _, _, _, lddx = nhwc_meta(out, 'pool dx')
    check(fn['unetk_maxpool2x2_bwd'](x.data_ptr(), ldx, dpool.data_ptr(), lddp, _ptr(dskip), ldds, out.data_ptr(), lddx,
                                     n, h, w, c, dtype_code(x.dtype), stream_ptr()), 'unetk_maxpool2x2_bwd')
    return out


# ------------------------------------------------------------------------------------ bilinear (+pad+cat)
def bilinear_fwd(x, cat, choff, pad_t, pad_l):
    n, h, w, c, ldx = nhwc_meta(x, 'upsample input')
    _, hs, ws_, _, ldc = nhwc_meta(cat, 'concat buffer')
    up = cat[..., choff:choff + c]
    check(fn['unetk_bilinear2x_fwd'](x.data_ptr(), ldx, up.data_ptr(), ldc, n, h, w, c, hs, ws_, pad_t, pad_l,
                                     dtype_code(x.dtype), stream_ptr()), 'unetk_bilinear2x_fwd')


def bilinear_bwd(dcat, choff, c, h, w, pad_t, pad_l):
    n, hs, ws_, _, ldd = nhwc_meta(dcat, 'concat gradient')
    dx = new_nhwc(n, h, w, c, dcat.dtype, dcat.device)
    up = dcat[..., choff:choff + c]
    check(fn['unetk_bilinear2x_bwd'](up.data_ptr(), ldd, dx.data_ptr(), c, n, h, w, c, hs, ws_, pad_t, pad_l,
                                     dtype_code(dcat.dtype), stream_ptr()), 'unetk_bilinear2x_bwd')
    return dx


def copy_channels(src, dst):
    n, h, w, c, lds = nhwc_meta(src, 'copy src')
    n2, h2, w2, c2, ldd = nhwc_meta(dst, 'copy dst')
    if (n, h, w, c) != (n2, h2, w2, c2) or src.dtype != dst.dtype:
        raise RuntimeError('copy_channels: shape/dtype mismatch')
    check(fn['unetk_copy_channels'](src.data_ptr(), lds, dst.data_ptr(), ldd, n * h * w, c, dtype_code(src.dtype),
                                    stream_ptr()), 'unetk_copy_channels')


# ------------------------------------------------------------------------------------ 1x1 head
def conv1x1_fwd(x, w_param, bias, n_classes):
    n, h, w, c, ldx = nhwc_meta(x, 'head input')
    out = new_nhwc(n, h, w, n_classes, x.dtype, x.device)
    check(fn['unetk_conv1x1_fwd'](x.data_ptr(), ldx, w_param.data_ptr(), bias.data_ptr(), out.data_ptr(), n_classes,
                                  n * h * w, c, n_classes, dtype_code(x.dtype), stream_ptr()), 'unetk_conv1x1_fwd')
    return out


def conv1x1_bwd(x, dl, w_param, dw, db, need_dx=True):
    n, h, w, c, ldx = nhwc_meta(x, 'head input')
    _, _, _, ncls, lddl = nhwc_meta(dl, 'head grad')
    dc = dtype_code(x.dtype)
    npix = n * h * w
    need = fn['unetk_conv1x1_wgrad_workspace_bytes'](c, ncls)
    ws = workspace(need, x.device)
    check(fn['unetk_conv1x1_wgrad'](x.data_ptr(), ldx, dl.data_ptr(), lddl, dw.data_ptr(), _ptr(db), ws.data_ptr(),
                                    ws.numel(), npix, c, ncls, dc, stream_ptr()), 'unetk_conv1x1_wgrad')
    dx = None
    if need_dx:
        dx = new_nhwc(n, h, w, c, x.dtype, x.device)
        check(fn['unetk_conv1x1_dgrad'](dl.data_ptr(), lddl, w_param.data_ptr(), dx.data_ptr(), c, npix, c, ncls, dc,
                                        stream_ptr()), 'unetk_conv1x1_dgrad')
    return dx


# ------------------------------------------------------------------------------------ loss
def seg_loss_fwd(logits, mask, eps=1e-6):
    """logits NHWC [N,H,W,C]; mask int64 [N,H,W].  Returns out4 = {loss, dice, alpha, beta} (device floats)."""
    n, h, w, c, ldz = nhwc_meta(logits, 'logits')
    if mask.dtype != torch.int64 or not mask.is_contiguous() or mask.numel() != n * h * w:
        raise RuntimeError('seg_loss: mask must be a contiguous int64 [N,H,W] tensor')
    sums = torch.empty(4, dtype=torch.float64, device=logits.device)
    out4 = torch.empty(4, dtype=torch.float32, device=logits.device)
    check(fn['unetk_seg_loss_fwd'](logits.data_ptr(), ldz, mask.data_ptr(), n * h * w, c, eps, sums.data_ptr(),
                                   out4.data_ptr(), dtype_code(logits.dtype), stream_ptr()), 'unetk_seg_loss_fwd')
    return out4


def seg_loss_bwd(logits, mask, out4, grad_out=None):
    n, h, w, c, ldz = nhwc_meta(logits, 'logits')
    dl = new_nhwc(n, h, w, c, logits.dtype, logits.device)
    check(fn['unetk_seg_loss_bwd'](logits.data_ptr(), ldz, mask.data_ptr(), n * h * w, c, out4.data_ptr(),
                                   _ptr(grad_out), dl.data_ptr(), c, dtype_code(logits.dtype), stream_ptr()),
          'unetk_seg_loss_bwd')
    return dl


def dice_fwd(inp, tgt, n_groups, group_size, eps, need_coef):
    sums = torch.empty(n_groups * 3, dtype=torch.float64, device=inp.device)
    out = torch.empty(1, dtype=torch.float32, device=inp.device)
    coef = torch.empty(n_groups * 2, dtype=torch.float32, device=inp.device) if need_coef else None
    check(fn['unetk_dice_fwd'](inp.data_ptr(), tgt.data_ptr(), n_groups, group_size, eps, sums.data_ptr(),
                               out.data_ptr(), _ptr(coef), stream_ptr()), 'unetk_dice_fwd')
    return out, coef


def dice_bwd(tgt, coef, grad_out, sign, n_groups, group_size):
    dx = torch.empty(n_groups * group_size, dtype=torch.float32, device=tgt.device)
    check(fn['unetk_dice_bwd'](tgt.data_ptr(), coef.data_ptr(), _ptr(grad_out), sign, n_groups, group_size,
                               dx.data_ptr(), stream_ptr()), 'unetk_dice_bwd')
    return dx


# ------------------------------------------------------------------------------------ validation metric
def eval_dice(logits, mask, first_class=1, eps=1e-6):
    """multiclass Dice of argmax(logits) against an int64 mask, classes >= first_class (evaluate.py:31-36).
    logits: [N,C,H,W] (any memory format; read as NHWC) bf16/fp32; mask: [N,H,W] int64.  Returns a 0-dim fp32 tensor."""
    require_cuda(logits, mask)
    if logits.dim() != 4 or mask.dim() != 3 or mask.dtype != torch.int64 or \
            (logits.shape[0],) + tuple(logits.shape[2:]) != tuple(mask.shape):
        raise RuntimeError('eval_dice: logits [N,C,H,W] and an int64 mask [N,H,W] are required')
    x = to_nhwc(logits)
    n, h, w, c, ld = nhwc_meta(x, 'eval_dice logits')
    m = mask.contiguous()
    need = fn['unetk_eval_dice_workspace_bytes'](n, c)
    if need < 0:
        check(int(need), 'unetk_eval_dice_workspace_bytes')
    ws = workspace(need, x.device, tag='eval')
    out = torch.empty(1, dtype=torch.float32, device=x.device)
    check(fn['unetk_eval_dice'](x.data_ptr(), ld, m.data_ptr(), n, h * w, c, first_class, eps, ws.data_ptr(), ws.numel(),
                                out.data_ptr(), dtype_code(x.dtype), stream_ptr()), 'unetk_eval_dice')
    return out.reshape(())


# ------------------------------------------------------------------------------------ prediction mask
def predict_mask(logits, out_h, out_w, threshold=0.5):
    """F.interpolate(logits, (out_h, out_w), mode='bilinear') then argmax (C > 1) or sigmoid > threshold (C == 1),
    fused (predict.py:26-31).  logits [N,C,h,w] bf16/fp32 -> int64 [N,out_h,out_w]."""
    require_cuda(logits)
    if logits.dim() != 4:
        raise RuntimeError('predict_mask: logits [N,C,h,w] required')
    x = to_nhwc(logits)
    n, h, w, c, ld = nhwc_meta(x, 'predict_mask logits')
    mask = torch.empty((n, out_h, out_w), dtype=torch.int64, device=x.device)
    check(fn['unetk_predict_mask'](x.data_ptr(), ld, n, h, w, c, out_h, out_w, float(threshold), mask.data_ptr(),
                                   dtype_code(x.dtype), stream_ptr()), 'unetk_predict_mask')
    return mask
