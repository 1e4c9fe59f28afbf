"""Thin Python wrappers over the C ABI (include/unetk.h): shape checks, output allocation, launch.

Tensors here are NHWC *views*: shape [N,H,W,C], stride[-1] == 1, stride[-2] == ld >= C (a channel
slice of a wider buffer is fine), rows and images dense.  Modules convert from/to PyTorch's logical
NCHW with `.permute` -- on a channels_last tensor that is free.
"""
import torch

from . import _native as nat
from ._native import fn, check, dtype_code, stream_ptr, require_cuda


def nhwc_meta(t, what='tensor'):
    if t.dim() != 4 or t.stride(3) != 1:
        raise RuntimeError(f'{what}: expected an NHWC view with unit channel stride, got shape {tuple(t.shape)} '
                           f'strides {t.stride()}')
    n, h, w, c = t.shape
    ld = t.stride(2) if w > 1 else max(c, t.stride(2))
    if (h > 1 and t.stride(1) != w * ld) or (n > 1 and t.stride(0) != h * w * ld) or ld < c:
        raise RuntimeError(f'{what}: rows/images must be dense (strides {t.stride()}, shape {tuple(t.shape)})')
    return n, h, w, c, ld


def new_nhwc(n, h, w, c, dtype, device):
    return torch.empty((n, h, w, c), dtype=dtype, device=device)


def to_nhwc(x):
    """Logical NCHW (any strides) -> NHWC view; copies only if the memory is not channels_last."""
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
def new_stats(c, device):
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


def conv3x3_dgrad(dy, wpk_d, c_in, out=None):
    require_cuda(dy, wpk_d)
    n, h, w, c_out, lddy = nhwc_meta(dy, 'conv3x3 dgrad input')
    if out is None:
        out = new_nhwc(n, h, w, c_in, dy.dtype, dy.device)
    on, oh, ow, oc, lddx = nhwc_meta(out, 'conv3x3 dgrad output')
    if (on, oh, ow, oc) != (n, h, w, c_in) or out.dtype != dy.dtype or wpk_d.dtype != dy.dtype:
        raise RuntimeError('conv3x3 dgrad: output/weight do not match the input')
    if wpk_d.numel() != 9 * c_in * c_out:
        raise RuntimeError('conv3x3 dgrad: packed weight has the wrong size')
    check(fn['unetk_conv3x3_dgrad'](dy.data_ptr(), lddy, 0, wpk_d.data_ptr(), out.data_ptr(), lddx, 0,
                                    n, h, w, c_in, c_out, dtype_code(dy.dtype), stream_ptr()), 'unetk_conv3x3_dgrad')
    return out
