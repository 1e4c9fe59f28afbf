"""GPU input formatting: the reference's BasicDataset.preprocess (utils/data_loading.py:65-91) on decoded uint8
pixels that are already in HBM -- Pillow's BICUBIC / NEAREST resize restated bit-exactly (csrc/preprocess.hip),
the /255 rule and the mask_values lookup.  SURVEY.md 8(f) row 4: with this the loader ships raw uint8 frames
(7.4 MB per Carvana image) and no CPU resize pass stands between the decoder and the 8 GPUs.
"""
import numpy as np
import torch

from . import _native as nat
from .ops import check, require_cuda, stream_ptr, workspace

fn = nat.fn
_PLANS = {}


def _plan(in_w, in_h, out_w, out_h, filt, device):
    key = (in_w, in_h, out_w, out_h, filt, str(device))
    hit = _PLANS.get(key)
    if hit is None:
        nbytes = int(fn['unetk_resize_plan_build'](in_w, in_h, out_w, out_h, filt, None, 0))
        if nbytes < 0:
            check(nbytes, 'unetk_resize_plan_build')
        host = np.zeros(nbytes, dtype=np.uint8)
        r = int(fn['unetk_resize_plan_build'](in_w, in_h, out_w, out_h, filt, host.ctypes.data, nbytes))
        if r < 0:
            check(r, 'unetk_resize_plan_build')
        hit = (host, torch.from_numpy(host).to(device))
        _PLANS[key] = hit
    return hit


def _as_batch(px, what):
    """[H,W] | [H,W,C] | [N,H,W,C] uint8 (what np.asarray(pil_img) gives, optionally batched) -> ([N,H,W,C], kind)."""
    if px.dtype != torch.uint8:
        raise RuntimeError(f'{what}: decoded pixels must be uint8 (np.asarray of an 8-bit PIL image)')
    require_cuda(px)
    if px.dim() == 2:
        return px.contiguous()[None, :, :, None], 'hw'
    if px.dim() == 3:
        return px.contiguous()[None], 'hwc'
    if px.dim() == 4:
        return px.contiguous(), 'nhwc'
    raise RuntimeError(f'{what}: expected [H,W], [H,W,C] or [N,H,W,C]')


def scaled_size(w, h, scale):
    new_w, new_h = int(scale * w), int(scale * h)                  # data_loading.py:67
    assert new_w > 0 and new_h > 0, 'Scale is too small, resized images would have no pixel'
    return new_w, new_h


def preprocess_image(pixels, scale):
    """uint8 pixels -> float32 in the reference's logical layout ([C,H',W'], or [N,C,H',W'] for a batch) backed by
    NHWC memory (channels_last: what train.py:97 asks for and the first convolution reads)."""
    x, kind = _as_batch(pixels, 'preprocess_image')
    n, h, w, c = x.shape
    if c not in (1, 3):
        raise RuntimeError('preprocess_image: 1 (L) or 3 (RGB) channels (Pillow premultiplies alpha for RGBA resizes)')
    new_w, new_h = scaled_size(w, h, scale)
    host, dev = _plan(w, h, new_w, new_h, 3, x.device)
    need = int(fn['unetk_resize_workspace_bytes'](host.ctypes.data, n, c))
    if need < 0:
        check(need, 'unetk_resize_workspace_bytes')
    ws = workspace(need, x.device, 'preprocess')
    out = torch.empty((n, new_h, new_w, c), dtype=torch.float32, device=x.device)
    check(fn['unetk_preprocess_image'](x.data_ptr(), n, c, host.ctypes.data, dev.data_ptr(), out.data_ptr(),
                                       ws.data_ptr(), ws.numel(), stream_ptr()), 'unetk_preprocess_image')
    nchw = out.permute(0, 3, 1, 2)
    return nchw if kind == 'nhwc' else nchw[0]


def preprocess_mask(pixels, mask_values, scale):
    """uint8 mask pixels -> int64 class indices [H',W'] (or [N,H',W'])."""
    x, kind = _as_batch(pixels, 'preprocess_mask')
    n, h, w, c = x.shape
    if c > 4:
        raise RuntimeError('preprocess_mask: at most 4 channels')
    new_w, new_h = scaled_size(w, h, scale)
    vals = np.asarray(mask_values)
    if vals.size == 0:
        table = np.zeros((0, c), dtype=np.uint8)
    else:
        vals = vals.reshape(len(mask_values), -1)
        if vals.shape[1] != c:
            raise RuntimeError('preprocess_mask: mask_values do not have the mask\'s channel count')
        if (vals < 0).any() or (vals > 255).any():
            raise RuntimeError('preprocess_mask: 8-bit masks only')
        table = vals.astype(np.uint8)
    host, dev = _plan(w, h, new_w, new_h, 0, x.device)
    tdev = torch.from_numpy(np.ascontiguousarray(table)).to(x.device) if table.size else None
    out = torch.empty((n, new_h, new_w), dtype=torch.int64, device=x.device)
    check(fn['unetk_preprocess_mask'](x.data_ptr(), n, c, host.ctypes.data, dev.data_ptr(),
                                      tdev.data_ptr() if tdev is not None else None, table.shape[0], out.data_ptr(),
                                      stream_ptr()), 'unetk_preprocess_mask')
    return out if kind == 'nhwc' else out[0]
