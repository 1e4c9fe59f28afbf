"""CPU restatement of the reference's input formatting, BasicDataset.preprocess (utils/data_loading.py:65-91).

TEST INFRASTRUCTURE ONLY (tests/, __graft_entry__.smoke, bench cpu leg): the product never imports this.

The arithmetic of `pil_img.resize` (data_loading.py:69) lives in Pillow, a third-party dependency that is not under
/root/reference (requirements.txt pins Pillow==9.3.0; this image has 12.2.0 -- the two paths below are unchanged
between them).  Its published algorithm is restated in NumPy:

  BICUBIC   libImaging/Resample.c: precompute_coeffs (windows + Keys cubic a=-0.5, normalised in double),
            normalize_coeffs_8bpc (22-bit fixed point), ImagingResampleHorizontal_8bpc then
            ImagingResampleVertical_8bpc with clip8 between the passes; the horizontal pass only produces the source
            rows the vertical pass reads (ImagingResampleInner).
  NEAREST   libImaging/Geometry.c ImagingScaleAffine: source positions by repeated double addition, COORD().

Pinned (tests/test_oracle_golden.py) against tests/golden/preprocess.npz -- outputs of the reference's own
BasicDataset.preprocess run in the build container (oracle/make_golden.py section 7) -- and, where Pillow is
importable, against Pillow itself on random sizes.
"""
import math

import numpy as np

PRECISION_BITS = 32 - 8 - 2


def _bicubic(x):
    a = -0.5
    x = abs(x)
    if x < 1.0:
        return ((a + 2.0) * x - (a + 3.0)) * x * x + 1
    if x < 2.0:
        return (((x - 5) * x + 8) * x - 4) * a
    return 0.0


def precompute_coeffs(in_size, out_size):
    """Resample.c precompute_coeffs + normalize_coeffs_8bpc for the box [0, in_size): (bounds [out,2], kk [out,ksize])."""
    in0, in1 = np.float32(0.0), np.float32(in_size)
    scale = filterscale = float(in1 - in0) / out_size
    if filterscale < 1.0:
        filterscale = 1.0
    support = 2.0 * filterscale
    ksize = int(math.ceil(support)) * 2 + 1
    bounds = np.zeros((out_size, 2), dtype=np.int32)
    kk = np.zeros((out_size, ksize), dtype=np.int32)
    for xx in range(out_size):
        center = float(in0) + (xx + 0.5) * scale
        ss = 1.0 / filterscale
        xmin = max(int(center - support + 0.5), 0)
        xmax = min(int(center + support + 0.5), in_size) - xmin
        k = [_bicubic((x + xmin - center + 0.5) * ss) for x in range(xmax)]
        ww = 0.0
        for w in k:
            ww += w
        if ww != 0.0:
            k = [w / ww for w in k]
        bounds[xx] = (xmin, xmax)
        for x, p in enumerate(k):
            kk[xx, x] = int(-0.5 + p * (1 << PRECISION_BITS)) if p < 0 else int(0.5 + p * (1 << PRECISION_BITS))
    return bounds, kk


def _clip8(v):
    return np.clip(v >> PRECISION_BITS, 0, 255).astype(np.uint8)


def _pass_rows(src, bounds, kk):
    """1-D convolution along axis 0 of src [len, ...] uint8 -> [out, ...] uint8."""
    out = np.empty((bounds.shape[0],) + src.shape[1:], dtype=np.uint8)
    s32 = src.astype(np.int32)
    for o, (lo, cnt) in enumerate(bounds):
        acc = np.full(src.shape[1:], 1 << (PRECISION_BITS - 1), dtype=np.int32)
        for i in range(cnt):
            acc = acc + s32[lo + i] * kk[o, i]          # int32 wrap-around like the C code (never reached in range)
        out[o] = _clip8(acc)
    return out


def resize_bicubic(img, out_w, out_h):
    """img [H, W] or [H, W, C] uint8 -> resized uint8 (ImagingResampleInner)."""
    in_h, in_w = img.shape[:2]
    if (in_w, in_h) == (out_w, out_h):
        return img.copy()                                # Image.resize: same size -> copy
    need_h, need_v = out_w != in_w, out_h != in_h
    bv, kv = precompute_coeffs(in_h, out_h)
    cur = img
    if need_h:
        bh, kh = precompute_coeffs(in_w, out_w)
        first = int(bv[0, 0]) if need_v else 0
        last = int(bv[-1, 0] + bv[-1, 1]) if need_v else in_h
        rows = cur[first:last]
        cur = np.swapaxes(_pass_rows(np.swapaxes(rows, 0, 1), bh, kh), 0, 1)
        bv = bv.copy()
        bv[:, 0] -= first
    if need_v:
        cur = _pass_rows(cur, bv, kv)
    return np.ascontiguousarray(cur)


def nearest_table(in_size, out_size):
    a = float(np.float32(in_size) - np.float32(0.0)) / out_size
    xo = 0.0 + a * 0.5
    tab = np.empty(out_size, dtype=np.int32)
    for x in range(out_size):
        xin = -1 if xo < 0.0 else int(xo)
        tab[x] = xin if 0 <= xin < in_size else -1
        xo += a
    return tab


def resize_nearest(img, out_w, out_h):
    in_h, in_w = img.shape[:2]
    if (in_w, in_h) == (out_w, out_h):
        return img.copy()
    xt, yt = nearest_table(in_w, out_w), nearest_table(in_h, out_h)
    out = img[np.clip(yt, 0, None)][:, np.clip(xt, 0, None)].copy()
    out[yt < 0] = 0
    out[:, xt < 0] = 0
    return out


def preprocess(mask_values, img_u8, scale, is_mask):
    """data_loading.py:65-91 on the decoded pixels `np.asarray(pil_img)` ([H,W] or [H,W,C] uint8)."""
    h, w = img_u8.shape[:2]
    new_w, new_h = int(scale * w), int(scale * h)
    assert new_w > 0 and new_h > 0, 'Scale is too small, resized images would have no pixel'
    if is_mask:
        img = resize_nearest(img_u8, new_w, new_h)
        mask = np.zeros((new_h, new_w), dtype=np.int64)
        for i, v in enumerate(mask_values):
            hit = (img == v) if img.ndim == 2 else (img == v).all(-1)
            mask[hit] = i
        return mask
    img = resize_bicubic(img_u8, new_w, new_h)
    img = img[np.newaxis, ...] if img.ndim == 2 else img.transpose((2, 0, 1))
    if (img > 1).any():
        img = img / 255.0
    return img
