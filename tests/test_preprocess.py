"""BasicDataset.preprocess (reference utils/data_loading.py:65-91): oracle pinned to the reference-generated fixture
(CPU), the host-side plan builder of the C ABI against the oracle (CPU), and the HIP path against fixture, oracle
and -- where importable -- Pillow itself at the Carvana size (GPU).  Bar: BIT-EXACT (byte/integer work; the float
conversion is float32(double(v)/255.0), also exact)."""
import hashlib
import os

import numpy as np
import pytest
import torch

from oracle import pil_resize as R
from oracle import weights as W

GOLD = np.load(os.path.join(os.path.dirname(__file__), 'golden', 'preprocess.npz'))
IMAGE_CASES = ['rgb_half', 'rgb_third', 'rgb_same', 'gray_half', 'rgb_075', 'rgb_tiny']


def _case(tag):
    h, w, c, sc = GOLD[tag + '.cfg']
    return W.photo('pre.' + tag, int(h), int(w), int(c)), float(sc)


def _masks():
    return {'mask2': ([0, 255], 0.5), 'mask3': ([0, 7, 200], 0.3), 'maskrgb': (GOLD['maskrgb.values'].tolist(), 0.5)}


# ------------------------------------------------------------------------------------------------ CPU
@pytest.mark.parametrize('tag', IMAGE_CASES)
def test_oracle_image_matches_reference_fixture(tag):
    a, sc = _case(tag)
    out = R.preprocess(None, a, sc, is_mask=False)
    assert np.array_equal(np.asarray(out, dtype=np.float64).astype(np.float32), GOLD[tag + '.out'])


def test_oracle_dark_image_is_not_divided():
    out = R.preprocess(None, GOLD['dark.in'], 0.5, is_mask=False)
    assert np.array_equal(out.astype(np.float32), GOLD['dark.out']) and GOLD['dark.out'].max() == 1.0


@pytest.mark.parametrize('tag', ['mask2', 'mask3', 'maskrgb'])
def test_oracle_mask_matches_reference_fixture(tag):
    values, sc = _masks()[tag]
    out = R.preprocess(values, GOLD[tag + '.in'], sc, is_mask=True)
    assert out.dtype == np.int64 and np.array_equal(out, GOLD[tag + '.out'])


def test_oracle_carvana_size_hash():
    a = W.photo('pre.carvana', 1280, 1918, 3)
    u8 = R.resize_bicubic(a, 959, 640)
    assert np.array_equal(u8[[0, 1, 319, 638, 639]], GOLD['carvana.rows'])
    assert hashlib.sha256(u8.tobytes()).hexdigest() == str(GOLD['carvana.sha256'])
    mk = (W.int_wave('pre.carvana_mask', (1280, 1918), 2) * 255).astype(np.uint8)
    mo = R.preprocess([0, 255], mk, 0.5, is_mask=True)
    assert hashlib.sha256(np.ascontiguousarray(mo).tobytes()).hexdigest() == str(GOLD['carvana_mask.sha256'])


def test_oracle_matches_pillow_on_random_sizes():
    Image = pytest.importorskip('PIL.Image')
    rng = np.random.default_rng(7)
    for _ in range(12):
        h, w, c = int(rng.integers(5, 150)), int(rng.integers(5, 150)), int(rng.choice([1, 3]))
        sc = float(rng.uniform(0.05, 1.0))
        nw, nh = int(sc * w), int(sc * h)
        if nw < 1 or nh < 1:
            continue
        a = rng.integers(0, 256, (h, w, c) if c > 1 else (h, w), dtype=np.uint8)
        im = Image.fromarray(a)
        assert np.array_equal(np.asarray(im.resize((nw, nh), resample=Image.BICUBIC)), R.resize_bicubic(a, nw, nh))
        assert np.array_equal(np.asarray(im.resize((nw, nh), resample=Image.NEAREST)), R.resize_nearest(a, nw, nh))


def _c_plan(iw, ih, ow, oh, filt):
    from unetk import _native as nat
    nbytes = int(nat.fn['unetk_resize_plan_build'](iw, ih, ow, oh, filt, None, 0))
    assert nbytes > 0
    buf = np.zeros(nbytes, dtype=np.uint8)
    assert int(nat.fn['unetk_resize_plan_build'](iw, ih, ow, oh, filt, buf.ctypes.data, nbytes)) == nbytes
    return buf, buf[:72].view(np.int32)


@pytest.mark.parametrize('iw,ih,ow,oh', [(1918, 1280, 959, 640), (61, 40, 30, 20), (53, 37, 15, 11), (96, 64, 96, 32),
                                         (90, 90, 9, 9), (70, 50, 52, 37)])
def test_c_plan_builder_equals_oracle(iw, ih, ow, oh):
    """Host logic of the C ABI (no GPU): windows, 22-bit coefficients and NEAREST tables, bit for bit."""
    buf, hdr = _c_plan(iw, ih, ow, oh, 3)
    ksh, ksv, obh, och, obv, ocv = hdr[8], hdr[9], hdr[12], hdr[13], hdr[14], hdr[15]
    bh, kh = R.precompute_coeffs(iw, ow)
    bv, kv = R.precompute_coeffs(ih, oh)
    assert (ksh, ksv) == (kh.shape[1], kv.shape[1])
    assert np.array_equal(buf[obh:obh + ow * 8].view(np.int32).reshape(ow, 2), bh)
    assert np.array_equal(buf[och:och + ow * ksh * 4].view(np.int32).reshape(ow, ksh), kh)
    assert np.array_equal(buf[obv:obv + oh * 8].view(np.int32).reshape(oh, 2), bv)
    assert np.array_equal(buf[ocv:ocv + oh * ksv * 4].view(np.int32).reshape(oh, ksv), kv)
    buf, hdr = _c_plan(iw, ih, ow, oh, 0)
    assert np.array_equal(buf[hdr[12]:hdr[12] + ow * 4].view(np.int32), R.nearest_table(iw, ow))
    assert np.array_equal(buf[hdr[14]:hdr[14] + oh * 4].view(np.int32), R.nearest_table(ih, oh))


def test_c_plan_builder_rejects_bad_arguments():
    from unetk import _native as nat
    f = nat.fn['unetk_resize_plan_build']
    assert int(f(0, 10, 5, 5, 3, None, 0)) < 0
    assert int(f(10, 10, 5, 5, 2, None, 0)) < 0             # BILINEAR etc.: not what the reference calls
    need = int(f(10, 10, 5, 5, 3, None, 0))
    small = np.zeros(need - 16, dtype=np.uint8)
    assert int(f(10, 10, 5, 5, 3, small.ctypes.data, need - 16)) < 0


# ------------------------------------------------------------------------------------------------ GPU
@pytest.mark.gpu
@pytest.mark.parametrize('tag', IMAGE_CASES)
def test_gpu_image_matches_reference_fixture(tag):
    from utils.data_loading import BasicDataset
    a, sc = _case(tag)
    out = BasicDataset.preprocess_gpu(None, torch.from_numpy(a).cuda(), sc, is_mask=False)
    ref = GOLD[tag + '.out']
    assert out.dtype == torch.float32 and tuple(out.shape) == ref.shape
    assert np.array_equal(out.cpu().numpy(), ref)


@pytest.mark.gpu
def test_gpu_dark_image_and_batch_flags_are_per_image():
    """data_loading.py:88 decides /255 per image: a batch of [dark, bright] must keep the first one undivided."""
    from unetk import preprocess as P
    dark = GOLD['dark.in']
    out = P.preprocess_image(torch.from_numpy(dark).cuda(), 0.5)
    assert np.array_equal(out.cpu().numpy(), GOLD['dark.out'])
    bright = W.photo('pre.bright', dark.shape[0], dark.shape[1], 3)
    both = P.preprocess_image(torch.from_numpy(np.stack([dark, bright])).cuda(), 0.5)
    assert both.shape[0] == 2 and both.is_contiguous(memory_format=torch.channels_last)
    assert np.array_equal(both[0].cpu().numpy(), GOLD['dark.out'])
    assert np.array_equal(both[1].cpu().numpy(), R.preprocess(None, bright, 0.5, False).astype(np.float32))


@pytest.mark.gpu
@pytest.mark.parametrize('tag', ['mask2', 'mask3', 'maskrgb'])
def test_gpu_mask_matches_reference_fixture(tag):
    from utils.data_loading import BasicDataset
    values, sc = _masks()[tag]
    out = BasicDataset.preprocess_gpu(values, torch.from_numpy(GOLD[tag + '.in']).cuda(), sc, is_mask=True)
    assert out.dtype == torch.int64 and np.array_equal(out.cpu().numpy(), GOLD[tag + '.out'])


@pytest.mark.gpu
def test_gpu_carvana_size_batch_hash_and_pillow():
    """1918x1280 -> 959x640 (BASELINE.json configs[1] input), batch of 2: SHA-256 of the reference's bytes; the second
    frame against Pillow directly when it is importable."""
    from unetk import preprocess as P
    a = W.photo('pre.carvana', 1280, 1918, 3)
    b = np.ascontiguousarray(a[::-1, ::-1])
    out = P.preprocess_image(torch.from_numpy(np.stack([a, b])).cuda(), 0.5)
    assert tuple(out.shape) == (2, 3, 640, 959)
    u8 = torch.round(out.permute(0, 2, 3, 1) * 255.0).to(torch.uint8).cpu().numpy()
    assert np.array_equal(out.permute(0, 2, 3, 1).cpu().numpy(), (u8 / 255.0).astype(np.float32))
    assert hashlib.sha256(np.ascontiguousarray(u8[0]).tobytes()).hexdigest() == str(GOLD['carvana.sha256'])
    try:
        from PIL import Image
        ref = np.asarray(Image.fromarray(b).resize((959, 640), resample=Image.BICUBIC))
    except ImportError:
        ref = R.resize_bicubic(b, 959, 640)
    assert np.array_equal(u8[1], ref)
    mk = (W.int_wave('pre.carvana_mask', (1280, 1918), 2) * 255).astype(np.uint8)
    mo = P.preprocess_mask(torch.from_numpy(mk).cuda(), [0, 255], 0.5)
    assert hashlib.sha256(np.ascontiguousarray(mo.cpu().numpy()).tobytes()).hexdigest() == str(GOLD['carvana_mask.sha256'])


@pytest.mark.gpu
def test_gpu_preprocess_feeds_the_net_like_the_cpu_path():
    """predict.py:21-23: preprocess -> unsqueeze -> net.  The GPU-formatted frame gives the same logits, bit for bit."""
    from unet import UNet
    from utils.data_loading import BasicDataset
    from PIL import Image
    rgb = W.photo('pre.net', 70, 101, 3)
    torch.manual_seed(0)
    net = UNet(3, 2, False).cuda().to(memory_format=torch.channels_last).eval()
    cpu = torch.from_numpy(BasicDataset.preprocess(None, Image.fromarray(rgb), 0.5, is_mask=False)).float()
    gpu = BasicDataset.preprocess_gpu(None, torch.from_numpy(rgb).cuda(), 0.5, is_mask=False)
    assert torch.equal(gpu.cpu(), cpu)
    with torch.no_grad():
        l_cpu = net(cpu.unsqueeze(0).cuda())
        l_gpu = net(gpu.unsqueeze(0))
    assert torch.equal(l_cpu, l_gpu)


@pytest.mark.gpu
def test_gpu_preprocess_argument_errors():
    from unetk import preprocess as P
    with pytest.raises(RuntimeError):
        P.preprocess_image(torch.zeros((8, 8, 3)).cuda(), 0.5)                       # not uint8
    with pytest.raises(RuntimeError):
        P.preprocess_image(torch.zeros((8, 8, 4), dtype=torch.uint8).cuda(), 0.5)    # RGBA
    with pytest.raises(RuntimeError):
        P.preprocess_image(torch.zeros((8, 8, 3), dtype=torch.uint8), 0.5)           # CPU tensor
    with pytest.raises(AssertionError):
        P.preprocess_image(torch.zeros((8, 8, 3), dtype=torch.uint8).cuda(), 0.01)   # data_loading.py:68
