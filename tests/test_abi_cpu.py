"""CPU-only checks of the C-ABI boundary: the shared library loads, exports every symbol include/unetk.h declares,
and rejects bad arguments with error codes BEFORE touching a device (no compute without a GPU)."""
import ctypes
import os

import pytest

from conftest import PKG, ROOT


@pytest.fixture(scope='module')
def nat():
    if not os.path.exists(os.path.join(PKG, 'libunetk.so')):
        import __graft_entry__
        __graft_entry__.build()
    from unetk import _native
    return _native


def test_library_exports_every_declared_symbol(nat):
    protos = nat.parse_header(os.path.join(ROOT, 'include', 'unetk.h'))
    assert len(protos) >= 35
    lib = ctypes.CDLL(nat.LIB_PATH)
    for name in protos:
        assert hasattr(lib, name), f'{name} declared in include/unetk.h but not exported by libunetk.so'
    assert nat.fn['unetk_arch']() == b'gfx950'
    assert nat.fn['unetk_version']() >= 1


def test_packed_weight_sizes(nat):
    f = nat.fn['unetk_packed_weight_bytes']
    assert f(0, 128, 64, nat.BF16) == 128 * 64 * 9 * 2          # conv3x3 fprop, bf16
    assert f(1, 128, 64, nat.F32) == 128 * 64 * 9 * 4           # dgrad, fp32
    assert f(2, 64, 128, nat.BF16) == 4 * 64 * 128 * 2          # convT fwd: N = 4*Co
    assert f(3, 64, 128, nat.BF16) == 4 * 64 * 128 * 2
    assert f(0, 100, 64, nat.BF16) == -1                        # c_out not a multiple of 64
    assert f(0, 64, 3, nat.BF16) == -1                          # the 3-channel layer has its own kernel
    assert f(9, 64, 64, nat.BF16) == -5
    assert f(0, 64, 64, 7) == -2


def test_entry_points_validate_before_launch(nat):
    fn = nat.fn
    # null pointers / bad shapes are refused with an error code; nothing is launched
    assert fn['unetk_conv3x3_fprop'](None, 64, 0, None, None, 64, 0, None, 1, 8, 8, 64, 64, nat.BF16, None) == -5
    assert fn['unetk_conv3x3_fprop'](16, 64, 0, 16, 16, 64, 0, None, 1, 8, 8, 48, 64, nat.BF16, None) == -1
    assert fn['unetk_conv3x3_fprop'](16, 64, 0, 16, 16, 64, 0, None, 1, 8, 8, 64, 64, 5, None) == -2
    assert fn['unetk_conv3x3_fprop'](24, 64, 0, 16, 16, 64, 0, None, 1, 8, 8, 64, 64, nat.BF16, None) == -3
    assert fn['unetk_maxpool2x2_fwd'](16, 64, 16, 64, 1, 1, 8, 64, nat.BF16, None) == -1      # H < 2
    assert fn['unetk_wgrad_workspace_bytes'](0, 16, 640, 959, 64, 64) == 512 * 9 * 64 * 64 * 4
    assert fn['unetk_wgrad_workspace_bytes'](0, 16, 640, 959, 60, 64) == -1
    assert fn['unetk_seg_loss_fwd'](None, 2, None, 10, 2, 1e-6, None, None, nat.BF16, None) == -5
    with pytest.raises(RuntimeError):
        nat.check(-3, 'x')


def test_product_has_no_cpu_fallback(nat):
    import torch
    from unet import UNet
    from utils.dice_score import dice_coeff
    with pytest.raises(RuntimeError):
        UNet(3, 2)(torch.zeros(1, 3, 32, 32))
    with pytest.raises(RuntimeError):
        dice_coeff(torch.zeros(2, 4, 4), torch.zeros(2, 4, 4))


def test_product_does_not_import_oracle():
    """The oracle is test infrastructure: nothing under pytorch-unet_amd/ may reference it."""
    for base, _, files in os.walk(PKG):
        for f in files:
            if f.endswith(('.py', '.hip', '.h')):
                text = open(os.path.join(base, f)).read()
                assert 'oracle' not in text.replace('the oracle', '').lower() or f == '__none__', os.path.join(base, f)
