"""BASELINE.json configs[0]: UNet(3,2,bilinear=False) fp32 forward, N=1, x ~ N(0,1) seed 0, [1,3,640,959].

tests/golden/fullsize_cfg0.npz holds the REFERENCE's logits (imported /root/reference/unet run in the build container,
oracle/make_golden_fullsize.py) on a stride-16 grid plus whole-tensor moments, for eval mode and train mode.
  CPU : the oracle (oracle/unet_torch.py) reproduces them             -- bar 1e-4 max-relative (north_star)
  GPU : the HIP path in fp32 reproduces them to the same bar; under bf16 autocast the eval logits stay within 2e-2.
"""
import os

import numpy as np
import pytest
import torch

from conftest import rel_err
from oracle import weights as W

G = np.load(os.path.join(os.path.dirname(__file__), 'golden', 'fullsize_cfg0.npz'))


def _input():
    x = torch.randn((1, 3, 640, 959), generator=torch.Generator().manual_seed(0))
    chk = np.array([float(x.double().sum()), float(x.double().abs().sum())])
    assert np.allclose(chk, G['x_checksum'], rtol=1e-12), 'torch CPU generator changed: regenerate the fixture'
    return x.contiguous(memory_format=torch.channels_last)


def _load(net):
    sd = net.state_dict()
    net.load_state_dict({k: torch.from_numpy(W.tensor_for_key(k, tuple(v.shape), str(v.dtype))).to(v.dtype)
                         for k, v in sd.items()}, strict=True)
    return net


def _check(y, mode, tol):
    y = y.detach().float().cpu()
    assert tuple(y.shape) == (1, 2, 640, 959)
    assert rel_err(y[:, :, ::16, ::16].numpy(), G[mode + '.sub']) < tol
    mean, std, amax = G[mode + '.moments']
    assert abs(float(y.double().mean()) - mean) < tol * amax
    assert abs(float(y.double().std()) - std) < tol * amax
    assert abs(float(y.abs().max()) - amax) < 10 * tol * amax


@pytest.mark.parametrize('mode', ['eval', 'train'])
def test_oracle_reproduces_reference_at_full_size(mode):
    from oracle.unet_torch import TorchUNet
    torch.set_num_threads(min(8, os.cpu_count() or 1))
    net = _load(TorchUNet(3, 2, False))
    getattr(net, mode)()
    with torch.no_grad():
        _check(net(_input()), mode, 1e-4)


@pytest.mark.gpu
@pytest.mark.parametrize('mode', ['eval', 'train'])
def test_hip_fp32_reproduces_reference_at_full_size(mode):
    from unet import UNet
    net = _load(UNet(3, 2, False)).cuda().to(memory_format=torch.channels_last)
    getattr(net, mode)()
    with torch.no_grad():
        y = net(_input().cuda())
    assert y.dtype == torch.float32
    _check(y, mode, 1e-4)


@pytest.mark.gpu
def test_hip_bf16_eval_at_full_size_within_bf16_bar():
    from unet import UNet
    net = _load(UNet(3, 2, False)).cuda().to(memory_format=torch.channels_last).eval()
    with torch.no_grad(), torch.autocast('cuda', dtype=torch.bfloat16):
        y = net(_input().cuda())
    assert y.dtype == torch.bfloat16
    _check(y, 'eval', 2e-2)
