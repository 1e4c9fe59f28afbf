"""TEST INFRASTRUCTURE ONLY -- tests/golden/unet_c2_separated.npz: a whole-net fixture with WELL-SEPARATED logits.

    python oracle/make_golden_separated.py          (build container only: imports /root/reference)

The closed-form random weights of oracle/make_golden.py give two-class logits within ~0.3 of each other, so even the
reference's own bf16 run flips ~0.5 % of the arg-max decisions and north_star's ">= 99.9 % agreement" cannot be
asserted on them.  Here OutConv's bias is offset per mode so that the class gap z1 - z0 sits ~3 sigma from zero
(a Carvana-like minority class of a fraction of a percent to a few percent instead of 50/50): the decision boundary
then runs through the thin tail of the gap distribution and the reference's own bf16 agreement is >= 99.95 %.
The fixture stores, for train mode (batch statistics, fresh net) and eval mode (running statistics after that one
training forward): the bias used, the reference's fp32 logits, and the reference's own bf16 numbers as a yardstick.
Everything else (weights, image) is regenerated from the formulas of oracle/weights.py.
"""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path.insert(0, ROOT)
sys.path.insert(0, '/root/reference')

from oracle import weights as W  # noqa: E402
from unet import UNet  # noqa: E402  (the reference)

TAG, SHAPE = 'unet_c2_separated', (2, 3, 96, 142)
SIGMAS = 3.0


def load_closed_form(module):
    sd = module.state_dict()
    module.load_state_dict({k: torch.from_numpy(W.tensor_for_key(k, tuple(v.shape), str(v.dtype))).to(v.dtype)
                            for k, v in sd.items()}, strict=True)
    return module


def with_bias(net, bias):
    with torch.no_grad():
        net.outc.conv.bias.copy_(torch.from_numpy(bias))
    return net


def main():
    torch.set_num_threads(8)
    img = torch.from_numpy(W.wave(TAG + '.img', SHAPE, 0.5, 0.5))
    base_bias = W.tensor_for_key('outc.conv.bias', (2,), 'torch.float32')
    arrs = {}
    # ---- train mode: fresh net, batch statistics
    net = load_closed_form(UNet(3, 2, False)).train()
    with torch.no_grad():
        z = net(img)
    d = (z[:, 1] - z[:, 0]).flatten()
    bias_train = base_bias.copy()
    bias_train[1] += np.float32(SIGMAS * float(d.std()) - float(d.mean()))
    net_t = with_bias(load_closed_form(UNet(3, 2, False)), bias_train).train()
    with torch.no_grad():
        zt = net_t(img)                                   # also moves net_t's running statistics (one step)
        with torch.autocast('cpu', dtype=torch.bfloat16):
            ztb = with_bias(load_closed_form(UNet(3, 2, False)), bias_train).train()(img).float()
    # ---- eval mode: running statistics after that one training forward
    net_t.eval()
    with torch.no_grad():
        z = net_t(img)
    d = (z[:, 1] - z[:, 0]).flatten()
    bias_eval = bias_train.copy()
    bias_eval[1] += np.float32(SIGMAS * float(d.std()) - float(d.mean()))
    with_bias(net_t, bias_eval)
    with torch.no_grad():
        ze = net_t(img)
        with torch.autocast('cpu', dtype=torch.bfloat16):
            zeb = net_t(img).float()

    def agree(a, b):
        return float((a.argmax(1) == b.argmax(1)).float().mean())

    def rel(a, b):
        return float((a - b).abs().max() / b.abs().max())

    arrs.update(bias_train=bias_train, bias_eval=bias_eval, logits_train=zt.numpy(), logits_eval=ze.numpy(),
                ref_bf16_agree_train=np.array(agree(ztb, zt)), ref_bf16_agree_eval=np.array(agree(zeb, ze)),
                ref_bf16_rel_train=np.array(rel(ztb, zt)), ref_bf16_rel_eval=np.array(rel(zeb, ze)),
                minority_train=np.array(float((zt.argmax(1) == 0).float().mean())),
                minority_eval=np.array(float((ze.argmax(1) == 0).float().mean())))
    for k in ('ref_bf16_agree_train', 'ref_bf16_agree_eval', 'ref_bf16_rel_train', 'ref_bf16_rel_eval', 'minority_train',
              'minority_eval'):
        print(k, float(arrs[k]))
    assert arrs['ref_bf16_agree_train'] >= 0.9995 and arrs['ref_bf16_agree_eval'] >= 0.9995
    assert 0 < arrs['minority_train'] < 0.1 and 0 < arrs['minority_eval'] < 0.1     # both classes are predicted
    np.savez_compressed(os.path.join(ROOT, 'tests', 'golden', TAG + '.npz'), **arrs)


if __name__ == '__main__':
    main()
