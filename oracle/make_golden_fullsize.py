"""TEST INFRASTRUCTURE ONLY -- tests/golden/fullsize_cfg0.npz: BASELINE.json configs[0] run by the imported reference.

    python oracle/make_golden_fullsize.py          (build container: /root/reference must exist)

`UNet(3, 2, bilinear=False)` from /root/reference/unet with the closed-form weights of oracle/weights.py, fp32, CPU,
on x ~ N(0,1) (torch CPU generator, seed 0) of shape [1, 3, 640, 959]:
  * eval-mode forward (running statistics of the closed-form state_dict) and
  * train-mode forward (batch statistics),
each stored as the logits sampled on a stride-16 grid plus whole-tensor moments (the full tensors are 4.9 MB each).
Fixtures are data only; the tests regenerate weights and input from the same formulas.
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

torch.set_num_threads(8)
net = UNet(3, 2, False)
sd = net.state_dict()
net.load_state_dict({k: torch.from_numpy(W.tensor_for_key(k, tuple(v.shape), str(v.dtype))).to(v.dtype)
                     for k, v in sd.items()}, strict=True)
x = torch.randn((1, 3, 640, 959), generator=torch.Generator().manual_seed(0))
out = {'x_checksum': np.array([float(x.double().sum()), float(x.double().abs().sum())])}
with torch.no_grad():
    for mode in ('eval', 'train'):
        getattr(net, mode)()
        y = net(x.contiguous(memory_format=torch.channels_last))
        out[mode + '.sub'] = y[:, :, ::16, ::16].contiguous().numpy()
        out[mode + '.moments'] = np.array([float(y.double().mean()), float(y.double().std()), float(y.abs().max())])
        print(mode, out[mode + '.moments'])
np.savez_compressed(os.path.join(ROOT, 'tests', 'golden', 'fullsize_cfg0.npz'), **out)
print('done')
