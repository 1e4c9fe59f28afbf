"""TEST INFRASTRUCTURE ONLY -- tests/golden/preprocess.npz from the reference's own BasicDataset.preprocess.

Run in the build container (where /root/reference exists):
    python oracle/make_golden_preprocess.py
Imports `/root/reference/utils/data_loading.py` (PIL resize + conversion, :65-91) and runs its staticmethod on
closed-form "photos" and masks (oracle/weights.py: the tests regenerate the inputs).  Small cases store the full
outputs; the Carvana-sized case a SHA-256 of the resized bytes (the work is integer: equal hashes = equal results)
plus a few rows.  Fixtures are data only.
"""
import hashlib
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path.insert(0, ROOT)
sys.path.insert(0, '/root/reference')

from PIL import Image  # noqa: E402
from oracle import weights as W  # noqa: E402
from utils.data_loading import BasicDataset as RefDataset  # noqa: E402  (the reference)

OUT = os.path.join(ROOT, 'tests', 'golden')
pre = {}

IMAGE_CASES = {'rgb_half': (40, 61, 3, 0.5), 'rgb_third': (37, 53, 3, 0.3), 'rgb_same': (32, 48, 3, 1.0),
               'gray_half': (33, 47, 1, 0.5), 'rgb_075': (50, 70, 3, 0.75), 'rgb_tiny': (90, 90, 3, 0.11)}
for tag, (h, w, c, sc) in IMAGE_CASES.items():
    a = W.photo('pre.' + tag, h, w, c)
    out = RefDataset.preprocess(None, Image.fromarray(a), sc, is_mask=False)
    pre[tag + '.out'] = torch.as_tensor(out.copy()).float().numpy()        # data_loading.py:110
    pre[tag + '.cfg'] = np.array([h, w, c, sc], dtype=np.float64)
# an image whose values never exceed 1: data_loading.py:88 skips the /255
a = (W.photo('pre.dark', 24, 36, 3) > 200).astype(np.uint8)
pre['dark.out'] = torch.as_tensor(RefDataset.preprocess(None, Image.fromarray(a), 0.5, is_mask=False).copy()).float().numpy()
pre['dark.in'] = a
# masks: binary {0,255} (Carvana), three grey levels, RGB colour classes
m = (W.int_wave('pre.mask2', (40, 61), 2) * 255).astype(np.uint8)
pre['mask2.in'] = m
pre['mask2.out'] = RefDataset.preprocess([0, 255], Image.fromarray(m), 0.5, is_mask=True)
lv = np.array([0, 7, 200], dtype=np.uint8)
m = lv[W.int_wave('pre.mask3', (37, 53), 3)]
pre['mask3.in'] = m
pre['mask3.out'] = RefDataset.preprocess([0, 7, 200], Image.fromarray(m), 0.3, is_mask=True)
cols = np.array([[0, 0, 0], [255, 0, 0], [0, 0, 255], [255, 255, 0]], dtype=np.uint8)
m = cols[W.int_wave('pre.maskrgb', (30, 44), 4)]
pre['maskrgb.in'] = m
pre['maskrgb.values'] = cols
pre['maskrgb.out'] = RefDataset.preprocess(cols.tolist(), Image.fromarray(m), 0.5, is_mask=True)
# Carvana-sized: 1918x1280 -> 959x640 (scale 0.5, BASELINE.json configs[1])
a = W.photo('pre.carvana', 1280, 1918, 3)
out = RefDataset.preprocess(None, Image.fromarray(a), 0.5, is_mask=False)
u8 = np.rint(out * 255.0).astype(np.uint8).transpose(1, 2, 0)            # back to the resized HWC bytes
assert np.array_equal((u8.transpose(2, 0, 1) / 255.0), out)
pre['carvana.sha256'] = np.array(hashlib.sha256(np.ascontiguousarray(u8).tobytes()).hexdigest())
pre['carvana.rows'] = u8[[0, 1, 319, 638, 639]]
mk = (W.int_wave('pre.carvana_mask', (1280, 1918), 2) * 255).astype(np.uint8)
mo = RefDataset.preprocess([0, 255], Image.fromarray(mk), 0.5, is_mask=True)
pre['carvana_mask.sha256'] = np.array(hashlib.sha256(np.ascontiguousarray(mo).tobytes()).hexdigest())
np.savez_compressed(os.path.join(OUT, 'preprocess.npz'), **pre)
print('done', sorted(pre))

