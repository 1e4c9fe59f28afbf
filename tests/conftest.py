import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(ROOT, 'pytorch-unet_amd')
for p in (ROOT, PKG):
    if p not in sys.path:
        sys.path.insert(0, p)

GOLDEN = os.path.join(ROOT, 'tests', 'golden')


def pytest_configure(config):
    config.addinivalue_line('markers', 'gpu: needs a real MI355X (run with -m gpu on the GPU box)')


def golden(name):
    return np.load(os.path.join(GOLDEN, name + '.npz'))


@pytest.fixture(scope='session')
def golden_loader():
    return golden


def rel_err(a, b):
    """max |a-b| / max |b| -- the 'relative' of BASELINE.json's tolerances."""
    a = np.asarray(a, dtype=np.float64)
    b = np.asarray(b, dtype=np.float64)
    return float(np.abs(a - b).max() / max(np.abs(b).max(), 1e-30))


def l2_err(a, b):
    """||a-b||_2 / ||b||_2.  Used for gradients: a ReLU whose pre-activation is within rounding of zero may flip
    between two correct fp32 evaluations (the reference's own fp32 and fp64 runs differ this way), which moves ONE
    BatchNorm channel's gradients by O(10 %) and would dominate a max-norm while the tensor as a whole agrees."""
    a = np.asarray(a, dtype=np.float64)
    b = np.asarray(b, dtype=np.float64)
    return float(np.linalg.norm(a - b) / max(np.linalg.norm(b), 1e-30))


# whole-network golden cases (tag, n_classes, bilinear, input shape): keep in sync with oracle/make_golden.py
UNET_CASES = [('unet_c2_convT', 2, False, (2, 3, 96, 142)), ('unet_c2_bilinear', 2, True, (2, 3, 96, 142)),
              ('unet_c21_convT', 21, False, (2, 3, 64, 80)), ('unet_c1_convT', 1, False, (2, 3, 112, 99))]
