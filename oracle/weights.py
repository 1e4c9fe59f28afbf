"""TEST INFRASTRUCTURE ONLY -- closed-form, RNG-free tensors for parity tests.

A full fp32 UNet checkpoint is 124 MB, too big to commit, so goldens are generated from
weights every machine can rebuild bit-identically: each tensor is `scale*sin(a*i+b)+offset`
over its flat index i, with (a, b) derived from a CRC of the tensor's state_dict key
("Whole-net weights").  The same generator fills inputs and masks.
"""
import zlib

import numpy as np


def _ab(name: str):
    h = zlib.crc32(name.encode())
    a = 0.37 + (h & 0xFFFF) / 65536.0          # in [0.37, 1.37)
    b = ((h >> 16) & 0xFFFF) / 65536.0 * 6.0   # in [0, 6)
    return a, b


def wave(name: str, shape, scale=1.0, offset=0.0, dtype=np.float32):
    """scale*sin(a*i+b)+offset over the flat (C-order) index, fp64 math, cast once."""
    n = int(np.prod(shape)) if len(shape) else 1
    a, b = _ab(name)
    i = np.arange(n, dtype=np.float64)
    v = scale * np.sin(a * i + b) + offset
    return v.astype(dtype).reshape(shape)


def int_wave(name: str, shape, n_values: int):
    """Deterministic integer field in [0, n_values) (segmentation masks)."""
    n = int(np.prod(shape))
    a, b = _ab(name)
    i = np.arange(n, dtype=np.float64)
    v = np.floor((np.sin(a * 0.05 * i + b) * 0.5 + 0.5) * n_values * 0.999999)
    return v.astype(np.int64).reshape(shape)


def tensor_for_key(key: str, shape, dtype_name: str):
    """Value for one state_dict entry, by the role its key names (row S of SURVEY 8a)."""
    leaf = key.rsplit('.', 1)[-1]
    if leaf == 'num_batches_tracked':
        return np.zeros(shape, dtype=np.int64)
    if leaf == 'running_mean':
        return wave(key, shape, 0.1)
    if leaf == 'running_var':
        return wave(key, shape, 0.2, 1.0)
    if len(shape) == 4:                       # conv / transposed-conv weight
        fan_in = shape[1] * shape[2] * shape[3]
        return wave(key, shape, float(np.sqrt(3.0 / fan_in)))
    if leaf == 'weight':                      # BatchNorm gamma
        return wave(key, shape, 0.25, 1.0)
    return wave(key, shape, 0.1)              # biases (BN beta, convT, 1x1 head)


def fill_state_dict(template: dict) -> dict:
    """template: key -> (shape tuple, dtype name).  Returns key -> ndarray."""
    return {k: tensor_for_key(k, tuple(s), d) for k, (s, d) in template.items()}
