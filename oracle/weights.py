"""TEST INFRASTRUCTURE ONLY -- closed-form, RNG-free tensors for parity tests.

A full fp32 UNet checkpoint is 124 MB, too big to commit, so goldens are generated from
weights every machine can rebuild bit-identically: each tensor is `scale*u+offset` with
u ~ U[-1,1) hashed (SplitMix64, integer arithmetic only) from a CRC of the tensor's state_dict
key and the flat element index ("Whole-net weights").  The same
generator fills inputs and masks.
"""
import zlib

import numpy as np


def _seed(name: str) -> np.uint64:
    return np.uint64(zlib.crc32(name.encode())) * np.uint64(0x9E3779B97F4A7C15)


def _splitmix64(x):
    """SplitMix64 finaliser on uint64 arrays: pure integer arithmetic, identical on every machine."""
    with np.errstate(over='ignore'):
        x = (x + np.uint64(0x9E3779B97F4A7C15))
        x = (x ^ (x >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)
        x = (x ^ (x >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)
        return x ^ (x >> np.uint64(31))


def _uniform(name: str, n: int):
    """n reproducible uniforms in [-1, 1): hash(key, index) -> 53-bit mantissa.  (A plain sinusoid over the
    flat index makes conv outputs nearly cancel, which leaves BatchNorm channels with variance at rounding
    level and lets even the reference's fp32 and fp64 runs disagree; hashed values are well conditioned.)"""
    with np.errstate(over='ignore'):
        idx = np.arange(n, dtype=np.uint64) + _seed(name)
    bits = _splitmix64(idx) >> np.uint64(11)
    return bits.astype(np.float64) * (2.0 / (1 << 53)) - 1.0


def wave(name: str, shape, scale=1.0, offset=0.0, dtype=np.float32):
    """scale*u + offset with u ~ U[-1,1) hashed from (name, flat index); fp64 math, cast once."""
    n = int(np.prod(shape)) if len(shape) else 1
    return (scale * _uniform(name, n) + offset).astype(dtype).reshape(shape)


def int_wave(name: str, shape, n_values: int):
    """Deterministic integer field in [0, n_values) made of 4x4 blocks (segmentation-mask like)."""
    shape = tuple(shape)
    coarse = tuple(shape[:-2]) + ((shape[-2] + 3) // 4, (shape[-1] + 3) // 4)
    u = (_uniform(name, int(np.prod(coarse))) * 0.5 + 0.5).reshape(coarse)
    v = np.floor(u * n_values * 0.999999).astype(np.int64)
    v = np.repeat(np.repeat(v, 4, axis=-2), 4, axis=-1)
    return np.ascontiguousarray(v[..., :shape[-2], :shape[-1]])


def photo(tag: str, h: int, w: int, c: int):
    """uint8 test image [h,w,c] ([h,w] for c == 1): 60 % hashed noise (worst case for the cubic's negative lobes,
    exercises the 8-bit clipping) + 40 % integer triangle wave.  Integer / IEEE mult-add only: identical everywhere."""
    shape = (h, w, c) if c > 1 else (h, w)
    noise = wave(tag + '.noise', shape, 0.5, 0.5, dtype=np.float64)
    yy, xx = np.mgrid[0:h, 0:w]
    smooth = np.abs(((xx * 9 + yy * 5) % 510) - 255).astype(np.float64)
    if c > 1:
        smooth = smooth[..., None]
    return np.clip(0.6 * 255.0 * noise + 0.4 * smooth, 0, 255).astype(np.uint8)


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
