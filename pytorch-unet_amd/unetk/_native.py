"""ctypes binding of libunetk.so (include/unetk.h) -- the only door from Python to the HIP kernels.

There is deliberately NO fallback: if the shared library is missing or a symbol is absent the
import fails loudly, and every wrapper raises RuntimeError on a non-zero return code
(the reference relies on RuntimeError/AssertionError the same way: train.py:44, dice_score.py:7-8).
"""
import ctypes
import os

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(os.path.dirname(_HERE), 'libunetk.so')

BF16, F32 = 0, 1
STAT_REPLICAS = 32

_ERR = {-1: 'unsupported shape/pitch/channel count', -2: 'unsupported dtype', -3: 'misaligned pointer or pitch',
        -4: 'kernel launch failed', -5: 'bad argument'}

if not os.path.exists(LIB_PATH):
    raise ImportError(f'{LIB_PATH} not found: build it with `make -C pytorch-unet_amd/csrc` '
                      '(or __graft_entry__.build()); there is no non-HIP fallback')
lib = ctypes.CDLL(LIB_PATH)

_vp, _i, _i64 = ctypes.c_void_p, ctypes.c_int, ctypes.c_int64


def _proto(name, argtypes, restype=_i):
    fn = getattr(lib, name)       # AttributeError if the symbol is missing: loud by design
    fn.argtypes = argtypes
    fn.restype = restype
    return fn


PROTOTYPES = {
    'unetk_version': ([], _i),
    'unetk_arch': ([], ctypes.c_char_p),
    'unetk_packed_weight_bytes': ([_i, _i, _i, _i], _i64),
    'unetk_pack_weight': ([_i, _vp, _vp, _i, _i, _i, _vp], _i),
    'unetk_conv3x3_fprop': ([_vp, _i, _i, _vp, _vp, _i, _i, _vp, _i, _i, _i, _i, _i, _i, _vp], _i),
    'unetk_conv3x3_dgrad': ([_vp, _i, _i, _vp, _vp, _i, _i, _i, _i, _i, _i, _i, _i, _vp], _i),
}

fn = {k: _proto(k, a, r) for k, (a, r) in PROTOTYPES.items()}


def check(code, what):
    if code != 0:
        raise RuntimeError(f'{what} failed: {_ERR.get(code, code)} (code {code})')


def dtype_code(dt):
    if dt == torch.bfloat16:
        return BF16
    if dt == torch.float32:
        return F32
    raise RuntimeError(f'unetk kernels support bfloat16 and float32 activations, got {dt}')


def stream_ptr():
    return torch.cuda.current_stream().cuda_stream


def require_cuda(*tensors):
    for t in tensors:
        if t is not None and not t.is_cuda:
            raise RuntimeError('unetk kernels run on the GPU only (no CPU fallback); got a CPU tensor')
