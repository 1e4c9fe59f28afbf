"""ctypes binding of libunetk.so (include/unetk.h) -- the only door from Python to the HIP kernels.

There is deliberately NO fallback: if the shared library is missing or a symbol is absent the
import fails loudly, and every wrapper raises RuntimeError on a non-zero return code
(the reference relies on RuntimeError/AssertionError the same way: train.py:44, dice_score.py:7-8).
"""
import ctypes
import os

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get('UNETK_LIB') or os.path.join(os.path.dirname(_HERE), 'libunetk.so')   # UNETK_LIB: perf A/B builds

BF16, F32, F16 = 0, 1, 2
STAT_REPLICAS = 32

_ERR = {-1: 'unsupported shape/pitch/channel count', -2: 'unsupported dtype', -3: 'misaligned pointer or pitch',
        -4: 'kernel launch failed', -5: 'bad argument',
        -6: 'problem exceeds the fixed capacity of this no-allocation entry point'}

if not os.path.exists(LIB_PATH):
    raise ImportError(f'{LIB_PATH} not found: build it with `make -C pytorch-unet_amd/csrc` '
                      '(or __graft_entry__.build()); there is no non-HIP fallback')
lib = ctypes.CDLL(LIB_PATH)

HEADER_PATH = os.path.join(os.path.dirname(os.path.dirname(_HERE)), 'include', 'unetk.h')

_CTYPE = {'int': ctypes.c_int, 'int64_t': ctypes.c_int64, 'float': ctypes.c_float}


def parse_header(path=HEADER_PATH):
    """Read the `extern "C"` declarations of include/unetk.h -> {name: (argtypes, restype)}.

    The header is the single source of truth for the ABI: the binding is generated from it, so a
    signature change cannot silently desynchronise Python from the library."""
    import re
    text = open(path).read()
    text = re.sub(r'/\*.*?\*/', ' ', text, flags=re.S)
    text = re.sub(r'^\s*#.*$', ' ', text, flags=re.M)
    protos = {}
    for m in re.finditer(r'(const\s+char\s*\*|int64_t|int)\s+(unetk_\w+)\s*\(([^)]*)\)\s*;', text):
        ret, name, args = m.group(1), m.group(2), m.group(3).strip()
        argtypes = []
        if args and args != 'void':
            for a in args.split(','):
                a = a.strip()
                if '*' in a:
                    argtypes.append(ctypes.c_void_p)
                else:
                    argtypes.append(_CTYPE[a.split()[-2] if len(a.split()) > 1 else a])
        restype = ctypes.c_char_p if 'char' in ret else _CTYPE[ret]
        protos[name] = (argtypes, restype)
    return protos


PROTOTYPES = parse_header()


def _bind(name, argtypes, restype):
    f = getattr(lib, name)       # AttributeError if the symbol is missing: loud by design
    f.argtypes = argtypes
    f.restype = restype
    return f


fn = {k: _bind(k, a, r) for k, (a, r) in PROTOTYPES.items()}


def check(code, what):
    if code != 0:
        raise RuntimeError(f'{what} failed: {_ERR.get(code, code)} (code {code})')


def dtype_code(dt):
    if dt == torch.bfloat16:
        return BF16
    if dt == torch.float32:
        return F32
    if dt == torch.float16:
        return F16
    raise RuntimeError(f'unetk kernels support bfloat16, float16 and float32 activations, got {dt}')


def stream_ptr():
    return torch.cuda.current_stream().cuda_stream


def require_cuda(*tensors):
    for t in tensors:
        if t is not None and not t.is_cuda:
            raise RuntimeError('unetk kernels run on the GPU only (no CPU fallback); got a CPU tensor')


# ---------------------------------------------------------------------------------------------- kernel timing
class KernelTimer:
    """Brackets every C-ABI launch with HIP events on the stream it is launched on (torch's current stream) and
    attributes algorithmic FLOPs to the MFMA convolutions.  Used by bench.py for the `roofline` object; off by
    default (zero overhead: the dict holds the raw ctypes functions)."""

    def __init__(self):
        self.records = []      # (name, start_event, end_event, flops, device kernel, conv bytes, call arguments)
        self._saved = None
        self._ms = None

    @staticmethod
    def _flops(name, a):
        # argument positions follow include/unetk.h
        if name in ('unetk_conv3x3_fprop', 'unetk_conv3x3_dgrad_sums'):
            return 2.0 * a[8] * a[9] * a[10] * 9 * a[11] * a[12]
        if name == 'unetk_conv3x3_fprop_bn_relu':
            return 2.0 * a[11] * a[12] * a[13] * 9 * a[14] * a[15]
        if name == 'unetk_conv3x3_dgrad':
            return 2.0 * a[7] * a[8] * a[9] * 9 * a[10] * a[11]
        if name == 'unetk_conv3x3_dgrad_bn_reduce':
            return 2.0 * a[14] * a[15] * a[16] * 9 * a[17] * a[18]
        if name == 'unetk_conv3x3_wgrad':
            return 2.0 * a[7] * a[8] * a[9] * 9 * a[10] * a[11]
        if name == 'unetk_conv3x3_first_fprop':
            return 2.0 * a[6] * a[7] * a[8] * 9 * a[9] * a[10]
        if name == 'unetk_conv3x3_first_fprop_bn_relu_train':
            return 2.0 * a[9] * a[10] * a[11] * 9 * a[12] * a[13]
        if name == 'unetk_conv3x3_first_wgrad':
            return 2.0 * a[7] * a[8] * a[9] * 9 * a[10] * a[11]
        if name == 'unetk_conv3x3_first_wgrad_bn':
            return 2.0 * a[13] * a[14] * a[15] * 9 * a[16] * a[17]
        if name == 'unetk_convT2x2_fwd':
            return 2.0 * a[7] * a[8] * a[9] * 4 * a[10] * a[11]
        if name == 'unetk_convT2x2_dgrad':
            return 2.0 * a[6] * a[7] * a[8] * 4 * a[9] * a[10]
        if name == 'unetk_convT2x2_dgrad_bn_reduce':
            return 2.0 * a[13] * a[14] * a[15] * 4 * a[16] * a[17]
        if name == 'unetk_convT2x2_wgrad':
            return 2.0 * a[8] * a[9] * a[10] * 4 * a[11] * a[12]
        return 0.0

    @staticmethod
    def _device_kernel(name, a):
        """Which conv3x3 kernel serves a launch (csrc/conv_config.h: conv3_use_k32, conv3_k32_nb16): every bf16 launch
        runs on v_mfma_f32_16x16x32_bf16 -- conv3x3_k32_kernel (128-channel n-tiles) when GEMM-N is a multiple of 128,
        else conv3x3_k32n64_kernel (64-channel n-tile, epilogue under the next tile; the inference instantiations of that
        n-tile are conv3x3_k32_kernel<4>); fp32 runs conv3x3_db_kernel (v_mfma_f32_32x32x2_f32)."""
        pos = {'unetk_conv3x3_fprop': (12, 13), 'unetk_conv3x3_dgrad_sums': (11, 13), 'unetk_conv3x3_dgrad': (10, 12),
               'unetk_conv3x3_dgrad_bn_reduce': (17, 19), 'unetk_conv3x3_fprop_bn_relu': (15, 16)}.get(name)
        if pos is None:
            return None
        n_gemm, dtype = a[pos[0]], a[pos[1]]
        if dtype == F32:
            return 'conv3x3_db_kernel'
        return 'conv3x3_k32_kernel' if n_gemm % 128 == 0 else 'conv3x3_k32n64_kernel'

    @staticmethod
    def _conv_bytes(name, a):
        """Algorithmic bytes of a conv3x3 fprop/dgrad launch: input + output tensor + weights, once each."""
        pos = {'unetk_conv3x3_fprop': (8, 13), 'unetk_conv3x3_dgrad_sums': (8, 13), 'unetk_conv3x3_dgrad': (7, 12),
               'unetk_conv3x3_dgrad_bn_reduce': (14, 19), 'unetk_conv3x3_fprop_bn_relu': (11, 16)}.get(name)
        if pos is None:
            return 0.0
        n, h, w, c_in, c_out = a[pos[0]:pos[0] + 5]
        es = 4 if a[pos[1]] == F32 else 2
        return float(es) * (n * h * w * (c_in + c_out) + 9 * c_in * c_out)

    # HBM-bound families: name -> (family, algorithmic bytes) from the call's arguments (positions follow
    # include/unetk.h): every tensor the pass must read or write, once, at its element size; per-channel vectors and
    # fp64 accumulators (kilobytes) are left out.
    @staticmethod
    def _hbm(name, a):
        def es(code):
            return 4 if code == F32 else 2
        if name == 'unetk_bn_relu_apply':                     # read z, write act
            return 'bn_apply', 2.0 * a[6] * a[7] * es(a[8])
        if name == 'unetk_bn_relu_apply_pool':                # + the pooled tensor
            n, h, w, c = a[8:12]
            return 'bn_apply', (2.0 * n * h * w + n * (h // 2) * (w // 2)) * c * es(a[12])
        if name == 'unetk_bn_relu_apply_head':                # + the logits
            return 'bn_apply', ((2.0 if a[4] else 1.0) * a[12] + a[8]) * a[11] * es(a[13])   # out may be NULL
        if name == 'unetk_bn_relu_bwd_reduce':                # read dout, z
            return 'bn_bwd_reduce', 2.0 * a[9] * a[10] * es(a[11])
        if name == 'unetk_bn_relu_bwd_reduce_pool':           # read dskip, z, dpool
            n, h, w, c = a[11:15]
            return 'bn_bwd_reduce', (2.0 * n * h * w + n * (h // 2) * (w // 2)) * c * es(a[15])
        if name == 'unetk_bn_relu_bwd_reduce_head':           # read z, dlogits
            return 'bn_bwd_reduce', (1.0 * a[13] + a[3]) * a[12] * es(a[14])
        if name == 'unetk_bn_relu_bwd_apply':                 # read dout, z, write dz
            return 'bn_bwd_apply', 3.0 * a[10] * a[11] * es(a[12])
        if name == 'unetk_bn_relu_bwd_apply_pool':
            n, h, w, c = a[12:16]
            return 'bn_bwd_apply', (3.0 * n * h * w + n * (h // 2) * (w // 2)) * c * es(a[16])
        if name == 'unetk_bn_relu_bwd_apply_head':            # read z, dlogits, write dz
            return 'bn_bwd_apply', (2.0 * a[13] + a[3]) * a[12] * es(a[14])
        if name == 'unetk_convT2x2_fwd':                      # read x, write 4 taps x c_out, weights once
            n, h, w, ci, co = a[7:12]
            return 'convT_fwd', (1.0 * n * h * w * (ci + 4 * co) + 4.0 * ci * co) * es(a[16])
        if name == 'unetk_convT2x2_dgrad':
            n, h, w, ci, co = a[6:11]
            return 'convT_dgrad', (1.0 * n * h * w * (ci + 4 * co) + 4.0 * ci * co) * es(a[15])
        if name == 'unetk_convT2x2_dgrad_bn_reduce':          # + one read of the layer below's pre-BN tensor
            n, h, w, ci, co = a[13:18]
            return 'convT_dgrad', (1.0 * n * h * w * (2 * ci + 4 * co) + 4.0 * ci * co) * es(a[22])
        if name == 'unetk_convT2x2_wgrad':                    # read x and the up-sampled gradient, write fp32 dw
            n, h, w, ci, co = a[8:13]
            return 'convT_wgrad', 1.0 * n * h * w * (ci + 4 * co) * es(a[18]) + 16.0 * ci * co
        if name == 'unetk_conv3x3_first_fprop':               # read the fp32 image, write c_out channels (y may be NULL)
            n, h, w, ci, co = a[6:11]
            return 'first_layer', 1.0 * n * h * w * (4 * ci + (co * es(a[11]) if a[3] else 0))
        if name == 'unetk_conv3x3_first_fprop_bn_relu_train':  # read the image, write y and the activation
            n, h, w, ci, co = a[9:14]
            return 'first_layer', 1.0 * n * h * w * (4 * ci + 2 * co * es(a[14]))
        if name == 'unetk_conv3x3_first_wgrad':               # read the image and dy
            n, h, w, ci, co = a[7:12]
            return 'first_layer', 1.0 * n * h * w * (4 * ci + co * es(a[12]))
        if name == 'unetk_conv3x3_first_wgrad_bn':            # read the image, dout and z
            n, h, w, ci, co = a[13:18]
            return 'first_layer', 1.0 * n * h * w * (4 * ci + 2 * co * es(a[18]))
        if name == 'unetk_seg_loss_fwd':                      # logits + int64 mask
            return 'loss', 1.0 * a[3] * (a[4] * es(a[8]) + 8)
        if name == 'unetk_seg_loss_bwd':                      # logits + mask, write dlogits
            return 'loss', 1.0 * a[3] * (2 * a[4] * es(a[9]) + 8)
        if name == 'unetk_clip_rmsprop_step':                 # norm: read grad; step: read + write param, grad, 2 states
            return 'optimizer', 36.0 * a[4]
        if name == 'unetk_maxpool2x2_fwd':
            n, h, w, c = a[4:8]
            return 'pool_upsample', (1.0 * n * h * w + n * (h // 2) * (w // 2)) * c * es(a[8])
        if name == 'unetk_bilinear2x_fwd':                    # read low-res, write the 2x window
            n, h, w, c = a[4:8]
            return 'pool_upsample', 5.0 * n * h * w * c * es(a[12])
        if name == 'unetk_bilinear2x_bwd':
            n, h, w, c = a[4:8]
            return 'pool_upsample', 5.0 * n * h * w * c * es(a[12])
        return None, 0.0

    def durations(self):
        """Milliseconds between the two events of every record -- call after torch.cuda.synchronize()."""
        if self._ms is None:
            self._ms = [r[1].elapsed_time(r[2]) for r in self.records]
        return self._ms

    def take_median(self, others):
        """Replace, launch by launch, this pass's duration by the median over this and the other passes (same launch
        sequence required; returns False and changes nothing otherwise).  An event pair also spans any time the GPU
        waits for the HOST to submit the launch between them: one hiccup of the launching thread in one pass must not
        be booked as kernel time, and a median -- unlike a minimum -- does not bias the average downwards."""
        names = [r[0] for r in self.records]
        if any([r[0] for r in o.records] != names for o in others):
            return False
        cols = [self.durations()] + [o.durations() for o in others]
        self._ms = [sorted(v)[len(v) // 2] for v in zip(*cols)]
        return True

    def by_hbm_family(self):
        """{family: (launches, total_ms, total_algorithmic_bytes)} over the HBM-bound launches."""
        out = {}
        for rec, dur in zip(self.records, self.durations()):
            fam, nbytes = self._hbm(rec[0], rec[6])
            if fam is not None:
                n, ms, b = out.get(fam, (0, 0.0, 0.0))
                out[fam] = (n + 1, ms + dur, b + nbytes)
        return out

    def by_device_kernel(self):
        """{device kernel: (launches, total_ms, total_flops, total_algorithmic_bytes)} over the conv3x3 fprop/dgrad
        launches."""
        out = {}
        for (name, s, e, fl, kern, nbytes, _), dur in zip(self.records, self.durations()):
            if kern is not None:
                n, ms, f, b = out.get(kern, (0, 0.0, 0.0, 0.0))
                out[kern] = (n + 1, ms + dur, f + fl, b + nbytes)
        return out

    def __enter__(self):
        self._saved = dict(fn)
        for name, raw in self._saved.items():
            if PROTOTYPES[name][1] is not ctypes.c_int or not PROTOTYPES[name][0] or name.endswith('_is_fused'):
                continue                                   # size / capability queries launch nothing

            def timed(*args, _raw=raw, _name=name):
                s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                s.record()
                rc = _raw(*args)
                e.record()
                self.records.append((_name, s, e, self._flops(_name, args), self._device_kernel(_name, args),
                                     self._conv_bytes(_name, args), args))
                return rc
            fn[name] = timed
        return self

    def __exit__(self, *exc):
        fn.update(self._saved)
        return False

    def summary(self):
        """{name: (launches, total_ms, total_flops)} -- call after torch.cuda.synchronize()."""
        out = {}
        for (name, s, e, fl, _, _, _), dur in zip(self.records, self.durations()):
            n, ms, f = out.get(name, (0, 0.0, 0.0))
            out[name] = (n + 1, ms + dur, f + fl)
        return out
