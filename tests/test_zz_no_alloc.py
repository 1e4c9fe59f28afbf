"""unetk_eval_dice_noalloc (include/unetk.h): the validation Dice for callers that cannot allocate.

Called through the C ABI (ctypes, raw device pointers) and compared BIT FOR BIT with unetk_eval_dice -- the entry
point it is a variant of -- and with the NumPy oracle of evaluate.py:31-36.  Two calls in a row share the library's
static scratch array, the second on a smaller input: stale counters of the first call must not leak into it.
This file sorts last on purpose: the round-end `pytest -x` would hide every later test behind a failure here.
"""
import numpy as np
import pytest

from conftest import rel_err

pytestmark = pytest.mark.gpu


def _oracle_dice(logits_nhwc, mask, first_class=1, eps=1e-6):
    """evaluate.py:31-36 restated: one-hot(argmax) vs one-hot(mask), per-(image, class) dice, classes >= first."""
    from oracle import ops_numpy as O
    n, h, w, c = logits_nhwc.shape
    pred = logits_nhwc.reshape(n, h * w, c).argmax(-1)
    true = mask.reshape(n, h * w)
    vals = []
    for i in range(n):
        for k in range(first_class, c):
            p = (pred[i] == k).astype(np.float64)
            t = (true[i] == k).astype(np.float64)
            inter, sets = 2 * (p * t).sum(), p.sum() + t.sum()
            if sets == 0:
                sets = inter
            vals.append((inter + eps) / (sets + eps))
    assert O is not None
    return float(np.mean(vals))


def _call_both(torch, fn, stream, logits, mask, n_classes, first_class=1, eps=1e-6):
    n, h, w, c = logits.shape
    dc = 0 if logits.dtype == torch.bfloat16 else 1
    need = fn['unetk_eval_dice_workspace_bytes'](n, c)
    ws = torch.empty(int(need), dtype=torch.uint8, device=logits.device)
    ref = torch.full((1,), -1.0, dtype=torch.float32, device=logits.device)
    got = torch.full((1,), -2.0, dtype=torch.float32, device=logits.device)
    rc = fn['unetk_eval_dice'](logits.data_ptr(), c, mask.data_ptr(), n, h * w, n_classes, first_class, eps,
                               ws.data_ptr(), ws.numel(), ref.data_ptr(), dc, stream)
    assert rc == 0
    rc = fn['unetk_eval_dice_noalloc'](logits.data_ptr(), c, mask.data_ptr(), n, h * w, n_classes, first_class, eps,
                                       got.data_ptr(), dc, stream)
    assert rc == 0
    torch.cuda.synchronize()
    return float(ref), float(got)


@pytest.mark.parametrize('dtype_name', ['float32', 'bfloat16'])
def test_noalloc_matches_eval_dice_twice_second_smaller(dtype_name):
    import torch
    from unetk._native import fn
    dtype = getattr(torch, dtype_name)
    dev = torch.device('cuda', 0)
    stream = torch.cuda.current_stream().cuda_stream
    g = torch.Generator(device=dev).manual_seed(7)
    # first call: 6 images x 21 classes (126 groups); second: 2 images x 3 classes on fewer pixels
    for (n, h, w, c) in [(6, 37, 53, 21), (2, 16, 19, 3)]:
        logits = torch.randn((n, h, w, c), device=dev, generator=g).to(dtype)
        mask = torch.randint(0, c, (n, h, w), device=dev, generator=g)
        mask[0, : h // 2] = 0                                   # some classes absent from an image: empty-set rule
        ref, got = _call_both(torch, fn, stream, logits, mask, c)
        assert got == ref, (got, ref)                           # integer counting + identical finalize: bit-exact
        want = _oracle_dice(logits.float().cpu().numpy(), mask.cpu().numpy())
        assert rel_err(got, want) < 1e-6


def test_noalloc_full_capacity_then_small_and_rejects_oversize():
    import torch
    from unetk._native import fn
    dev = torch.device('cuda', 0)
    stream = torch.cuda.current_stream().cuda_stream
    g = torch.Generator(device=dev).manual_seed(11)
    # exactly the maximum: 128 images x 32 classes = 4096 groups, every counter of the static array in use
    for (n, h, w, c) in [(128, 8, 9, 32), (1, 5, 7, 2)]:
        logits = torch.randn((n, h, w, c), device=dev, generator=g)
        mask = torch.randint(0, c, (n, h, w), device=dev, generator=g)
        ref, got = _call_both(torch, fn, stream, logits, mask, c)
        assert got == ref, (got, ref)
    # one group too many -> clear error, nothing launched, output untouched
    n, c = 4097, 2
    logits = torch.randn((n, 2, 2, c), device=dev, generator=g)
    mask = torch.randint(0, c, (n, 2, 2), device=dev, generator=g)
    out = torch.full((1,), 123.0, device=dev)
    rc = fn['unetk_eval_dice_noalloc'](logits.data_ptr(), c, mask.data_ptr(), n, 4, c, 1, 1e-6, out.data_ptr(), 1, stream)
    torch.cuda.synchronize()
    assert rc == -6 and float(out) == 123.0


def test_noalloc_does_not_allocate():
    """The caching allocator's counters stay put across the call (the library itself never calls hipMalloc:
    the scratch is a static device array)."""
    import torch
    from unetk._native import fn
    dev = torch.device('cuda', 0)
    stream = torch.cuda.current_stream().cuda_stream
    logits = torch.randn((3, 11, 13, 4), device=dev)
    mask = torch.randint(0, 4, (3, 11, 13), device=dev)
    out = torch.zeros(1, device=dev)
    fn['unetk_eval_dice_noalloc'](logits.data_ptr(), 4, mask.data_ptr(), 3, 143, 4, 1, 1e-6, out.data_ptr(), 1, stream)
    torch.cuda.synchronize()
    free0, _ = torch.cuda.mem_get_info()
    before = torch.cuda.memory_stats()['allocation.all.allocated']
    for _ in range(3):
        rc = fn['unetk_eval_dice_noalloc'](logits.data_ptr(), 4, mask.data_ptr(), 3, 143, 4, 1, 1e-6, out.data_ptr(), 1,
                                           stream)
        assert rc == 0
    torch.cuda.synchronize()
    free1, _ = torch.cuda.mem_get_info()
    assert torch.cuda.memory_stats()['allocation.all.allocated'] == before
    assert free1 == free0
