"""Ad-hoc timing of the ConvTranspose2d(k2,s2) kernels at the config-2 decoder shapes (batch 16)."""
import os, sys, time
import torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'pytorch-unet_amd'))
from unetk import ops

N = 16
# (low-res h, w, c_in, skip H, W)
layers = [(40, 59, 1024, 80, 119), (80, 119, 512, 160, 239), (160, 239, 256, 320, 479), (320, 479, 128, 640, 959)]
dt = torch.bfloat16
if os.environ.get('ONLY'):
    layers = [layers[int(i)] for i in os.environ['ONLY'].split(',')]


def timeit(f, reps=20):
    for _ in range(3):
        f()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        f()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps


for (h, w, cin, hs, ws) in layers:
    cout = cin // 2
    x = (torch.rand((N, h, w, cin), device='cuda') - 0.5).to(dt)
    wt = (torch.rand((cin, cout, 2, 2), device='cuda') - 0.5) * 0.05
    bias = torch.zeros(cout, device='cuda')
    cat = torch.empty((N, hs, ws, cin), device='cuda', dtype=dt)
    wpk = ops.pack_weight(2, wt, cout, cin, dt)
    wpk_d = ops.pack_weight(3, wt, cout, cin, dt)
    pt, pl = (hs - 2 * h) // 2, (ws - 2 * w) // 2
    dw = torch.empty_like(wt)
    db = torch.empty(cout, device='cuda')
    fl = 2.0 * N * h * w * 4 * cin * cout
    byt_f = (x.numel() + N * 4 * h * w * cout) * 2
    from unetk._native import fn, check, dtype_code, stream_ptr
    t = timeit(lambda: fn['unetk_convT2x2_fwd'](x.data_ptr(), cin, wpk.data_ptr(), bias.data_ptr(), cat.data_ptr(), cin, cout, N, h, w, cin, cout, hs, ws, pt, pl, 0, stream_ptr()))
    print(f'{h}x{w} {cin}->{cout} fwd  : {t*1e3:.3f} ms {fl/t/1e12:7.1f} TF  {byt_f/t/1e12:.2f} TB/s', flush=True)
    t = timeit(lambda: ops.convT2x2_dgrad(cat, cout, wpk_d, cin, cout, h, w, pt, pl))
    print(f'{h}x{w} {cin}->{cout} dgrad: {t*1e3:.3f} ms {fl/t/1e12:7.1f} TF  {byt_f/t/1e12:.2f} TB/s', flush=True)
    t = timeit(lambda: ops.convT2x2_wgrad(x, cat, cout, dw, db, pt, pl))
    print(f'{h}x{w} {cin}->{cout} wgrad: {t*1e3:.3f} ms {fl/t/1e12:7.1f} TF  {byt_f/t/1e12:.2f} TB/s', flush=True)
