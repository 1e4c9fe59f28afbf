"""Phase times of conv3x3_k32n64_kernel from the diagnostic build (UNETK_K32_STAMP=1, exp/libunetk_stamp.so):
per wave the summed s_memtime cycles of [stage prologue LDS reads, MFMA k-steps, end-of-stage waitcnt, barrier,
tile hand-over].  UNETK_LIB must point at the stamp build."""
import os, sys
import torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'pytorch-unet_amd'))
N, h, w = 16, 640, 959
cin, cout = int(os.environ.get('CIN', 64)), 64
nblk_max = 8192
buf = torch.zeros(nblk_max * 8 * 8, dtype=torch.int64, device='cuda')
os.environ['UNETK_STAMP_PTR'] = str(buf.data_ptr())
from unetk import ops
x = (torch.rand((N, h, w, cin), device='cuda') - 0.5).bfloat16()
wt = (torch.rand((cout, cin, 3, 3), device='cuda') - 0.5) * 0.05
wpk = ops.pack_weight(0, wt, cout, cin, torch.bfloat16)
stats = ops.new_stats(cout, 'cuda')
y = ops.conv3x3_fprop(x, wpk, cout, stats=stats)
for _ in range(200):                      # let the clock settle under load
    ops.conv3x3_fprop(x, wpk, cout, out=y, stats=stats)
torch.cuda.synchronize()
buf.zero_()
ops.conv3x3_fprop(x, wpk, cout, out=y, stats=stats)
torch.cuda.synchronize()
r = buf.view(-1, 8).cpu().numpy()
r = r[r[:, 5] > 0]
tiles = r[:, 5].astype(float)
names = ['prologue reads', 'mfma steps', 'waitcnt', 'barrier', 'hand-over']
tot = 0
for q, n in enumerate(names):
    per = (r[:, q] / tiles).mean()
    tot += per
    print(f'{n:16s} {per:9.0f} cycles per tile per wave')
print(f'{"sum":16s} {tot:9.0f}   (waves {len(r)}, tiles per workgroup {tiles.mean():.1f})')
