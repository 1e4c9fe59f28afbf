"""Run ONE conv3x3 fprop shape a few times (for rocprofv3 --pmc passes)."""
import os, sys
import torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'pytorch-unet_amd'))
from unetk import ops
h, w, cin, cout = [int(v) for v in os.environ.get('SHAPE', '80,119,512,512').split(',')]
N = int(os.environ.get('N', 16))
x = (torch.rand((N, h, w, cin), device='cuda') - 0.5).bfloat16()
wt = (torch.rand((cout, cin, 3, 3), device='cuda') - 0.5) * 0.05
wpk = ops.pack_weight(0, wt, cout, cin, torch.bfloat16)
stats = ops.new_stats(cout, 'cuda') if os.environ.get('STATS', '1') == '1' else None
for _ in range(3):
    y = ops.conv3x3_fprop(x, wpk, cout, stats=stats)
torch.cuda.synchronize()
