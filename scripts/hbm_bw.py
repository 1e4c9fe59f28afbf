"""What the HBM of the box delivers to simple streaming kernels (torch element-wise ops over 2 GiB): the yardstick for
the HBM-bound kernels of DESIGN.md section 4 (measured: fill 6.7, x+x 6.1, x+y 6.0 TB/s of 8 TB/s nominal)."""
import torch, time
x = torch.empty(1<<30, dtype=torch.bfloat16, device='cuda').normal_()   # 2 GiB
y = torch.empty_like(x)
def t(f, reps=10):
    f(); torch.cuda.synchronize()
    a,b = torch.cuda.Event(True), torch.cuda.Event(True)
    a.record()
    for _ in range(reps): f()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b)/reps
gb = x.numel()*2/1e9
ms = t(lambda: y.copy_(x)); print(f'copy  {2*gb/ms:.2f} TB/s')
xf = x.view(torch.float32)
ms = t(lambda: torch.sum(xf)); print(f'sum(read only) {gb/ms:.2f} TB/s')
ms = t(lambda: y.zero_()); print(f'fill(write only) {gb/ms:.2f} TB/s')
ms = t(lambda: torch.add(x, x, out=y)); print(f'add x+x (1 read 1 write) {2*gb/ms:.2f} TB/s')
z = torch.empty_like(x)
ms = t(lambda: torch.add(x, y, out=z)); print(f'add x+y (2 reads 1 write) {3*gb/ms:.2f} TB/s')
