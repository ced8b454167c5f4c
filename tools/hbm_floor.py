#!/usr/bin/env python3
"""Practical HBM floor for the C3 traffic mix, with plain torch kernels (run on the GPU box):
617 MB streamed f32 stores (obs) + 134 MB u8 read + 134 MB u8 write (grid)."""
import torch
E = 65536
obs = torch.empty((E, 8, 6, 7, 7), dtype=torch.float32, device="cuda")
g1 = torch.zeros((E, 2, 32, 32), dtype=torch.uint8, device="cuda")
g2 = torch.zeros_like(g1)
def timeit(fn, n=50):
    for _ in range(5): fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / n * 1000
t_fill = timeit(lambda: obs.fill_(1.0))
t_copy = timeit(lambda: g2.copy_(g1))
def both():
    obs.fill_(1.0); g2.copy_(g1)
t_both = timeit(both)
mb = obs.numel() * 4 / 1e6
print(f"fill {mb:.0f} MB: {t_fill:.1f} us = {mb/t_fill:.2f} TB/s")
print(f"copy 134+134 MB: {t_copy:.1f} us = {268.4/t_copy:.2f} TB/s")
print(f"fill+copy (serial launches): {t_both:.1f} us = {(mb+268.4)/t_both:.2f} TB/s")
big = torch.empty(1 << 30, dtype=torch.uint8, device="cuda"); big2 = torch.empty_like(big)
t = timeit(lambda: big2.copy_(big), 20)
print(f"1 GiB copy: {t:.1f} us = {2*1073.7/t:.2f} TB/s")
t = timeit(lambda: big.fill_(3), 20)
print(f"1 GiB fill: {t:.1f} us = {1073.7/t:.2f} TB/s")
for gb in (0.6, 1.2, 2.85, 5.7):
    x = torch.empty(int(gb * 1e9) // 4, dtype=torch.float32, device="cuda")
    t = timeit(lambda: x.fill_(1.0), 20)
    print(f"{gb} GB f32 fill: {t:.1f} us = {x.numel() * 4 / t / 1e6:.2f} TB/s")
    del x
