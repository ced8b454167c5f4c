"""What a plain write-only kernel reaches on this card: torch's fill_ over buffers of the sizes the step kernels write (380 MB = config 5's
observations of 2 048 envs, 617 MB = config 3's of 65 536, 190 MB fits the 256 MB Infinity Cache).  GPU box."""
import torch
for mb in (190, 380, 617, 1234, 4936):
    n = mb * 1000 * 1000 // 4
    x = torch.empty(n, device="cuda:0")
    for _ in range(50):
        x.fill_(1.0)
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(100):
        x.fill_(2.0)
    b.record()
    torch.cuda.synchronize()
    us = a.elapsed_time(b) * 10
    print(f"fill_ {mb:5d} MB: {us:8.1f} us  {n * 4 / us / 1e6:.2f} TB/s", flush=True)
    del x
