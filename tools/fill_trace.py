"""torch's fill_ over config 5's 380 MB, a few launches: run under `rocprofv3 --kernel-trace` to read its grid / workgroup size (GPU box)."""
import torch
x = torch.empty(2048 * 64 * 6 * 121, device="cuda:0")
for _ in range(30):
    x.fill_(1.0)
torch.cuda.synchronize()
