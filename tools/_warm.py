"""Timing hygiene shared by the diagnostic benches: this chip answers a few milliseconds of idle with its power ramp (tools/idle_probe.py:
after 3 / 10 ms of idle the next 20 launches run 5 / 15-20 % slower), so a timed loop must follow ~80 ms of uninterrupted launches."""
import time

import torch


def warm(fn, ms: float = 80.0, probe: int = 20):
    """Run `fn` back to back for about `ms` milliseconds of GPU time (estimated from `probe` calls), ending synchronised."""
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(probe):
        fn()
    torch.cuda.synchronize()
    per = max((time.perf_counter() - t0) / probe, 1e-6)
    for _ in range(int(ms * 1e-3 / per) + 1):
        fn()
    torch.cuda.synchronize()


def timed_us(fn, iters: int, ms: float = 80.0) -> float:
    """Microseconds per call of `fn`: HIP events around `iters` calls that follow a `warm` pass without a host-side gap."""
    warm(fn, ms)
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(iters):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / iters * 1000.0
