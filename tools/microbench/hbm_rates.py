#!/usr/bin/env python3
"""What a streaming kernel can reach on this GPU: device-to-device copy, fill and read-only reduction rates of torch's own
kernels over 256 MB ... 2 GB (HIP events, 20 repetitions).  The ceiling the per-kernel GB/s figures in DESIGN.md are held against."""
import torch

dev = torch.device("cuda", 0)


def timed(fn, reps=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / reps * 1e-3


for mb in (64, 256, 1024, 2048):
    n = mb * (1 << 20) // 4
    x = torch.empty(n, device=dev, dtype=torch.float32).normal_()
    y = torch.empty_like(x)
    t_copy = timed(lambda: y.copy_(x))
    t_fill = timed(lambda: y.fill_(1.0))
    t_sum = timed(lambda: x.sum())
    t_axpy = timed(lambda: torch.add(x, y, alpha=2.0, out=y))
    gb = n * 4 / 1e9
    print("%5d MB: copy %.2f TB/s (read + write), fill %.2f TB/s, sum %.2f TB/s (read only), y = x + 2y %.2f TB/s (2 reads + 1 write)" % (
        mb, 2 * gb / t_copy / 1e3, gb / t_fill / 1e3, gb / t_sum / 1e3, 3 * gb / t_axpy / 1e3))
