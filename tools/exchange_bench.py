#!/usr/bin/env python3
"""Times the pieces of the factored SH-gradient exchange on ONE GPU at C3 size: the raw-SH backward with / without
materialised SH rows, and adgs_sh_grad_expand for n = 1, 2, 4, 8 cameras (what every rank runs after the all-gather)."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "ad-gs_amd")):
    sys.path.insert(0, p)
from adgs import dp  # noqa: E402


def timeit(fn, n=30):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n):
        fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / n


def main():
    P, Ns, M, C = 1_000_000, 800_000, 16, 12
    dev = torch.device("cuda", 0)
    g = torch.Generator(device=dev).manual_seed(0)
    head = torch.randn(Ns, 3, device=dev, generator=g) + torch.tensor([0, 0, 8.0], device=dev)
    outs = [torch.empty(s, device=dev) for s in ((Ns, 1, 3), (P - Ns, 1, 3), (Ns, M - 1, 3), (P - Ns, M - 1, 3), (Ns, 3, C), (P - Ns, 3, C))]
    for n in (1, 2, 4, 8, 16):
        cams = []
        for c in range(n):
            rgb = torch.randn(P, 3, device=dev, generator=g)
            rgb[torch.rand(P, device=dev, generator=g) < 0.1] = 0
            cams.append((rgb, torch.randn(P - Ns, 3, device=dev, generator=g) + 8.0, [0.1 * c, 0.0, 0.0]))
        W = torch.randn(n, C, device=dev, generator=g)
        ms = timeit(lambda: dp.hip_sh_grad_expand(cams, W, C, P, Ns, Ns, head, 3, M, outs))
        byts = P * (12 * M + 12 * C) + n * (P * 12 * 2 + (P - Ns) * 12) + Ns * 12
        print("expand n=%2d cameras: %.3f ms  (%.0f GB/s of %d MB algorithmic)" % (n, ms, byts / ms / 1e6, byts >> 20))
    # the blob copies + a same-device "gather" stand-in
    send = torch.empty(1, 3 * P + 3 * (P - Ns), device=dev)
    f, x = torch.randn(P, 3, device=dev), torch.randn(P, 3, device=dev)
    ms = timeit(lambda: (send[0, :3 * P].copy_(f.reshape(-1)), send[0, 3 * P:].copy_(x[Ns:].reshape(-1))))
    print("packing one camera's blob (%.1f MB): %.3f ms" % (send.numel() * 4 / 1e6, ms))


if __name__ == "__main__":
    main()
