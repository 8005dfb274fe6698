#!/usr/bin/env python3
"""set_obj_near_idx at C3's object count (run under rocprofv3 --kernel-trace --stats for the per-kernel split)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "ad-gs_amd")):
    sys.path.insert(0, p)
import torch
from adgs import knn
No = int(sys.argv[1]) if len(sys.argv) > 1 else 200_000
g = torch.Generator(device="cuda").manual_seed(0)


class M:
    pass


m = M()
centres = (torch.rand(8, 3, device="cuda", generator=g) - 0.5) * torch.tensor([40.0, 2.0, 70.0], device="cuda") + torch.tensor([0.0, 1.5, 40.0], device="cuda")
m._obj_xyz = centres[torch.randint(0, 8, (No,), device="cuda", generator=g)] + torch.randn(No, 3, device="cuda", generator=g)
m.gs_time = torch.rand(No, 1, device="cuda", generator=g)
m.scene_extent, m.use_time_mask, m.use_near_idx, m.near_num = 20.0, True, True, 8
for mode in ("slab", "brute"):
    os.environ["ADGS_KNN_POINTS"] = mode
    for _ in range(2):
        knn.set_obj_near_idx(m)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(5):
        knn.set_obj_near_idx(m)
    torch.cuda.synchronize()
    print("%s: %.3f ms per set_obj_near_idx (%d object Gaussians, K = 8, 4-D)" % (mode, (time.perf_counter() - t0) / 5 * 1e3, No))
