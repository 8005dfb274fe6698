#!/usr/bin/env python3
"""Times the SURVEY 8(f) row-4 pieces at C3 / C5 size on one GPU: densify_and_prune (selection, plan, gathers of every
parameter tensor and both Adam moments), reset_opacity and the object neighbour index (K = 8, xyz + time)."""
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "ad-gs_amd")):
    sys.path.insert(0, p)
from adgs import synthetic  # noqa: E402
from adgs.model import SyntheticGaussianModel  # noqa: E402


def main():
    dev = torch.device("cuda", 0)
    for cfg_name in sys.argv[1:] or ["C3", "C5"]:
        sc = synthetic.make_config_scene(cfg_name)
        for rep in range(2):
            model = SyntheticGaussianModel.from_scene(sc, dev, seed=0)
            model.training_setup(percent_dense=0.01, scene_extent=20.0, object_extent=4.0, near_num=0)
            for p in model.parameters():           # populate the Adam moments
                p.grad = torch.randn_like(p) * 1e-3
            model.optimizer.step(); model.optimizer.zero_grad(set_to_none=True)
            N = model.get_pts_num
            g = torch.Generator(device=dev).manual_seed(1)
            model.denom = torch.randint(0, 4, (N, 1), device=dev, generator=g).float()
            model.xyz_gradient_accum = torch.rand(N, 1, device=dev, generator=g) * 3e-3 * (model.denom > 0)
            state_bytes = sum(p.numel() * 4 * 3 for p in model.parameters())
            torch.cuda.synchronize(); t0 = time.perf_counter()
            info = model.densify_and_prune(2.2e-3, 2.2e-3, 0.005, True)
            torch.cuda.synchronize(); t1 = time.perf_counter()
            model.reset_opacity()
            torch.cuda.synchronize(); t2 = time.perf_counter()
            model.use_near_idx, model.near_num = True, 8
            model.set_obj_near_idx()
            torch.cuda.synchronize(); t3 = time.perf_counter()
            if rep == 1:
                print("%s: %d -> %d Gaussians (scene clone/split/out %s, obj %s); densify_and_prune %.2f ms (%.1f GB of parameters + moments "
                      "gathered: %.0f GB/s read+write), reset_opacity %.2f ms, neighbour index of %d object Gaussians (K=8, 4-D) %.2f ms"
                      % (cfg_name, N, model.get_pts_num, info["scene"], info["obj"], (t1 - t0) * 1e3, state_bytes / 1e9,
                         2 * state_bytes / (t1 - t0) / 1e9, (t2 - t1) * 1e3, model.get_obj_pts_num, (t3 - t2) * 1e3))
            del model
            torch.cuda.empty_cache()


if __name__ == "__main__":
    main()
