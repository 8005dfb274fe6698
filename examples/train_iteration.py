#!/usr/bin/env python3
"""One AD-GS training iteration (train.py:74-167 without data loading and the rare densify step) on the HIP path only:
render() with deformation + flow + semantic + environment-map background, fused L1+SSIM loss plus simple depth / opacity /
flow / semantic terms, backward, densification statistics, fused Adam on the Gaussian parameters and the environment map.

    python examples/train_iteration.py [--config C3] [--iters 50] [--env-res 8192]

Synthetic scene and targets; prints iterations/s and where the time goes.  This is NOT the headline
metric (bench.py measures the rasterizer path as BASELINE.json defines it); it shows the drop-in pieces working together.
"""
import argparse
import os
import sys
import time
import types

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "ad-gs_amd")):
    if p not in sys.path:
        sys.path.insert(0, p)


def build(config, env_res, device):
    import torch
    from adgs import synthetic, env, optim
    from adgs.model import SyntheticGaussianModel
    cfg = synthetic.CONFIGS[config]
    sc = synthetic.make_config_scene(config)
    model = SyntheticGaussianModel.from_scene(sc, device=device, seed=0)
    model.raw_sh = True
    model.raw_scene = True          # scene-range activations inside the rasterizer's preprocess
    cam = synthetic.camera_object(sc, time=0.37)
    cam.cam_id = 0
    env_map = env.EnvironmentMap(env_res, 3, device=device)
    env_map.training_setup(types.SimpleNamespace(env_lr=1e-2))
    lrs = {"_scene_xyz": 1.6e-4, "_obj_xyz": 1.6e-4, "_scene_shs_dc": 2.5e-3, "_obj_shs_dc": 2.5e-3, "_scene_shs_rest": 1.25e-4, "_obj_shs_rest": 1.25e-4,
           "_scene_opacity": 0.05, "_obj_opacity": 0.05, "_scene_scaling": 5e-3, "_obj_scaling": 5e-3, "_scene_rotation": 1e-3, "_obj_rotation": 1e-3}
    from adgs.model import _RAW
    groups = [{"params": [getattr(model, n)], "lr": lrs.get(n, 1e-3), "name": n} for n in _RAW if getattr(model, n, None) is not None and getattr(model, n).numel() > 0]
    model.optimizer = optim.FusedAdam(groups, lr=0.0, eps=1e-15)
    N = sc["P"]
    stats = dict(accum=torch.zeros(N, 1, device=device), denom=torch.zeros(N, 1, device=device), max_r=torch.zeros(N, device=device))
    g = torch.Generator().manual_seed(11)
    H, W = cfg["H"], cfg["W"]
    targets = dict(image=torch.rand(3, H, W, generator=g).to(device), depth=torch.rand(H, W, generator=g).to(device) * 50,
                   flow=torch.randn(3, H, W, generator=g).to(device), sem=(torch.rand(1, H, W, generator=g) > 0.8).float().to(device))
    return cfg, model, cam, env_map, stats, targets


def iteration(model, cam, env_map, stats, targets, lambda_dssim=0.2):
    import torch
    from adgs import loss, optim
    from gaussian_renderer import render
    pipe = types.SimpleNamespace(inv_depth=True, debug=False)
    flow_pkg = (cam.time + 0.05, None, None, None, None, None)
    pkg = render(cam, model, env_map, pipe, flow_pkg=flow_pkg, render_objmask=True)
    total, l1, dssim = loss.photometric_loss(pkg["render"], targets["image"], lambda_dssim)
    total = total + 0.01 * (pkg["depth"] - targets["depth"]).abs().mean() + 0.01 * (pkg["img_flow"] - targets["flow"]).abs().mean() \
        + 0.01 * torch.nn.functional.binary_cross_entropy(pkg["img_semantic"].clamp(1e-6, 1 - 1e-6), targets["sem"]) \
        + 0.01 * pkg["img_opacity"].mean()
    total.backward()
    with torch.no_grad():
        optim.add_densification_stats(stats["accum"], stats["denom"], stats["max_r"], pkg["viewspace_points"].grad, pkg["radii"])
        model.optimizer.step(zero_grad=False)
        env_map.optimizer.step(zero_grad=False)
        for p in model.parameters():
            p.grad = None
        env_map.grid_map.grad = None
    return total.detach(), l1.detach(), dssim.detach()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--config", default="C3")
    ap.add_argument("--iters", type=int, default=50)
    ap.add_argument("--env-res", type=int, default=8192)
    args = ap.parse_args()
    import torch
    if not torch.cuda.is_available():
        raise SystemExit("needs an MI355X: there is no CPU fallback")
    device = torch.device("cuda", 0)
    cfg, model, cam, env_map, stats, targets = build(args.config, args.env_res, device)
    for _ in range(5):
        iteration(model, cam, env_map, stats, targets)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    first = last = None
    for _ in range(args.iters):
        total, l1, dssim = iteration(model, cam, env_map, stats, targets)
        first = total if first is None else first
        last = total
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / args.iters
    print("%s: %.2f ms/iteration = %.1f iterations/s (render + env map %d^2 + L1/SSIM + aux losses + backward + densify stats + Adam); "
          "loss %.5f -> %.5f" % (args.config, dt * 1e3, 1.0 / dt, args.env_res, float(first), float(last)))


if __name__ == "__main__":
    main()
