#!/usr/bin/env python3
"""Camera-parallel AD-GS training iterations on N MI355X (one process per GPU, RCCL), HIP path only:

    python -m torch.distributed.run --nproc-per-node N --master-addr 127.0.0.1 examples/train_dp.py [--config C4] [--iters 40]
    python examples/train_dp.py --cams 3            # one GPU: the same code accumulates the cameras locally

Every iteration renders `--cams` cameras (default: one per rank) dealt round-robin over the ranks -- render() with deformation,
flow and semantic outputs, fused L1+SSIM + auxiliary losses, backward -- then exchanges the gradients in factored form
(adgs.dp.FactoredSHExchange: all-gather of the colour-gradient factors, all-reduce of the dense remainder, local expansion),
all-reduces the densification statistics, takes one fused Adam step (identical on every replica) and, every
`--densify-every` iterations, densifies / prunes with a shared seed.  At the end the replicas are checked to be bit-identical.
The reference is single-GPU, one camera per iteration (train.py:55-167); this is the multi-GPU path of SURVEY.md 8(e).
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


def make_cameras(cfg, n):
    from adgs import synthetic
    cams = []
    for c in range(n):
        d = synthetic.make_camera(cfg["W"], cfg["H"], cfg["focal"], cam_seed=c)
        cams.append(synthetic.camera_object(d, time=0.1 + 0.8 * c / max(n - 1, 1)))
    return cams


def iteration(model, ex, cameras, targets, it, rank, world, densify_every=0, lambda_dssim=0.2):
    import torch
    import torch.distributed as dist
    from adgs import dp, loss
    from gaussian_renderer import render
    pipe = types.SimpleNamespace(inv_depth=True, debug=False)
    total = torch.zeros((), device=model._scene_xyz.device)
    ex.begin(len(cameras))
    for cam in dp.shard_cameras(cameras, rank, world):
        pkg = render(cam, model, None, pipe, flow_pkg=(cam.time + 0.05,) + (None,) * 5, render_objmask=True, sh_factor_sink=ex.sink_for)
        l, _, _ = loss.photometric_loss(pkg["render"], targets["image"], lambda_dssim)
        l = (l + 0.01 * (pkg["depth"] - targets["depth"]).abs().mean() + 0.01 * pkg["img_opacity"].mean()) / len(cameras)
        l.backward()
        with torch.no_grad():
            model.add_densification_stats(pkg)
        total += l.detach()
    ex.reduce([c.time for c in cameras], [c.camera_center.tolist() for c in cameras], flow_times=[c.time + 0.05 for c in cameras])
    with torch.no_grad():
        model.optimizer.step(zero_grad=False)
        model.zero_grad()
        if densify_every and (it + 1) % densify_every == 0:
            dp.allreduce_densification_stats(model.xyz_gradient_accum, model.denom, model.max_radii2D)
            dp.seed_all_ranks(1000 + it)
            thr = 2e-7
            model.densify_and_prune(thr, thr, 0.005, False)
    if world > 1:
        dist.all_reduce(total)
    return float(total)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--config", default="C4")
    ap.add_argument("--iters", type=int, default=40)
    ap.add_argument("--cams", type=int, default=0, help="cameras per iteration (default: one per rank)")
    ap.add_argument("--densify-every", type=int, default=20)
    args = ap.parse_args()
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")      # dmabuf IPC for RCCL across processes (set before HIP initialises)
    import torch
    import torch.distributed as dist
    from adgs import dp, synthetic
    from adgs.model import SyntheticGaussianModel
    world, rank, local = int(os.environ.get("WORLD_SIZE", "1")), int(os.environ.get("RANK", "0")), int(os.environ.get("LOCAL_RANK", "0"))
    if not torch.cuda.is_available():
        raise SystemExit("needs MI355X GPUs: there is no CPU fallback")
    # ADGS_DP_BACKEND=gloo: control-flow / bit-identity check of the multi-rank loop on a box with fewer GPUs than ranks (ranks share
    # devices, collectives go through the host) -- never a measurement
    backend = os.environ.get("ADGS_DP_BACKEND", "nccl")
    local = local if backend == "nccl" else local % torch.cuda.device_count()
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    if world > 1:
        if backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)       # nccl == RCCL on ROCm
        else:
            dist.init_process_group(backend, rank=rank, world_size=world)
    cfg = synthetic.CONFIGS[args.config]
    sc = synthetic.make_config_scene(args.config)
    model = SyntheticGaussianModel.from_scene(sc, dev, seed=0)            # same seed on every rank: identical replicas
    model.raw_sh = True
    model.raw_scene = True          # scene-range activations inside the rasterizer's preprocess
    model.training_setup(lrs={"scene_xyz": 1.6e-4, "obj_xyz": 1.6e-4, "scene_shs_dc": 2.5e-3, "obj_shs_dc": 2.5e-3, "scene_opacity": 0.05,
                              "obj_opacity": 0.05, "scene_scaling": 5e-3, "obj_scaling": 5e-3}, scene_extent=20.0, object_extent=4.0)
    cameras = make_cameras(cfg, args.cams or world)
    g = torch.Generator().manual_seed(11)
    targets = dict(image=torch.rand(3, cfg["H"], cfg["W"], generator=g).to(dev), depth=torch.rand(cfg["H"], cfg["W"], generator=g).to(dev) * 0.3)
    ex = dp.FactoredSHExchange(model, factor_xyz=True)       # render() evaluates positions and flow points in one deformation pass
    losses = []
    for it in range(args.iters + 3):
        if it == 3:
            torch.cuda.synchronize(); t0 = time.perf_counter()
        losses.append(iteration(model, ex, cameras, targets, it, rank, world, args.densify_every))
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / args.iters
    # replicas must hold the same bits
    same = True
    if world > 1:
        for p in model.parameters():
            ref = p.detach().clone()
            dist.broadcast(ref, src=0)
            same = same and bool(torch.equal(ref, p.detach()))
        flag = torch.tensor([int(same)], device=dev)
        dist.all_reduce(flag, op=dist.ReduceOp.MIN)
        same = bool(flag.item())
    if rank == 0:
        print("%s on %d GPU(s), %d cameras/iteration: %.2f ms/iteration = %.1f cameras/s; %d Gaussians at the end; loss %.5f -> %.5f; "
              "replicas identical: %s" % (args.config, world, len(cameras), dt * 1e3, len(cameras) / dt, model.get_pts_num, losses[0], losses[-1], same))
        print("LOSSES " + " ".join("%.9g" % l for l in losses))
    if world > 1:
        dist.destroy_process_group()
    return losses, same


if __name__ == "__main__":
    main()
