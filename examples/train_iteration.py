#!/usr/bin/env python3
"""One AD-GS training iteration, train.py:47-167 term for term, on the HIP path only:

    camera draw -> render() (fused deformation, flow time, semantic mask, environment map composited in the blend epilogue)
    -> L1 + SSIM, scale/shift-invariant depth loss, flow re-projection loss, object BCE, sky BCE (train.py:78-103)
    -> the three regularisers reg_loss / sigma_loss / reg_sigma_loss over obj_near_idx (train.py:104-113)
    -> backward -> densification statistics -> every `densification_interval` iterations densify_and_prune, otherwise every
       `near_idx_reset_interval` iterations set_obj_near_idx, every `opacity_reset_interval` iterations reset_opacity (train.py:146-158)
       -> both Adam steps (train.py:163-167).

    python examples/train_iteration.py [--config C3] [--iters 60] [--env-res 8192] [--cameras 16] [--no-adam-in-backward] [--json]

The Adam step of the SH `rest` and SH deformation tensors (81 of a scene Gaussian's 95 parameters) is applied inside the rasterizer's
backward (FusedAdam(in_backward=True) + arm_backward(): bit-identical to the separate step, tests/test_gpu_optim.py) in the
iterations whose step follows their backward unconditionally; --no-adam-in-backward steps everything from materialised gradients.

Synthetic scene, cameras and targets; the lambdas and intervals are the reference's defaults
(arguments/__init__.py:104-133).  The reference reads `loss.item()` every iteration (train.py:132: a host synchronisation per
iteration); here the running loss stays on the device and is read when the progress bar would print it (every 10 iterations).
This is NOT the headline metric (bench.py measures the rasterizer path as BASELINE.json defines it): it shows the drop-in pieces
working together and where an iteration's time goes; bench.py attaches its summary as the secondary `train_iteration` object.
"""
import argparse
import json
import os
import sys
import time
import types

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "ad-gs_amd")):
    if p not in sys.path:
        sys.path.insert(0, p)

# arguments/__init__.py:104-133
OPT = types.SimpleNamespace(lambda_dssim=0.2, lambda_l1=1.0, lambda_depth=0.1, lambda_flow=0.1, lambda_obj=0.1, lambda_sky=0.05, lambda_sigma=0.01,
                            lambda_reg=0.5, lambda_sigma_reg=0.5, near_num=8, near_idx_reset_interval=10, densification_interval=200,
                            densify_scene_grad_threshold=None, densify_obj_grad_threshold=None, opacity_reset_interval=3000, min_opacity=0.005, env_lr=1e-2)
FUSED_IMAGE_LOSSES = os.environ.get("ADGS_FUSED_IMAGE_LOSSES", "1") != "0"
STAGES = ("regularisers", "render", "losses", "backward", "densify_stats", "near_idx_or_densify", "adam_gaussians", "adam_env_map")


def build(config, env_res, device, n_cameras=16, adam_in_backward=True):
    import torch
    from adgs import synthetic, env
    from adgs.model import SyntheticGaussianModel
    cfg = synthetic.CONFIGS[config]
    # driving-dataset world frame (z up, cameras look horizontally): in the canonical camera's own frame every camera would stare
    # at the environment map's pole, where neighbouring pixels land on texels all around the azimuth circle
    sc = synthetic.to_z_up_world(synthetic.make_config_scene(config))
    model = SyntheticGaussianModel.from_scene(sc, device=device, seed=0)
    model.raw_sh = True
    model.raw_scene = True          # scene-range activations inside the rasterizer's preprocess
    model.frame_gap = 0.02          # scene/gaussian_model.py:266 (1 / number of frames)
    lrs = {"scene_xyz": 1.6e-4, "obj_xyz": 1.6e-4, "scene_shs_dc": 2.5e-3, "obj_shs_dc": 2.5e-3, "scene_shs_rest": 1.25e-4, "obj_shs_rest": 1.25e-4,
           "scene_opacity": 0.05, "obj_opacity": 0.05, "scene_scaling": 5e-3, "obj_scaling": 5e-3, "scene_rotation": 1e-3, "obj_rotation": 1e-3}
    model.training_setup(lrs=lrs, scene_extent=20.0, object_extent=4.0, near_num=OPT.near_num, adam_in_backward=adam_in_backward)
    # sparse_grad: the map's backward marks the tiles it writes and reuses the gradient buffer the optimizer zeroes (adgs/env.py) --
    # no 0.8 GB fill and no 0.8 GB scan of the dense gradient per iteration; the update itself is unchanged
    env_map = env.EnvironmentMap(env_res, 3, device=device, sparse_grad=True)
    env_map.training_setup(types.SimpleNamespace(env_lr=OPT.env_lr))
    g = torch.Generator().manual_seed(11)
    H, W = cfg["H"], cfg["W"]
    cams = []
    import bench
    for cam, t in bench.camera_pool(cfg, n_cameras):
        cam = synthetic.camera_to_z_up(cam)
        c = synthetic.camera_object(cam, time=t)
        c.cam_id = len(cams)
        # scene/cameras.py:77-80: the reference's Camera holds its matrices on the GPU
        for name in ("world_view_transform", "full_proj_transform", "camera_center"):
            setattr(c, name, getattr(c, name).to(device))
        # per-camera supervision (scene/cameras.py: original_image, depth, semantic, sky, flow packages)
        c.original_image = torch.rand(3, H, W, generator=g).to(device)
        c.depth = (torch.rand(H, W, generator=g) * 0.5 + 0.01).to(device)                  # monocular inverse depth
        c.semantic = (torch.rand(H, W, generator=g) > 0.8).float().to(device)
        c.sky = (torch.rand(H, W, generator=g) > 0.7).float().to(device)
        K = torch.tensor([[cfg["focal"], 0.0, W / 2.0], [0.0, cfg["focal"], H / 2.0], [0.0, 0.0, 1.0]])
        flow = torch.stack([torch.rand(H, W, generator=g) * (W - 1), torch.rand(H, W, generator=g) * (H - 1)]).to(device)
        vis = (torch.rand(H, W, generator=g) > 0.3).float().to(device)
        c.flow = [(t + 0.05, K, cam["viewmatrix"][:3, :3].t().contiguous(), cam["viewmatrix"][3, :3].contiguous(), flow, vis)]
        cams.append(c)
    return cfg, model, cams, env_map


class StageClock:
    """HIP events around the stages of an iteration (recorded on the launch stream; read after a synchronisation)."""

    def __init__(self, on):
        import torch
        self.on, self.torch, self.marks, self.iters = on, torch, [], []

    def mark(self, name):
        if self.on:
            e = self.torch.cuda.Event(enable_timing=True)
            e.record()
            self.marks.append((name, e, time.perf_counter()))

    def end_iteration(self):
        if self.on:
            self.iters.append(self.marks)
            self.marks = []

    def summary(self):
        tot = {}
        for marks in self.iters:
            for (_, a, _ta), (name, b, _tb) in zip(marks[:-1], marks[1:]):
                tot[name] = tot.get(name, 0.0) + a.elapsed_time(b)
        n = max(len(self.iters), 1)
        return {k: round(v / n, 4) for k, v in tot.items()}

    def host_summary(self):
        """Host time between the same marks (time.perf_counter: what the Python side of a stage costs, enqueue included; where it
        exceeds the device figure the GPU waits for the host)."""
        tot = {}
        for marks in self.iters:
            for (_, _a, ta), (name, _b, tb) in zip(marks[:-1], marks[1:]):
                tot[name] = tot.get(name, 0.0) + (tb - ta) * 1e3
        n = max(len(self.iters), 1)
        return {k: round(v / n, 4) for k, v in tot.items()}


def densify_threshold(model):
    """A gradient threshold that clones / splits ~0.1 % of the Gaussians on this synthetic scene (the reference's thresholds are tuned to real
    data, arguments/__init__.py:118-119): the 99.9 % quantile of the accumulated statistics.  One-off (a torch sort + a host read)."""
    import torch
    g = (model.xyz_gradient_accum / model.denom.clamp_min(1)).reshape(-1)
    return float(torch.quantile(g[g > 0][:2_000_000], 0.999)) if bool((g > 0).any()) else 1.0


def iteration(it, model, cams, env_map, clock, state):
    import torch
    from adgs import loss
    from gaussian_renderer import render
    opt = OPT
    pipe = types.SimpleNamespace(inv_depth=True, debug=False)
    cam = cams[it % len(cams)]                                               # train.py:55-61
    flow_pkg = cam.flow[0]                                                   # :66-71
    clock.mark("start")
    # The three regularisers (train.py:101-110) do not depend on the render: they are evaluated FIRST.  Same terms, same sum -- but their
    # launches (and the Python around them) now run while the GPU is still busy with the previous iteration's Adam, and autograd, which
    # runs the nodes created last first, back-propagates them LAST, under the rasterizer's long backward kernels, instead of between
    # the render and the image losses where the GPU has nothing queued (the rasterizer's forward returns when the device has published
    # the frame's pair count: the host is at most one blend kernel ahead at that point).
    reg_loss = loss.reg_loss(model.xyz_deform_param, model.obj_near_idx)     # :101-103
    sigma_loss = loss.sigma_loss(model.gs_time_sigma, model.frame_gap)       # :105-107
    reg_sigma_loss = loss.reg_sigma_loss(model.gs_time_sigma, model.obj_near_idx)      # :108-110
    clock.mark("regularisers")
    pkg = render(cam, model, env_map, pipe, flow_pkg=flow_pkg, render_objmask=opt.lambda_obj > 0.0)      # :73
    image = pkg["render"]
    clock.mark("render")
    if FUSED_IMAGE_LOSSES:
        # :78-99 as one autograd node (adgs.loss.image_losses: the same six kernels and values, two Python-level calls instead of twelve)
        Ll1, s, depth_loss, flow_loss, obj_loss, sky_loss = loss.image_losses(
            image, cam.original_image, pkg["depth"], cam.depth, pkg["img_flow"], flow_pkg, pkg["img_opacity"], pkg["img_semantic"], cam.semantic, cam.sky,
            dist=model.scene_extent * 1e-3)
        dssim = 1.0 - s
    else:
        Ll1, s = loss.l1_ssim(image, cam.original_image)                         # :78-80 (one fused kernel for both)
        dssim = 1.0 - s
        depth_loss = loss.get_depth_loss(pkg["depth"], cam.depth)                # :83-86
        flow_loss = loss.get_flow_loss(pkg["img_flow"], flow_pkg, pkg["img_opacity"], dist=model.scene_extent * 1e-3)      # :88-89
        obj_loss = loss.obj_loss(pkg["img_semantic"], cam.semantic)              # :91-94
        sky_loss = loss.sky_loss(pkg["img_opacity"], cam.sky)                    # :96-99
    # :112-115 -- the reference's chain of python scalar products and sums is ~40 launches of 2 - 4 us; same total in three
    total = loss.weighted_total([((1.0 - opt.lambda_dssim) * opt.lambda_l1, Ll1), (opt.lambda_dssim, dssim), (opt.lambda_depth, depth_loss),
                                 (opt.lambda_flow, flow_loss), (opt.lambda_sky, sky_loss), (opt.lambda_obj, obj_loss), (opt.lambda_sigma, sigma_loss),
                                 (opt.lambda_reg, reg_loss), (opt.lambda_sigma_reg, reg_sigma_loss)])
    clock.mark("losses")
    if getattr(model.optimizer, "backward_epilogue", None) is not None and (it + 1) % opt.densification_interval != 0:
        # the step of this iteration follows its one backward whatever happens in between (a densification replaces the parameter
        # tensors: the reference's step then finds no gradients and changes nothing, so those iterations stay on the ordinary path)
        model.optimizer.arm_backward()
    total.backward()                                                         # :116
    clock.mark("backward")
    with torch.no_grad():
        state["ema"] = 0.4 * total.detach() + 0.6 * state.get("ema", total.detach())       # :132, kept on the device
        state["l1"] = Ll1.detach()
        model.add_densification_stats(pkg)                                   # :148-150 (max_radii2D and the gradient statistics, one kernel)
        clock.mark("densify_stats")
        n = it + 1
        if n % opt.densification_interval == 0:                              # :152-153
            if state.get("thr") is None:
                state["thr"] = densify_threshold(model)
            model.densify_and_prune(state["thr"], state["thr"], opt.min_opacity, False)
            state["densified"] = state.get("densified", 0) + 1
        elif model.use_near_idx and n % opt.near_idx_reset_interval == 0:    # :154-155
            model.set_obj_near_idx()
        if n % opt.opacity_reset_interval == 0:                              # :157-158
            model.reset_opacity()
        clock.mark("near_idx_or_densify")
        model.optimizer.step(zero_grad=True)                                 # :163-167 (after a densification the new parameter tensors
        clock.mark("adam_gaussians")                                         #  carry no gradient: that step is skipped, as in the reference)
        env_map.optimizer.step(zero_grad=True)
        clock.mark("adam_env_map")
    clock.end_iteration()
    return total.detach()


def run(config="C3", iters=60, env_res=8192, cameras=16, warm=12, device=None, stages=True, adam_in_backward=True):
    import torch
    device = device or torch.device("cuda", 0)
    cfg, model, cams, env_map = build(config, env_res, device, cameras, adam_in_backward)
    state = {}
    off = StageClock(False)
    for i in range(warm):
        iteration(i, model, cams, env_map, off, state)
    state["thr"] = densify_threshold(model)          # from the warm-up's statistics: not part of an iteration
    # the FIRST densify_and_prune of a process costs ~100 ms of one-off set-up (code loading, first use of a dozen torch ops, allocator
    # growth; every later one 1.8 ms): done here, like the rest of the warm-up, so that the timed window holds steady-state calls only
    with torch.no_grad():
        model.densify_and_prune(state["thr"], state["thr"], 0.005, False)
    for i in range(2):
        iteration(warm + i, model, cams, env_map, off, state)
    torch.cuda.synchronize()
    clock = StageClock(stages)
    t0 = time.perf_counter()
    first = last = None
    for i in range(iters):
        last = iteration(warm + 2 + i, model, cams, env_map, clock, state)
        first = last if first is None else first
        if (i + 1) % 10 == 0:
            _ = float(state["ema"])                                          # what the progress bar prints (train.py:133-138)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / iters
    return {"workload": "%s training iteration: render (deformation, flow, semantic, %d^2 environment map) + L1/SSIM + depth + flow + 2 BCE + 3 regularisers "
                        "+ backward + densification statistics + fused Adam (Gaussians, environment map); set_obj_near_idx every %d and "
                        "densify_and_prune every %d iterations; %d cameras" % (config, env_res, OPT.near_idx_reset_interval, OPT.densification_interval, len(cams)),
            "iterations": iters, "adam_in_backward": bool(adam_in_backward), "ms_per_iteration": round(dt * 1e3, 4), "iterations_per_s": round(1.0 / dt, 2),
            "stage_ms": clock.summary(), "host_stage_ms": clock.host_summary(), "densify_calls": state.get("densified", 0), "points_end": int(model.get_pts_num),
            "loss_first_last": [round(float(first), 6), round(float(last), 6)],
            "note": "stage_ms: HIP events on the launch stream, mean per iteration (near_idx_or_densify is amortised over the iterations)"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--config", default="C3")
    ap.add_argument("--iters", type=int, default=60)
    ap.add_argument("--env-res", type=int, default=8192)
    ap.add_argument("--cameras", type=int, default=16)
    ap.add_argument("--no-adam-in-backward", action="store_true")
    ap.add_argument("--json", action="store_true")
    args = ap.parse_args()
    import torch
    if not torch.cuda.is_available():
        raise SystemExit("needs an MI355X: there is no CPU fallback")
    res = run(args.config, args.iters, args.env_res, args.cameras, adam_in_backward=not args.no_adam_in_backward)
    if args.json:
        print(json.dumps(res))
    else:
        print("%s\n%.3f ms/iteration = %.1f iterations/s; stages (ms): %s; loss %.5f -> %.5f" % (
            res["workload"], res["ms_per_iteration"], res["iterations_per_s"], res["stage_ms"], res["loss_first_last"][0], res["loss_first_last"][1]))


if __name__ == "__main__":
    main()
