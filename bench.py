#!/usr/bin/env python3
"""Headline benchmark: fwd+bwd frames/s of the AD-GS hot path on MI355X.

    python bench.py --gpus N --steps K --warmup W [--config C3]

One "step" = one frame = per-frame deformation (when the config has dynamic objects)
+ GaussianRasterizer forward + backward with non-zero upstream gradients on colour,
depth, accumulated opacity, flow and semantic; loss and optimizer are
excluded, as in BASELINE.md.  Inputs are synthetic (seeded) and resident
in HBM before the timed region.  With N > 1 every rank (one process per GPU, launched by
torch.distributed.run) renders its own camera of the same replicated scene and the
parameter gradients are all-reduced over RCCL inside the step (weak scaling).

Prints ONE JSON line on rank 0 (contract in the task description) with two extra
objects: "roofline" for the dominant kernel (algorithmic bytes per launch / HIP-event
time of that kernel) and "cpu_baseline" (the CPU oracle timed on the host cores).
"""
import argparse
import gc
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
for p in (ROOT, os.path.join(ROOT, "ad-gs_amd")):
    if p not in sys.path:
        sys.path.insert(0, p)

HBM_PEAK_GBS = 8000.0      # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec


def alg_bytes(P, V, R, X, T, M, F, D_S, passes):
    """Algorithmic bytes per launch of every stage."""
    pay = 12 + 4 + 12 * F + 4 * D_S
    return {
        "preprocess_fwd": P * (12 + 12 + 16 + 4 + 12 * M) + P * 4 + V * 68,
        "scan": P * 8,
        "duplicate_keys": V * 16 + R * 12,
        "radix_sort": passes * R * 24 + R * 8,
        "tile_ranges": R * 8 + T * 8,
        "render_fwd": R * 28 + R * pay + X * (12 + 4 + 4 + 12 * F + 4 * D_S + 4),
        "render_bwd": R * 28 + R * pay + X * (12 + 4 + 4 + 12 * F + 4 * D_S + 4 + 4) + V * (12 + 16 + 4 + 12 + 4 + 12 * F + 4 * D_S),
        "preprocess_bwd": P * (12 + 12 + 16 + 4 + 12 * M) + V * (24 + 3 + 4) + V * (12 + 16 + 12 + 4) + P * (12 + 12 * M + 12 + 16),
    }


def alg_bytes_v2(P, V, Rc, E, X, T, M, F, D_S, passes, n_obj_rows, E_pub=None):
    """Algorithmic bytes per launch of the default (v2, coarse-binned) pipeline -- DESIGN.md section 5.
    Rc = (cell, Gaussian) pairs that are sorted, E = (tile, Gaussian) entries actually blended."""
    pay = 12 + 4 + 12 * F + 4 * D_S
    out = X * (12 + 4 + 4 + 12 * F + 4 * D_S + 4)
    return {
        "preprocess_fwd": P * (12 + 12 + 16 + 4 + 12 * M) + P * 12 + V * (64 + 32 + 24 + 1 + 64),
        "scan": P * 12,
        "duplicate_keys": V * (8 + 4) + Rc * 12,
        "radix_sort": passes * Rc * 24 + Rc * 8,
        "tile_ranges": Rc * 8,
        "render_fwd": E * (4 + 64 + 4) + out,
        "render_bwd": (E if E_pub is None else E_pub) * (4 + 64 + 56) + X * (12 + 4 + 4 + 12 * F + 4 * D_S + 4 + 4),      # replays the published entries only
        "preprocess_bwd": P * (12 + 12 + 16 + 4 + 12 * M) + V * (64 + 32 + 24 + 1) + V * 64 + P * (12 + 12 * M + 12 + 16 + 12 + 16 + 4 + 12 + 4 + 24),
    }


def blended_entries_and_reference_pairs(frame, sc, settings, use_fs, device):
    """(E, R, E_pub): E = (tile, Gaussian) entries the v2 forward hands to the blend loop (64 x the chunks it published; the last
    chunk of a tile is partly empty, so this is an upper bound within #tiles x 63), R = the reference's num_rendered for the
    same frame (one extra forward in classic mode), E_pub = the entries at least one pixel blends, i.e. what the backward replays."""
    import torch
    from diff_gaussian_rasterization import _C
    from adgs import deform
    with torch.no_grad():
        if isinstance(frame, DeformFrame):
            pkg = deform.get_deformed_pkg(frame.model, frame.t)
            flow = frame.model.get_deformed_xyz(frame.t + 0.05) if use_fs else torch.empty(0, device=device)
            t = dict(means3D=pkg["xyz"], opacities=pkg["opacity"], scales=pkg["scales"], rotations=pkg["rotation"], shs=pkg["shs"])
            sem = frame.sem if use_fs else torch.empty(0, device=device)
        else:
            t = {k: v.detach() for k, v in frame.leaf.items()}
            flow = frame.flow if use_fs else torch.empty(0, device=device)
            sem = frame.sem if use_fs else torch.empty(0, device=device)
        e = torch.empty(0, device=device)
        s = settings
        call = lambda: _C.rasterize_gaussians(s.bg, t["means3D"], e, t["opacities"], t["scales"], t["rotations"], s.scale_modifier, e, s.viewmatrix,
                                              s.projmatrix, s.tanfovx, s.tanfovy, s.image_height, s.image_width, t["shs"], flow, sem, s.sh_degree,
                                              s.campos, s.prefiltered, s.inv_depth, False)
        out = call()
        chunks = int(out[6][:4].view(torch.int32)[0].item())      # BinStateV2 starts with the chunk-pool cursor
        import ctypes
        from adgs import _lib
        published = int(_lib.lib().adgs_test_v2_published_entries(out[7].data_ptr(), int(s.image_width), int(s.image_height),
                                                                  ctypes.c_void_p(torch.cuda.current_stream(device).cuda_stream)))
        old = os.environ.get("ADGS_RASTER_MODE")
        os.environ["ADGS_RASTER_MODE"] = "classic"
        try:
            r_ref = int(call()[0])
        finally:
            if old is None:
                del os.environ["ADGS_RASTER_MODE"]
            else:
                os.environ["ADGS_RASTER_MODE"] = old
    return chunks * 64, r_ref, published


def measured_frame_traffic(config, measured_case, fps_per_gpu):
    """HBM bytes per frame summed over every kernel of the frame, from the committed PMC passes (profiles/r01/)."""
    if config != "C3" or not measured_case:
        return {}
    try:
        tot = json.load(open(os.path.join(ROOT, "profiles", "r01", "hbm_traffic_per_frame.json")))
        return {"measured_hbm_bytes_per_frame": int(tot["hbm_bytes_per_frame"]),
                "measured_hbm_frac_of_8TBs": round(tot["hbm_bytes_per_frame"] * fps_per_gpu / 1e9 / HBM_PEAK_GBS, 4)}
    except (OSError, ValueError, KeyError):
        return {}


def pmc_annotations(stage, config, measured_case):
    """HBM traffic / VALU occupancy of the dominant kernel from the committed rocprofv3 --pmc passes (profiles/r01/, collected
    with this same command line; counters cannot be read from inside the process).  Only attached for the case they were
    measured on (C3, flow+semantic, default pipeline)."""
    if config != "C3" or not measured_case:
        return {}
    kname = {"render_bwd": "render_bwd_v2_kernel", "render_fwd": "render_fwd_v2_kernel", "preprocess_bwd": "preprocess_bwd_kernel",
             "preprocess_fwd": "preprocess_fwd_kernel"}.get(stage)
    out = {}
    try:
        tr = json.load(open(os.path.join(ROOT, "profiles", "r01", "hbm_traffic_per_kernel.json")))
        for k, v in tr.items():
            if kname and kname in k:
                out["traffic"] = v["hbm_bytes_per_launch"]
                out["traffic_source"] = "profiles/r01/hbm_traffic_per_kernel.json (rocprofv3 --pmc FETCH_SIZE x2 + WRITE_SIZE, per launch)"
                break
        pm = json.load(open(os.path.join(ROOT, "profiles", "r01", "pmc_blend_kernels.json")))["kernels"].get(kname)
        if pm:
            out["valu_busy_frac"] = pm["valu_busy_frac"]
            out["valu_insts_per_simd_cycle"] = pm.get("valu_insts_per_simd_cycle")
            out["valu_note"] = ("fp32-VALU-bound kernel (SURVEY.md 8(d)): valu_busy_frac = min(1, 4 x SQ_ACTIVE_INST_VALU / (1024 SIMDs x kernel cycles)); "
                                "valu_insts_per_simd_cycle against a full-rate peak of 0.5 (quarter-rate exp / rcp included); profiles/r01/pmc_blend_kernels.json")
    except (OSError, ValueError, KeyError):
        pass
    return out


class StaticFrame:
    """Per-frame work on already-deformed parameters: rasterizer forward (+ autograd backward)."""

    deform_bytes = 0
    deform_desc = "none (static, already activated parameters)"

    def __init__(self, sc, rasterizer, device, use_flow_sem):
        import torch
        self.rast = rasterizer
        self.leaf = {k: sc[k].to(device).clone().requires_grad_(True) for k in ("means3D", "opacities", "shs", "scales", "rotations")}
        self.means2D = torch.zeros(sc["P"], 3, device=device, requires_grad=True)
        self.flow = sc["flow_points"].to(device) if use_flow_sem else None
        self.sem = sc["semantic"].to(device) if use_flow_sem else None
        self.last_radii = None

    def parameters(self):
        return list(self.leaf.values())

    def zero_grad(self):
        for p in self.parameters():
            p.grad = None
        self.means2D.grad = None

    def forward(self):
        L = self.leaf
        color, radii, depth, op, flow, sem = self.rast(
            means3D=L["means3D"], means2D=self.means2D, opacities=L["opacities"], shs=L["shs"], scales=L["scales"],
            rotations=L["rotations"], flow_points=self.flow, semantic=self.sem)
        self.last_radii = radii
        return [color, depth, op] + ([flow, sem] if self.flow is not None else [])


class DeformFrame:
    """Per-frame work of the dynamic configs (C3-C5): fused B-spline/Fourier/quaternion-spline
    deformation of the raw parameters at the camera time (+ the flow points at t+0.05), then the
    rasterizer; the backward runs through both into every raw parameter."""

    def __init__(self, sc, rasterizer, device, use_flow_sem, t=0.37):
        from adgs.model import SyntheticGaussianModel
        self.rast, self.t, self.use_fs = rasterizer, t, use_flow_sem
        self.model = SyntheticGaussianModel.from_scene(sc, device, seed=0)
        self.model.raw_sh = os.environ.get("ADGS_BENCH_RAW_SH", "1") != "0"     # SH read straight from the raw parameters
        self.fused_flow = os.environ.get("ADGS_BENCH_FUSED_FLOW", "1") != "0"      # flow-time xyz in the same deformation pass
        self.means2D = None
        self.sem = self.model.get_obj_mask.float()[:, None].contiguous() if use_flow_sem else None
        self.last_radii = None
        self.deform_bytes = self.model.deform_bytes_per_frame()
        oa = self.model.order_args
        self.deform_desc = "fused HIP: xyz %s, rotation %s (quaternion spline), shs %s%s, time-masked opacity; %d object Gaussians" % (
            oa["xyz"], oa["rotation"], oa["shs"], " (read in place by the preprocess: raw-SH path)" if self.model.raw_sh else "",
            self.model.get_obj_pts_num)

    def parameters(self):
        return self.model.parameters()

    def zero_grad(self):
        self.model.zero_grad()

    def forward(self, sink_for=None):
        """sink_for: adgs.dp.FactoredSHExchange.sink_for -- the backward then leaves the SH gradients in factored form."""
        import torch
        m = self.model
        if self.use_fs and self.fused_flow:
            pkg = m.get_deformed_pkg(self.t, flow_time=self.t + 0.05)
            flow = pkg["flow_xyz"]
        else:
            pkg = m.get_deformed_pkg(self.t)
            flow = m.get_deformed_xyz(self.t + 0.05) if self.use_fs else None
        from gaussian_renderer import screenspace_points
        means2D = screenspace_points(pkg["xyz"].shape[0], pkg["xyz"].device)      # what render() uses: a fresh leaf over cached zeros
        if torch.is_tensor(pkg["shs"]):
            color, radii, depth, op, fl, sem = self.rast(
                means3D=pkg["xyz"], means2D=means2D, opacities=pkg["opacity"], shs=pkg["shs"], scales=pkg["scales"],
                rotations=pkg["rotation"], flow_points=flow, semantic=self.sem)
        else:
            color, radii, depth, op, fl, sem = self.rast.forward_rawsh(pkg["xyz"], means2D, pkg["opacity"], pkg["shs"], pkg["scales"],
                                                                       pkg["rotation"], flow_points=flow, semantic=self.sem,
                                                                       factor_sink=None if sink_for is None else sink_for(pkg["xyz"]))
        self.last_radii = radii
        return [color, depth, op] + ([fl, sem] if self.use_fs else [])

    def activated(self):
        """Activated tensors of this frame as CPU float32 (inputs of the CPU baseline)."""
        import torch
        with torch.no_grad():
            from adgs import deform
            pkg = deform.get_deformed_pkg(self.model, self.t)
            flow = self.model.get_deformed_xyz(self.t + 0.05)
        return {k: v.detach().cpu() for k, v in pkg.items()}, flow.cpu()


def cpu_baseline(sc, cam, cfg, use_fs, up):
    """The CPU oracle (oracle/, a port of the reference kernels -- the reference has no CPU path)
    timed on this host's cores for ONE frame of the same workload."""
    import numpy as np
    from oracle import oracle
    H, W = cfg["H"], cfg["W"]
    o = oracle.RasterOracle("f32")
    t0 = time.perf_counter()
    fwd = o.forward(sc["bg"], sc["means3D"], None, sc["opacities"], sc["scales"], sc["rotations"], 1.0, None, cam["viewmatrix"],
                    cam["projmatrix"], cam["tanfovx"], cam["tanfovy"], H, W, sc["shs"], sc["flow_points"] if use_fs else None,
                    sc["semantic"] if use_fs else None, cfg["sh_degree"], cam["campos"], False, True)
    t1 = time.perf_counter()
    o.backward(up["color"], up["depth"], up["flow"] if use_fs else np.zeros((3, H, W), np.float32), up["semantic"] if use_fs else None,
               up["img_opacity"])
    t2 = time.perf_counter()
    base = {"value": round(1.0 / (t2 - t0), 5), "unit": "frames/s", "cores": oracle.num_threads(), "kind": "port",
            "sample": "1 full frame (rasterizer fwd %.2f s + bwd %.2f s; the O(N) deformation is not included) of the same scene "
                      "and camera, OpenMP over Gaussians/tiles, g++ -O3 -fno-fast-math -ffp-contract=off" % (t1 - t0, t2 - t1)}
    return base, fwd


def parity_vs_oracle(hip_outs, oracle_fwd):
    """BASELINE.json's 'PSNR vs ref' (utils/image_utils.py:17-19: 20 log10(1 / sqrt(mse))) of the HIP frame against the CPU
    oracle's frame of the same inputs, plus the largest absolute deviations."""
    import numpy as np
    color = hip_outs[0].detach().float().cpu().numpy(); depth = hip_outs[1].detach().float().cpu().numpy().reshape(-1)
    oc = np.asarray(oracle_fwd["color"], np.float32).reshape(color.shape); od = np.asarray(oracle_fwd["depth"], np.float32).reshape(-1)
    mse = float(np.mean((color.astype(np.float64) - oc.astype(np.float64)) ** 2))
    return {"psnr_vs_oracle_db": round(20.0 * np.log10(1.0 / np.sqrt(mse)), 2) if mse > 0 else float("inf"),
            "max_abs_err_color": float(np.abs(color - oc).max()), "max_abs_err_depth": float(np.abs(depth - od).max()),
            "depth_scale": float(np.abs(od).max()),
            "frac_color_outside_1e-4": float(np.mean(np.abs(color - oc) > 1e-4 * (1.0 + np.abs(oc))))}


def quick_measure(config, steps, device, use_fs):
    """frames/s of another BASELINE.json config in the same process (single GPU, `steps` timed steps after a short warm-up):
    reported next to the headline as secondary information, never as `value`."""
    import torch
    from adgs import synthetic
    from diff_gaussian_rasterization import GaussianRasterizationSettings, GaussianRasterizer
    cfg = synthetic.CONFIGS[config]
    sc = synthetic.make_config_scene(config)
    cam = synthetic.make_camera(cfg["W"], cfg["H"], cfg["focal"])
    d = lambda t: t.to(device)
    settings = GaussianRasterizationSettings(cfg["H"], cfg["W"], cam["tanfovx"], cam["tanfovy"], d(sc["bg"]), 1.0, d(cam["viewmatrix"]),
                                             d(cam["projmatrix"]), cfg["sh_degree"], d(cam["campos"]), False, True, False)
    rast = GaussianRasterizer(settings)
    frame = DeformFrame(sc, rast, device, use_fs) if cfg["n_objects"] > 0 else StaticFrame(sc, rast, device, use_fs)
    up = synthetic.make_upstream_grads(sc, 0)
    ups = [d(up["color"]), d(up["depth"]), d(up["img_opacity"])] + ([d(up["flow"]), d(up["semantic"])] if use_fs else [])

    def step():
        torch.autograd.backward(frame.forward(), ups)
        frame.zero_grad()
    for _ in range(30):
        step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        step()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    return {"workload": "%s: %d Gaussians, %dx%d, SH deg %d, %d dynamic objects" % (config, cfg["P"], cfg["W"], cfg["H"], cfg["sh_degree"], cfg["n_objects"]),
            "frames_per_s": round(steps / dt, 1), "ms_per_step": round(dt / steps * 1e3, 4), "steps": steps}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=1000)
    ap.add_argument("--warmup", type=int, default=50)
    ap.add_argument("--config", default="C3", help="BASELINE.json config: C1, C2, C3 (default), C5")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-flow-sem", action="store_true", help="render without the flow / semantic outputs")
    ap.add_argument("--no-secondary", action="store_true", help="skip the secondary single-GPU measurements of the other configs")
    args = ap.parse_args()

    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")      # the hosts only support dmabuf IPC: RCCL across processes needs this (set before HIP initialises)
    import torch
    import torch.distributed as dist
    from adgs import _lib, synthetic, dp
    from diff_gaussian_rasterization import GaussianRasterizationSettings, GaussianRasterizer

    if os.environ.get("ADGS_BENCH_WATCHDOG"):              # debugging aid: dump every thread's stack and exit if the run takes longer than N seconds
        import faulthandler
        faulthandler.dump_traceback_later(float(os.environ["ADGS_BENCH_WATCHDOG"]), exit=True)
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus > 1 and world == 1:
        raise SystemExit("--gpus %d needs `python -m torch.distributed.run --nproc-per-node %d bench.py ...`" % (args.gpus, args.gpus))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: the HIP path has no CPU fallback")
    # ADGS_BENCH_BACKEND=gloo: control-flow dry run of the multi-rank path on a box with fewer GPUs than ranks (ranks share devices,
    # collectives go through the host); never a measurement
    backend = os.environ.get("ADGS_BENCH_BACKEND", "nccl")
    dev_index = local_rank if backend == "nccl" else local_rank % torch.cuda.device_count()
    torch.cuda.set_device(dev_index)
    device = torch.device("cuda", dev_index)
    force_coll = world == 1 and os.environ.get("ADGS_BENCH_FORCE_COLLECTIVES") == "1"   # 1-GPU dry run of the N-GPU step: the
    if world > 1 or force_coll:                                                          # collectives run in a one-rank RCCL group
        if force_coll:
            os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29571")
        if backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=device)   # nccl == RCCL on ROCm
        else:
            dist.init_process_group(backend, rank=rank, world_size=world)
    if _lib.lib().adgs_device_check() != 0:
        raise SystemExit(_lib.last_error())

    cfg = synthetic.CONFIGS[args.config]
    sc = synthetic.make_config_scene(args.config)
    cam = synthetic.make_camera(cfg["W"], cfg["H"], cfg["focal"], cam_seed=None if world == 1 else rank)
    H, W, P = cfg["H"], cfg["W"], cfg["P"]
    use_fs = not args.no_flow_sem
    d = lambda t: t.to(device)
    settings = GaussianRasterizationSettings(H, W, cam["tanfovx"], cam["tanfovy"], d(sc["bg"]), 1.0, d(cam["viewmatrix"]),
                                             d(cam["projmatrix"]), cfg["sh_degree"], d(cam["campos"]), False, True, False)
    rasterizer = GaussianRasterizer(settings)
    frame = DeformFrame(sc, rasterizer, device, use_fs) if cfg["n_objects"] > 0 else StaticFrame(sc, rasterizer, device, use_fs)
    up = synthetic.make_upstream_grads(sc, 0)
    up_list = [d(up["color"]), d(up["depth"]), d(up["img_opacity"])] + ([d(up["flow"]), d(up["semantic"])] if use_fs else [])

    # Gradient exchange of the multi-GPU step.  Default on the raw-SH path: the SH gradients travel in
    # factored form (one all-gather of 12 B per Gaussian and camera + a dense all-reduce of the rest + a local expansion);
    # ADGS_DP_EXCHANGE=dense all-reduces every materialised gradient instead.  ADGS_BENCH_FACTORED=1 runs the factored
    # step on one GPU as well (one camera; measures the backward without SH rows + the expansion).
    factored = (isinstance(frame, DeformFrame) and frame.model.raw_sh and os.environ.get("ADGS_DP_EXCHANGE", "factored") != "dense"
                and (world > 1 or force_coll or os.environ.get("ADGS_BENCH_FACTORED") == "1"))
    exchange = "none"
    if factored:
        ex = dp.FactoredSHExchange(frame.model, factor_xyz=(frame.fused_flow or not use_fs) and os.environ.get("ADGS_DP_FACTOR_XYZ", "1") != "0")
        ex.force_collectives = force_coll
        flow_times = [frame.t + 0.05 if use_fs else None] * world
        cam_times = [frame.t] * world
        cam_positions = [synthetic.make_camera(cfg["W"], cfg["H"], cfg["focal"], cam_seed=None if world == 1 else r)["campos"].tolist() for r in range(world)]
        exchange = "factored SH%s gradients: all-gather of the factors + dense all-reduce of the rest + local expansion" % (" and xyz-deformation" if ex.factor_xyz() else "")
    elif world > 1:
        exchange = "dense all-reduce of every parameter gradient"

    def step():
        if factored:
            ex.begin(world)                      # the all-gather starts from inside the backward, as soon as the factor exists
            outs = frame.forward(sink_for=ex.sink_for)
            torch.autograd.backward(outs, up_list)
            ex.reduce(cam_times, cam_positions, flow_times=flow_times)
        else:
            outs = frame.forward()
            torch.autograd.backward(outs, up_list)
            if world > 1 or force_coll:
                dp.allreduce_gradients(frame.parameters(), force=force_coll)
        frame.zero_grad()

    # Settle phase (setup, not a measurement): a fresh process on a fresh box shows one-off stalls of 0.1-0.6 s in its first
    # second (driver / allocator / clock ramp -- seen as a single 200+ ms step right after another GPU process exited).
    # Windows of 20 untimed steps run until two consecutive windows agree within 10 % (at most 8 windows), then the contract's
    # W warm-up steps and K timed steps follow unchanged.
    if os.environ.get("ADGS_BENCH_SETTLE", "1") != "0":
        # With several ranks every step contains collectives, so all ranks must run the SAME number of windows: the window time
        # every rank decides on is the maximum over the ranks, and the bound is a window count, not a per-rank clock.
        prev_w, agree = None, 0
        for _ in range(8):
            t_w = time.perf_counter()
            for _ in range(20):
                step()
            torch.cuda.synchronize()
            w = time.perf_counter() - t_w
            if world > 1:
                tw = torch.tensor([w], device=device, dtype=torch.float64)
                dist.all_reduce(tw, op=dist.ReduceOp.MAX)
                w = float(tw.item())
            agree = agree + 1 if (prev_w is not None and abs(w - prev_w) <= 0.1 * min(w, prev_w)) else 0
            prev_w = w
            if agree >= 2:
                break
    # Stage breakdown (all stages timed with HIP events) on the last warm-up steps; the timed region below only keeps
    # the events around the dominant kernel, because every timed stage leaves a ~10 us bubble in the queue.
    stages_all, dom = None, None
    if args.warmup > 0:
        n_prof = min(args.warmup, 2)              # the last warm-up steps (allocator and caches already warm)
        for _ in range(args.warmup - n_prof):
            step()
        torch.cuda.synchronize()
        wprof = _lib.StageProfiler()
        wprof.enable(True)
        for _ in range(n_prof):
            step()
        torch.cuda.synchronize()
        wprof.enable(False)
        stages_all = wprof.collect()
        dom = max(stages_all, key=lambda k: stages_all[k][0] * max(stages_all[k][1], 1))
    prof = _lib.StageProfiler()
    prof.reserve(2 * args.steps * (1 if dom else 8) + 64)      # no event creation inside the timed region
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    prof.enable(True, stages=[dom] if dom else None)       # --warmup 0: no breakdown yet, time every stage
    gc.collect(); gc.disable()                              # no collector pauses inside the timed region
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    gc.enable()
    prof.enable(False)
    stages = prof.collect()
    if world > 1:
        t = torch.tensor([elapsed], device=device, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    if rank == 0:
        outs = frame.forward()
        stats = _lib.frame_stats()
        V = int((frame.last_radii > 0).sum().item())
        Rc = stats["num_rendered"]                # pairs that were sorted: (cell, Gaussian) in v2, (tile, Gaussian) in classic mode
        X, T = H * W, stats["tiles"]
        M = sc["shs"].shape[1]
        F, D_S = (1, 1) if use_fs else (0, 0)
        v2 = _lib.lib().adgs_raster_needs_zero_init(D_S) == 0
        extra = {}
        if v2:
            # scene-level work figures: the reference's pair count R and what v2 actually blends
            if os.environ.get("ADGS_BENCH_SKIP_STATS"):           # PMC passes: keep foreign launches out of the counter totals
                E, R_ref, E_pub = 0, 0, 0
            else:
                try:
                    E, R_ref, E_pub = blended_entries_and_reference_pairs(frame, sc, settings, use_fs, device)
                except Exception as exc:                 # statistics only
                    print("bench: scene statistics failed: %r" % (exc,), file=sys.stderr)
                    E, R_ref, E_pub = 0, 0, 0
            ab = alg_bytes_v2(P, V, Rc, E, X, T, M, F, D_S, stats["sort_passes"], 0, E_pub)
            extra = {"pipeline": "v2 (coarse cells + lazy per-tile filtering)", "reference_pairs_R": R_ref, "R_over_P": round(R_ref / max(P, 1), 2),
                     "cell_pairs_sorted": Rc, "fine_pairs_bound": stats["fine_pairs"], "blended_entries": E, "published_entries": E_pub,
                     "mean_entries_per_tile": round(E / max(T, 1), 1)}
        else:
            ab = alg_bytes(P, V, Rc, X, T, M, F, D_S, stats["sort_passes"])
            extra = {"pipeline": "classic (reference stage order)", "reference_pairs_R": Rc, "R_over_P": round(Rc / max(P, 1), 2)}
        frame_bytes = sum(ab.values()) + frame.deform_bytes
        # the same frame priced with the REFERENCE algorithm's bytes (every (tile, Gaussian) pair is duplicated,
        # sorted and streamed through the blend kernels) -- what the north star's "fraction of the HBM roofline" refers to
        ref_bytes = sum(alg_bytes(P, V, extra.get("reference_pairs_R", Rc), X, T, M, F, D_S, (32 + max(T - 1, 1).bit_length() + 7) // 8).values()) + frame.deform_bytes
        if stages_all is None:
            stages_all = stages
            dom = max(stages, key=lambda k: stages[k][0] * max(stages[k][1], 1))
        dom_ms = stages[dom][0]                   # HIP events on the launch stream, over the timed region
        achieved = ab.get(dom, 0) / (dom_ms * 1e-3) / 1e9 if dom_ms > 0 else 0.0
        fps = args.steps * world / elapsed
        roof = {"bound": "hbm", "kernel": dom, "achieved": round(achieved, 2), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                "frac": round(achieved / HBM_PEAK_GBS, 5), "traffic": None,
                "alg_bytes_per_launch": int(ab.get(dom, 0)), "avg_launch_ms": round(dom_ms, 4)}
        roof.update(pmc_annotations(dom, args.config, use_fs and v2))
        if v2 and dom in ("render_fwd", "render_bwd"):
            roof["pixel_entry_evals_per_s"] = round((extra["published_entries"] if dom == "render_bwd" else extra["blended_entries"]) * 256 / (dom_ms * 1e-3), 1)
        result = {
            "metric": "fwd+bwd frames/s",
            "value": round(fps, 3), "unit": "frames/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(elapsed / args.steps * 1e3, 4), "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": dict({"workload": "%s: %d Gaussians, %dx%d, SH deg %d, %d dynamic objects%s, 1 camera/GPU/step" % (
                args.config, P, W, H, cfg["sh_degree"], cfg["n_objects"], ", flow+semantic outputs" if use_fs else ""),
                "P": P, "P_visible": V, "tiles": T, "deformation": frame.deform_desc,
                "parallelism": "dp%d (camera-parallel over RCCL)" % world if world > 1 else "single GPU", "gradient_exchange": exchange,
                "alg_bytes_per_frame": int(frame_bytes),
                "frame_hbm_frac_of_8TBs": round(frame_bytes * fps / world / 1e9 / HBM_PEAK_GBS, 4),
                "reference_alg_bytes_per_frame": int(ref_bytes),
                "reference_alg_hbm_frac_of_8TBs": round(ref_bytes * fps / world / 1e9 / HBM_PEAK_GBS, 4)}, **measured_frame_traffic(args.config, use_fs and v2, fps / world), **extra),
            "roofline": roof,
            "stages_ms": {k: round(v[0], 4) for k, v in stages_all.items()},    # all stages timed: from the last warm-up steps
            # every stage against the HBM roofline (algorithmic bytes of DESIGN.md section 5 / its HIP-event time in the last warm-up
            # steps): the streaming stages sit at 45-70 % of the 8 TB/s peak, the two blend stages are VALU-bound (see "roofline")
            "stage_rooflines": {k: {"alg_bytes": int(ab.get(k, 0)), "GB/s": round(ab.get(k, 0) / (v[0] * 1e-3) / 1e9, 1),
                                    "frac_of_8TBs": round(ab.get(k, 0) / (v[0] * 1e-3) / 1e9 / HBM_PEAK_GBS, 4)}
                                for k, v in stages_all.items() if v[0] > 0},
        }
        if world == 1 and not force_coll and not args.no_secondary and args.config == "C3":
            # the other single-GPU configs of BASELINE.json, same build, same process (secondary: `value` stays the C3 headline)
            try:
                result["other_configs"] = [quick_measure(c, 300, device, use_fs) for c in ("C1", "C2")]
            except Exception as exc:                     # secondary information must never cost the headline line
                result["other_configs"] = "failed: %r" % (exc,)
        if world == 1 and not args.no_cpu_baseline:
            if isinstance(frame, DeformFrame):
                pkg, flow = frame.activated()
                sc_cpu = dict(sc, means3D=pkg["xyz"], opacities=pkg["opacity"], scales=pkg["scales"], rotations=pkg["rotation"],
                              shs=pkg["shs"], flow_points=flow)
            else:
                sc_cpu = sc
            try:
                result["cpu_baseline"], oracle_fwd = cpu_baseline(sc_cpu, cam, cfg, use_fs, up)
                result["parity"] = parity_vs_oracle(outs, oracle_fwd)
            except Exception as exc:                     # e.g. the oracle library could not be built on this host
                result["cpu_baseline"] = {"value": None, "unit": "frames/s", "cores": 0, "kind": "port", "sample": "failed: %r" % (exc,)}
        # RCCL prints a version banner through C stdio, which (on a pipe) would only be flushed at exit, i.e. AFTER the JSON
        # line: flush it first so that the JSON line is the last line of stdout
        try:
            import ctypes
            ctypes.CDLL(None).fflush(None)
        except OSError:
            pass
        sys.stdout.flush()
        print(json.dumps(result), flush=True)
    if world > 1 or force_coll:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
