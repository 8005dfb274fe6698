#!/usr/bin/env python3
"""Per-wave timeline of the two blend kernels (experiment build: make -C ad-gs_amd/csrc variant TAG=timeline DEFS=-DADGS_TIMELINE):
every wave (= tile) stores its start and end in 100 MHz ticks; this script turns them into the distribution of wave lives, the number
of waves in flight over the kernel's duration and the share of the kernel during which the chip was less than half full.

    ADGS_LIB=ad-gs_amd/lib/libadgs_hip_timeline.so python tools/wave_timeline.py [C3]
"""
import ctypes, json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "ad-gs_amd")):
    sys.path.insert(0, p)
import numpy as np
import torch
import bench
from adgs import _lib, synthetic


def summarize(start, end, name):
    start = start.astype(np.int64); end = end.astype(np.int64)
    end = np.where(end < start, end + (1 << 32), end)
    t0 = start.min()
    s, e = (start - t0) * 0.01, (end - t0) * 0.01            # microseconds
    life = e - s
    dur = e.max()
    grid = np.linspace(0, dur, 201)
    alive = np.array([((s <= t) & (e > t)).sum() for t in grid])
    return {"kernel": name, "waves": int(len(s)), "kernel_us": round(float(dur), 1), "wave_life_us": {k: round(float(np.percentile(life, q)), 1) for k, q in
            (("p10", 10), ("p50", 50), ("p90", 90), ("max", 100))}, "sum_of_wave_lives_over_kernel_time": round(float(life.sum() / dur), 1),
            "waves_in_flight": {"max": int(alive.max()), "mean": round(float(alive.mean()), 1), "at_25_50_75_90_pct_of_kernel": [int(alive[i]) for i in (50, 100, 150, 180)]},
            "last_start_us": round(float(s.max()), 1), "share_of_kernel_below_half_of_max_in_flight": round(float((alive < alive.max() / 2).mean()), 3)}


def main():
    config = sys.argv[1] if len(sys.argv) > 1 else "C3"
    device = torch.device("cuda", 0)
    cfg = synthetic.CONFIGS[config]
    sc = bench.build_scene(config)
    cam = synthetic.make_camera(cfg["W"], cfg["H"], cfg["focal"])
    frame = bench.make_frame(sc, cfg, cam, device, True)
    up = synthetic.make_upstream_grads(sc, 0)
    ups = [up[k].to(device) for k in ("color", "depth", "img_opacity", "flow", "semantic")]
    lib = _lib.lib()
    out = []
    for which in ("fwd", "bwd"):
        os.environ["ADGS_TIMELINE_BWD"] = "1" if which == "bwd" else "0"
        for _ in range(5):
            torch.autograd.backward(frame.forward(), ups); frame.zero_grad()
        # the image-state buffer of a forward through the plain API (bench.frame_work_figures does the same)
        from diff_gaussian_rasterization import _C
        from adgs import deform
        with torch.no_grad():
            pkg = deform.get_deformed_pkg(frame.model, frame.t) if hasattr(frame, "model") else None
        s = bench.make_settings(cfg, cam, sc, device)
        e = torch.empty(0, device=device)
        if pkg is not None:
            t = dict(means3D=pkg["xyz"], opacities=pkg["opacity"], scales=pkg["scales"], rotations=pkg["rotation"], shs=pkg["shs"])
            flow = frame.model.get_deformed_xyz(frame.t + 0.05)
        else:
            t = {k: v.detach() for k, v in frame.leaf.items()}; flow = frame.flow
        sem = frame.sem
        leafs = {k: v.detach().clone().requires_grad_(True) for k, v in t.items()}
        r = _C.rasterize_gaussians(s.bg, leafs["means3D"], e, leafs["opacities"], leafs["scales"], leafs["rotations"], s.scale_modifier, e, s.viewmatrix, s.projmatrix,
                                   s.tanfovx, s.tanfovy, s.image_height, s.image_width, leafs["shs"], flow, sem, s.sh_degree, s.campos, s.prefiltered, s.inv_depth, False)
        if which == "bwd":
            _C.rasterize_gaussians_backward(s.bg, leafs["means3D"], r[4], e, leafs["scales"], leafs["rotations"], s.scale_modifier, e, s.viewmatrix, s.projmatrix, s.tanfovx,
                                            s.tanfovy, ups[0], ups[1], ups[3], ups[4], sem, flow, leafs["shs"], s.sh_degree, s.campos, r[5], r[0], r[6], r[7], r[3], ups[2],
                                            s.inv_depth, False)
        torch.cuda.synchronize()
        n = int(lib.adgs_test_v2_tile_counters(r[7].data_ptr(), int(s.image_width), int(s.image_height), None, None, 0, None))
        a = (ctypes.c_uint32 * n)(); b = (ctypes.c_uint32 * n)()
        # tile_counters hands back (consumed, scanned); the timeline build keeps start in `scanned`, end in `batches`: read both arrays raw
        lib.adgs_test_v2_tile_words.restype = ctypes.c_longlong
        lib.adgs_test_v2_tile_words.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_longlong, ctypes.c_void_p]
        lib.adgs_test_v2_tile_words(r[7].data_ptr(), int(s.image_width), int(s.image_height), a, b, n, None)
        out.append(summarize(np.frombuffer(a, np.uint32).copy(), np.frombuffer(b, np.uint32).copy(), "render_%s_v2" % which))
    print(json.dumps({"config": config, "timelines": out}, indent=1))


if __name__ == "__main__":
    main()
