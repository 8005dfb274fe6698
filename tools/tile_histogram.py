#!/usr/bin/env python3
"""Per-tile work distribution of one forward frame of a BASELINE config: blended (published) entries and scanned candidates per
16x16 tile -- mean, percentiles, maximum -- i.e. how long the longest one-wave tile walk is against the average (the blend
kernels are one wave per tile: the longest tile bounds the kernel time from below).   python tools/tile_histogram.py C3 [C5] [C3:street]
(config[:variant], the variants of bench.build_scene)"""
import ctypes
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "ad-gs_amd")):
    sys.path.insert(0, p)
import numpy as np
import torch

import bench
from adgs import _lib, synthetic, deform
from diff_gaussian_rasterization import _C

out = {}
for name in sys.argv[1:] or ["C3"]:
    config, _, variant = name.partition(":")
    cfg = synthetic.CONFIGS[config]
    sc = bench.build_scene(config, variant or "default")
    cam = synthetic.make_camera(cfg["W"], cfg["H"], cfg["focal"])
    dev = torch.device("cuda", 0)
    frame = bench.make_frame(sc, cfg, cam, dev, True)
    s = bench.make_settings(cfg, cam, sc, dev)
    with torch.no_grad():
        if hasattr(frame, "model"):
            pkg = deform.get_deformed_pkg(frame.model, frame.t)
            t = dict(means3D=pkg["xyz"], opacities=pkg["opacity"], scales=pkg["scales"], rotations=pkg["rotation"], shs=pkg["shs"])
            flow = frame.model.get_deformed_xyz(frame.t + 0.05)
        else:
            t = {k: v.detach() for k, v in frame.leaf.items()}; flow = frame.flow
        e = torch.empty(0, device=dev)
        r = _C.rasterize_gaussians(s.bg, t["means3D"], e, t["opacities"], t["scales"], t["rotations"], s.scale_modifier, e, s.viewmatrix, s.projmatrix, s.tanfovx,
                                   s.tanfovy, s.image_height, s.image_width, t["shs"], flow, frame.sem, s.sh_degree, s.campos, s.prefiltered, s.inv_depth, False)
    n = 1 << 20
    cons = np.zeros(n, np.uint32); scan = np.zeros(n, np.uint32)
    nt = _lib.lib().adgs_test_v2_tile_counters(r[7].data_ptr(), cfg["W"], cfg["H"], cons.ctypes.data, scan.ctypes.data, n, ctypes.c_void_p(torch.cuda.current_stream(dev).cuda_stream))
    cons, scan = cons[:nt], scan[:nt]
    st = lambda a: dict(mean=float(a.mean()), p50=float(np.percentile(a, 50)), p90=float(np.percentile(a, 90)), p99=float(np.percentile(a, 99)), max=int(a.max()),
                        share_of_top_1pct=float(np.sort(a)[-max(nt // 100, 1):].sum() / max(a.sum(), 1)))
    cr = np.zeros(2 * 4096, np.uint32)
    nc = _lib.lib().adgs_test_v2_cell_ranges(r[7].data_ptr(), cfg["W"], cfg["H"], cr.ctypes.data, 4096, ctypes.c_void_p(torch.cuda.current_stream(dev).cuda_stream))
    cr = cr[:2 * nc].reshape(-1, 2).astype(np.int64)
    csz = cr[:, 1] - cr[:, 0]
    chunks = (csz + 8191) // 8192
    out[name] = dict(tiles=int(nt), blended_entries_per_tile=st(cons), scanned_candidates_per_tile=st(scan),
                     cells=int(nc), candidates_per_cell=dict(mean=float(csz.mean()), p50=float(np.percentile(csz, 50)), p90=float(np.percentile(csz, 90)),
                                                            p99=float(np.percentile(csz, 99)), max=int(csz.max())),
                     chunks_per_cell={str(k): int((chunks == k).sum()) for k in np.unique(chunks)})
    print(name, json.dumps(out[name]))
json.dump(out, open(os.path.join(ROOT, "gpurun_out", "tile_histogram.json"), "w"), indent=1)
