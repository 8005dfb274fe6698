"""Per-frame view of the bucket binning: which path a frame took, its fullest depth slab (in units of 4096 entries) and the HIP-event time of its binning stages.
    python tools/slab_trace.py C5 [frames]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "ad-gs_amd")):
    sys.path.insert(0, p)
import torch
import bench
from adgs import _lib, synthetic

config = sys.argv[1] if len(sys.argv) > 1 else "C3"
frames = int(sys.argv[2]) if len(sys.argv) > 2 else 12
dev = torch.device("cuda", 0)
cfg = synthetic.CONFIGS[config]
sc = bench.build_scene(config)
pool = bench.camera_pool(cfg, 4)
fr = bench.frame_pool(sc, cfg, pool, dev, True)
prof = _lib.StageProfiler(); prof.reserve(64 * frames)
for i in range(frames):
    prof.enable(True)
    with torch.no_grad():
        fr[i % len(fr)].forward()
    torch.cuda.synchronize()
    prof.enable(False)
    st = prof.collect(); prof.total = type(prof.total)(); prof.count = type(prof.count)()
    fs, stats = _lib.frame_status(), _lib.frame_stats()
    print("frame %2d cam %d bucket=%d pairs=%d fullest_slab_units=%d  scan %.3f scatter %.3f sort %.3f ranges %.3f fwd %.3f ms" % (
        i, i % len(fr), stats.get("bucket_binning", -1), fs["pairs"], fs["fullest_slab_units"],
        st["scan"][0] * st["scan"][1], st["duplicate_keys"][0] * st["duplicate_keys"][1], st["radix_sort"][0] * st["radix_sort"][1],
        st["tile_ranges"][0] * st["tile_ranges"][1], st["render_fwd"][0] * st["render_fwd"][1]), flush=True)
