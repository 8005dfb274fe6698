#!/usr/bin/env python3
"""Why rocprofv3's kernel-trace duration of the streaming stages exceeds their HIP-event brackets.

The preprocess kernel reads what the kernels in front of it wrote microseconds earlier (deformed parameters, sh0: ~100 MB) -- in the
undisturbed frame that comes out of the 256 MiB Infinity Cache.  Under rocprofv3 --kernel-trace every dispatch is intercepted and
serialised, and the data is gone by the time the preprocess runs.  This tool times the SAME launch sequence with HIP events, (a) as
is and (b) with a fill of N MiB between the sh0 kernel and the preprocess (ADGS_DBG_EVICT_MB, api.hip): if (b) reproduces the
profiler's figure, the profiler's figure is the cold-cache one and the event bracket the warm one -- both are true.

    python tools/stage_cache_experiment.py [C3] > profiles/r04/stage_cache_experiment.json
"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "ad-gs_amd")):
    sys.path.insert(0, p)


def main():
    import torch
    import bench
    from adgs import _lib, synthetic
    config = sys.argv[1] if len(sys.argv) > 1 else "C3"
    device = torch.device("cuda", 0)
    cfg = synthetic.CONFIGS[config]
    sc = bench.build_scene(config)
    cam = synthetic.make_camera(cfg["W"], cfg["H"], cfg["focal"])
    frame = bench.make_frame(sc, cfg, cam, device, True)
    up = synthetic.make_upstream_grads(sc, 0)
    ups = [up[k].to(device) for k in ("color", "depth", "img_opacity", "flow", "semantic")]

    def step():
        torch.autograd.backward(frame.forward(), ups)
        frame.zero_grad()

    out = {"config": config, "note": "HIP-event stage times (ms per step, 40 steps each) of the same launch sequence; evict_mb = size of a fill between sh0 and the preprocess"}
    for evict in (0, 128, 512, 0):
        if evict:
            os.environ["ADGS_DBG_EVICT_MB"] = str(evict)
        else:
            os.environ.pop("ADGS_DBG_EVICT_MB", None)
        for _ in range(15):
            step()
        torch.cuda.synchronize()
        prof = _lib.StageProfiler()
        prof.reserve(64 * 40)
        prof.enable(True)
        for _ in range(40):
            step()
        torch.cuda.synchronize()
        prof.enable(False)
        st = prof.collect()
        key = "evict_%d_mb%s" % (evict, "_again" if ("evict_0_mb" in out and evict == 0) else "")
        out[key] = {k: round(v[0] * v[1] / 40.0, 4) for k, v in st.items() if v[1]}
    os.environ.pop("ADGS_DBG_EVICT_MB", None)
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
