#!/usr/bin/env python3
"""Host (enqueue) time of one bench.py step at a config: wall time of step() while the GPU queue is never waited on, the first
step after a synchronize, and a cProfile of the Python side.  `python tools/host_profile.py [C3] [steps]` on a GPU box."""
import cProfile
import io
import os
import pstats
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "ad-gs_amd")):
    if p not in sys.path:
        sys.path.insert(0, p)


def main():
    import torch
    import bench
    from adgs import synthetic
    config = sys.argv[1] if len(sys.argv) > 1 else "C3"
    steps = int(sys.argv[2]) if len(sys.argv) > 2 else 200
    dev = torch.device("cuda", 0)
    cfg = synthetic.CONFIGS[config]
    sc = synthetic.make_config_scene(config)
    cam = synthetic.make_camera(cfg["W"], cfg["H"], cfg["focal"])
    use_fs = cfg["n_objects"] > 0
    frame = bench.make_frame(sc, cfg, cam, dev, use_fs)
    up = synthetic.make_upstream_grads(sc, 0)
    ups = [up[k].to(dev) for k in (("color", "depth", "img_opacity") + (("flow", "semantic") if use_fs else ()))]

    def step():
        torch.autograd.backward(frame.forward(), ups)
        frame.zero_grad()
    for _ in range(30):
        step()
    torch.cuda.synchronize()
    host = []
    t_all = time.perf_counter()
    for _ in range(steps):
        t0 = time.perf_counter(); step(); host.append((time.perf_counter() - t0) * 1e3)
    t_enq = time.perf_counter() - t_all
    torch.cuda.synchronize()
    t_tot = time.perf_counter() - t_all
    host.sort()
    print("%s: host enqueue per step median %.3f ms (p10 %.3f, p90 %.3f), loop %.3f ms/step enqueue, %.3f ms/step with the GPU drained" % (
        config, host[len(host) // 2], host[len(host) // 10], host[len(host) * 9 // 10], t_enq / steps * 1e3, t_tot / steps * 1e3))
    # forward / backward split on an idle GPU
    f, b = [], []
    for _ in range(20):
        torch.cuda.synchronize()
        t0 = time.perf_counter(); outs = frame.forward(); t1 = time.perf_counter()
        torch.autograd.backward(outs, ups); frame.zero_grad(); t2 = time.perf_counter()
        f.append((t1 - t0) * 1e3); b.append((t2 - t1) * 1e3)
    f.sort(); b.sort()
    print("idle GPU: forward enqueue %.3f ms, backward + zero_grad enqueue %.3f ms (medians)" % (f[10], b[10]))
    pr = cProfile.Profile()
    torch.cuda.synchronize()
    pr.enable()
    for _ in range(steps):
        step()
    pr.disable()
    torch.cuda.synchronize()
    s = io.StringIO()
    pstats.Stats(pr, stream=s).sort_stats("tottime").print_stats(28)
    print(s.getvalue())
    s = io.StringIO()
    pstats.Stats(pr, stream=s).sort_stats("cumulative").print_stats(40)
    print(s.getvalue())


if __name__ == "__main__":
    main()
