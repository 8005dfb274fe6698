#!/usr/bin/env python3
"""cProfile of BOTH host threads of a bench.py step: the caller's (forward, zero_grad) and the autograd engine's worker thread, where the
Python side of every backward runs (a profiler sees only the thread that enabled it: tools/host_profile.py shows the backward as one
opaque `run_backward`).  A probe node on the first output switches the worker thread's profiler on from inside the backward pass.

    python tools/host_profile_threads.py [C3] [steps]        (GPU box)
"""
import cProfile, io, os, pstats, sys, threading, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "ad-gs_amd")):
    if p not in sys.path:
        sys.path.insert(0, p)
import torch

PROFS, ON = {}, {}


class Probe(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, stop):
        ctx.stop = stop
        return x.view_as(x)

    @staticmethod
    def backward(ctx, g):
        tid = threading.get_ident()
        pr = PROFS.setdefault(tid, cProfile.Profile())
        if ctx.stop and ON.get(tid):
            pr.disable(); ON[tid] = False
        elif not ctx.stop and not ON.get(tid):
            pr.enable(); ON[tid] = True
        return g, None


def main():
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

    def step(stop=False):
        outs = frame.forward()
        outs[0] = Probe.apply(outs[0], stop)
        torch.autograd.backward(outs, ups)
        frame.zero_grad()
    for _ in range(30):
        step(True)
    torch.cuda.synchronize()
    main_pr = cProfile.Profile()
    t0 = time.perf_counter()
    main_pr.enable()
    for _ in range(steps):
        step()
    main_pr.disable()
    step(True)
    torch.cuda.synchronize()
    print("%s: %.3f ms per step under both profilers" % (config, (time.perf_counter() - t0) / (steps + 1) * 1e3))
    for name, pr in [("caller thread", main_pr)] + [("autograd worker %d" % i, p) for i, p in enumerate(PROFS.values())]:
        for key, n in (("tottime", 30), ("cumulative", 30)):
            s = io.StringIO()
            try:
                pstats.Stats(pr, stream=s).sort_stats(key).print_stats(n)
            except TypeError:
                continue
            print("==== %s, by %s (per step: divide by %d)" % (name, key, steps)); print(s.getvalue()[:7000])


if __name__ == "__main__":
    main()
