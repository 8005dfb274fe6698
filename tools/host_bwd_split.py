#!/usr/bin/env python3
"""Where the host time of the rasterizer's backward goes, unprofiled: wall time of the Python binder (`_C.rasterize_gaussians_backward_rawsh`)
and, inside it, of the native call -- the difference is the Python prologue (allocations, pointer marshalling, the ctypes structs) that sits
between the loss kernels and `render_bwd_v2`'s launch.  Same for the forward (the native call includes the wait for the device's totals).

    python tools/host_bwd_split.py [C3] [steps]        (GPU box)
"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "ad-gs_amd")):
    if p not in sys.path:
        sys.path.insert(0, p)
import torch


class Timed:
    def __init__(self, fn):
        self.fn, self.t, self.n = fn, 0.0, 0

    def __call__(self, *a, **k):
        t0 = time.perf_counter()
        try:
            return self.fn(*a, **k)
        finally:
            self.t += time.perf_counter() - t0; self.n += 1

    def us(self):
        return 1e6 * self.t / max(self.n, 1)


class LibProxy:
    """Attribute access falls through to the ctypes library; the wrapped entry points are timed."""
    def __init__(self, lib, names):
        object.__setattr__(self, "_lib", lib)
        object.__setattr__(self, "_timed", {n: Timed(getattr(lib, n)) for n in names})

    def __getattr__(self, n):
        t = self._timed.get(n)
        return t if t is not None else getattr(self._lib, n)


def main():
    import bench
    from adgs import synthetic, _lib
    from diff_gaussian_rasterization import _C
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
    native = ("adgs_raster_forward_rawsh", "adgs_raster_backward_rawsh", "adgs_raster_forward", "adgs_raster_backward", "adgs_deform_forward_flow", "adgs_deform_backward_flow",
              "adgs_deform_forward", "adgs_deform_backward")
    real = _lib.lib()
    proxy = LibProxy(real, [n for n in native if hasattr(real, n)])
    _lib._lib = proxy                                                   # what _lib.lib() hands out
    binders = {}
    for name in ("rasterize_gaussians_rawsh", "rasterize_gaussians_backward_rawsh", "rasterize_gaussians", "rasterize_gaussians_backward"):
        binders[name] = Timed(getattr(_C, name)); setattr(_C, name, binders[name])
    t0 = time.perf_counter()
    for _ in range(steps):
        step()
    host = time.perf_counter() - t0
    torch.cuda.synchronize()
    total = time.perf_counter() - t0
    print("%s: %.1f us per step of host time, %.1f us per step with the GPU drained" % (config, 1e6 * host / steps, 1e6 * total / steps))
    for name, t in binders.items():
        if t.n:
            print("  binder %-38s %7.1f us per call (%d calls)" % (name, t.us(), t.n))
    for name, t in proxy._timed.items():
        if t.n:
            print("  native %-38s %7.1f us per call (%d calls)" % (name, t.us(), t.n))


if __name__ == "__main__":
    main()
