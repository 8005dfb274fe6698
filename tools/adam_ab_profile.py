#!/usr/bin/env python3
"""Where the HOST time of the backward goes in the training iteration, in-backward Adam against the separate step, alternated in one process:
wall time per call of the autograd nodes' backward functions, of the optimizer's claim and of the native rasterizer / deformation calls.
(tools/adam_ab.py showed runs whose backward takes 1 ms more HOST time with identical kernels.)  python tools/adam_ab_profile.py [rounds] [iters]"""
import gc, importlib.util, json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
spec = importlib.util.spec_from_file_location("train_iteration", os.path.join(ROOT, "examples", "train_iteration.py"))
ti = importlib.util.module_from_spec(spec); spec.loader.exec_module(ti)
import torch
from adgs import _lib, deform, loss, env, optim
import diff_gaussian_rasterization as dgr
from diff_gaussian_rasterization import _C


class Timed:
    def __init__(self, fn): self.fn, self.t, self.n, self.worst = fn, 0.0, 0, 0.0
    def __call__(self, *a, **k):
        t0 = time.perf_counter()
        try: return self.fn(*a, **k)
        finally:
            d = time.perf_counter() - t0; self.t += d; self.n += 1; self.worst = max(self.worst, d)
    def take(self):
        r = (round(1e6 * self.t / max(self.n, 1), 1), self.n, round(1e6 * self.worst, 1)); self.t, self.n, self.worst = 0.0, 0, 0.0
        return r


class LibProxy:
    def __init__(self, lib, names):
        object.__setattr__(self, "_lib", lib); object.__setattr__(self, "_timed", {n: Timed(getattr(lib, n)) for n in names if hasattr(lib, n)})
    def __getattr__(self, n):
        t = self._timed.get(n); return t if t is not None else getattr(self._lib, n)


timers = {}
real = _lib.lib()
proxy = LibProxy(real, ["adgs_raster_backward_rawsh", "adgs_deform_backward_flow", "adgs_envmap_backward", "adgs_adam_step", "adgs_raster_forward_rawsh"])
_lib._lib = proxy
for n, t in proxy._timed.items(): timers["native " + n] = t
for cls, name in ((dgr._RasterizeGaussiansRawSH, "raster"), (deform._DeformPkgFn, "deform")):
    t = Timed(cls.backward); cls.backward = staticmethod(t); timers["node %s.backward" % name] = t
for modname, mod in (("loss", loss), ("env", env)):
    for k, v in list(vars(mod).items()):
        if isinstance(v, type) and issubclass(v, torch.autograd.Function) and v is not torch.autograd.Function:
            t = Timed(v.backward); v.backward = staticmethod(t); timers["node %s.%s.backward" % (modname, k)] = t
t_claim = Timed(optim.BackwardEpilogue.claim); optim.BackwardEpilogue.claim = lambda self, *a, **k: t_claim(self, *a, **k); timers["BackwardEpilogue.claim"] = t_claim
import cProfile, pstats, io
PROF = [cProfile.Profile()]
_orig_bind = _C.rasterize_gaussians_backward_rawsh
def _profiled_bind(*a, **k):      # runs on the autograd worker thread: the profiler is enabled there, around the binder only
    pr = PROF[0]; pr.enable()
    try: return _orig_bind(*a, **k)
    finally: pr.disable()
t_bind = Timed(_profiled_bind); _C.rasterize_gaussians_backward_rawsh = t_bind; timers["binder rasterize_gaussians_backward_rawsh"] = t_bind

if os.environ.get("ADGS_AB_NO_GC"):
    gc.disable()
rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 3
iters = int(sys.argv[2]) if len(sys.argv) > 2 else 60
dev = torch.device("cuda", 0)
for r in range(rounds):
    for mode in (True, False):
        cfg, model, cams, env_map = ti.build("C3", 8192, dev, 16, mode)
        state, off = {}, ti.StageClock(False)
        for i in range(12):
            ti.iteration(i, model, cams, env_map, off, state)
        torch.cuda.synchronize()
        for t in timers.values(): t.take()
        clock = ti.StageClock(True)
        t0 = time.perf_counter()
        for i in range(12, 12 + iters):
            ti.iteration(i, model, cams, env_map, clock, state)
        torch.cuda.synchronize()
        ms = 1e3 * (time.perf_counter() - t0) / iters
        took = {k: v.take() for k, v in timers.items() if v.n}
        print(json.dumps({"adam_in_backward": mode, "gc": gc.isenabled(), "ms": round(ms, 4), "host_backward_ms": clock.host_summary().get("backward"),
                          "binder_worst_call_us": took.get("binder rasterize_gaussians_backward_rawsh", (0, 0, 0))[2], "gc_collections": [g["collections"] for g in gc.get_stats()],
                          "us_per_call": {k: v[0] for k, v in took.items()}}), flush=True)
        if took.get("binder rasterize_gaussians_backward_rawsh", (0,))[0] > 500:      # a slow run: where inside the binder?
            st = io.StringIO(); pstats.Stats(PROF[0], stream=st).sort_stats("tottime").print_stats(8)
            print("\n".join(l[:150] for l in st.getvalue().splitlines() if ("{" in l or ".py" in l) and "function calls" not in l), flush=True)
        PROF[0] = cProfile.Profile()
        del model, cams, env_map, state
        torch.cuda.empty_cache()
