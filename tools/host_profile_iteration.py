#!/usr/bin/env python3
"""Where the HOST time of a training iteration goes: cProfile over examples/train_iteration.iteration() (C3, in-backward Adam),
plus the un-profiled host time per iteration with the GPU drained after every iteration (pure enqueue cost) and free-running.
python tools/host_profile_iteration.py [iters]"""
import cProfile, importlib.util, io, os, pstats, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
spec = importlib.util.spec_from_file_location("train_iteration", os.path.join(ROOT, "examples", "train_iteration.py"))
ti = importlib.util.module_from_spec(spec); spec.loader.exec_module(ti)
import torch
iters = int(sys.argv[1]) if len(sys.argv) > 1 else 100
dev = torch.device("cuda", 0)
cfg, model, cams, env_map = ti.build("C3", 8192, dev, 16, True)
state, off = {}, ti.StageClock(False)
for i in range(14):
    ti.iteration(i, model, cams, env_map, off, state)
torch.cuda.synchronize()
# host enqueue time per iteration with an idle GPU in front of every iteration
t_host = []
for i in range(14, 14 + 40):
    if (i + 1) % ti.OPT.near_idx_reset_interval == 0:
        ti.iteration(i, model, cams, env_map, off, state); torch.cuda.synchronize(); continue
    t0 = time.perf_counter(); ti.iteration(i, model, cams, env_map, off, state); t_host.append(time.perf_counter() - t0); torch.cuda.synchronize()
t_host.sort()
print("host enqueue per iteration (GPU drained before each): median %.3f ms, p10 %.3f, p90 %.3f" % (
    1e3 * t_host[len(t_host) // 2], 1e3 * t_host[len(t_host) // 10], 1e3 * t_host[9 * len(t_host) // 10]))
t0 = time.perf_counter()
for i in range(54, 54 + iters):
    ti.iteration(i, model, cams, env_map, off, state)
torch.cuda.synchronize()
print("free-running: %.3f ms per iteration" % (1e3 * (time.perf_counter() - t0) / iters))
pr = cProfile.Profile(); pr.enable()
for i in range(54 + iters, 54 + 2 * iters):
    ti.iteration(i, model, cams, env_map, off, state)
pr.disable(); torch.cuda.synchronize()
for key in ("tottime", "cumulative"):
    s = io.StringIO(); pstats.Stats(pr, stream=s).sort_stats(key).print_stats(45); print(s.getvalue()[:9000])
