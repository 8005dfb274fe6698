"""Per-step host timestamps of the bench frame WITHOUT per-step synchronisation (the forward's mailbox wait couples host and GPU
once per frame anyway): where do slow steps sit in a fresh process, and how long are they?"""
import sys, time, os, torch
T_START = time.perf_counter()
sys.path.insert(0, '/root/repo'); sys.path.insert(0, '/root/repo/ad-gs_amd')
import bench
from adgs import synthetic
from diff_gaussian_rasterization import GaussianRasterizationSettings, GaussianRasterizer
cfg = synthetic.CONFIGS["C3"]; sc = synthetic.make_config_scene("C3"); cam = synthetic.make_camera(cfg["W"], cfg["H"], cfg["focal"])
dev = torch.device("cuda", 0); d = lambda t: t.to(dev)
s = GaussianRasterizationSettings(cfg["H"], cfg["W"], cam["tanfovx"], cam["tanfovy"], d(sc["bg"]), 1.0, d(cam["viewmatrix"]), d(cam["projmatrix"]), 3, d(cam["campos"]), False, True, False)
frame = bench.DeformFrame(sc, GaussianRasterizer(s), dev, True)
up = synthetic.make_upstream_grads(sc, 0)
ups = [d(up[k]) for k in ("color", "depth", "img_opacity", "flow", "semantic")]
N = int(sys.argv[1]) if len(sys.argv) > 1 else 4000
import gc
gc.collect()
if os.environ.get('JITTER_NO_GC'): gc.disable()
torch.cuda.synchronize()
t_loop = time.perf_counter()
ts = [t_loop]
for i in range(N):
    outs = frame.forward(); torch.autograd.backward(outs, ups); frame.zero_grad()
    ts.append(time.perf_counter())
    if i % 500 == 0: print('step', i, 'allocated %.2f GB reserved %.2f GB' % (torch.cuda.memory_allocated() / 1e9, torch.cuda.memory_reserved() / 1e9), flush=True)
torch.cuda.synchronize()
t_end = time.perf_counter()
dt = [(ts[i + 1] - ts[i]) * 1e3 for i in range(N)]
sd = sorted(dt)
print("setup %.1f s; %d steps in %.3f s = %.3f ms/step; median %.3f p90 %.3f p99 %.3f max %.3f" % (t_loop - T_START, N, t_end - t_loop, (t_end - t_loop) / N * 1e3,
      sd[N // 2], sd[int(N * .9)], sd[int(N * .99)], sd[-1]))
slow = [(i, round(ts[i] - t_loop, 3), round(x, 2)) for i, x in enumerate(dt) if x > 3.0]
print("steps > 3 ms (index, seconds since loop start, ms):", slow[:40], "count", len(slow), "excess ms total %.1f" % sum(x - sd[N // 2] for x in dt if x > 3.0))
for w in range(0, N, 500):
    seg = dt[w:w + 500]
    print("steps %4d-%4d: mean %.3f ms" % (w, w + len(seg) - 1, sum(seg) / len(seg)))
