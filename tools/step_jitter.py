import sys, time, os, torch
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
ts = []
for i in range(300):
    t0 = time.perf_counter()
    outs = frame.forward(); torch.autograd.backward(outs, ups); frame.zero_grad()
    torch.cuda.synchronize()
    ts.append((time.perf_counter() - t0) * 1e3)
ts2 = sorted(ts[5:])
print("median %.3f p99 %.3f max %.3f argmax %d first5 %s" % (ts2[len(ts2)//2], ts2[int(len(ts2)*0.99)], max(ts[5:]), ts.index(max(ts[5:])), [round(x,2) for x in ts[:5]]))
print("slow steps:", [(i, round(x,2)) for i, x in enumerate(ts) if i >= 5 and x > 3.0])
