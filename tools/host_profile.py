"""cProfile of the host side of the bench step (C1: GPU time negligible, so wall time ~ host + launch overhead)."""
import cProfile, pstats, sys, os, io, time, torch
sys.path.insert(0, '/root/repo'); sys.path.insert(0, '/root/repo/ad-gs_amd')
import bench
from adgs import synthetic
from diff_gaussian_rasterization import GaussianRasterizationSettings, GaussianRasterizer
cfgname = sys.argv[1] if len(sys.argv) > 1 else "C1"
cfg = synthetic.CONFIGS[cfgname]; sc = synthetic.make_config_scene(cfgname); cam = synthetic.make_camera(cfg["W"], cfg["H"], cfg["focal"])
dev = torch.device("cuda", 0); d = lambda t: t.to(dev)
s = GaussianRasterizationSettings(cfg["H"], cfg["W"], cam["tanfovx"], cam["tanfovy"], d(sc["bg"]), 1.0, d(cam["viewmatrix"]), d(cam["projmatrix"]), cfg["sh_degree"], d(cam["campos"]), False, True, False)
frame = bench.DeformFrame(sc, GaussianRasterizer(s), dev, True) if cfg["n_objects"] > 0 else bench.StaticFrame(sc, GaussianRasterizer(s), dev, True)
up = synthetic.make_upstream_grads(sc, 0)
ups = [d(up[k]) for k in ("color", "depth", "img_opacity", "flow", "semantic")]
def step():
    outs = frame.forward(); torch.autograd.backward(outs, ups); frame.zero_grad()
for _ in range(20): step()
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(300): step()
torch.cuda.synchronize(); print("%.3f ms/step" % ((time.perf_counter() - t0) / 300 * 1e3))
pr = cProfile.Profile(); pr.enable()
for _ in range(300): step()
torch.cuda.synchronize(); pr.disable()
st = io.StringIO(); pstats.Stats(pr, stream=st).sort_stats("tottime").print_stats(22); print(st.getvalue()[:4500])
