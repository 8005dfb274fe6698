import sys, time, os, torch
sys.path.insert(0, '/root/repo'); sys.path.insert(0, '/root/repo/ad-gs_amd')
import bench
from adgs import synthetic, deform
from diff_gaussian_rasterization import GaussianRasterizationSettings, GaussianRasterizer
cfg = synthetic.CONFIGS["C3"]; sc = synthetic.make_config_scene("C3"); cam = synthetic.make_camera(cfg["W"], cfg["H"], cfg["focal"])
dev = torch.device("cuda", 0); d = lambda t: t.to(dev)
s = GaussianRasterizationSettings(cfg["H"], cfg["W"], cam["tanfovx"], cam["tanfovy"], d(sc["bg"]), 1.0, d(cam["viewmatrix"]), d(cam["projmatrix"]), 3, d(cam["campos"]), False, True, False)
frame = bench.DeformFrame(sc, GaussianRasterizer(s), dev, True)
up = synthetic.make_upstream_grads(sc, 0)
ups = [d(up[k]) for k in ("color", "depth", "img_opacity", "flow", "semantic")]
orig = deform.make_func_eval
tf = []
def timed(*a, **k):
    t0 = time.perf_counter(); r = orig(*a, **k); tf.append((time.perf_counter() - t0) * 1e3); return r
deform.make_func_eval = timed
N = int(sys.argv[1]) if len(sys.argv) > 1 else 2000
for i in range(10):
    outs = frame.forward(); torch.autograd.backward(outs, ups); frame.zero_grad()
torch.cuda.synchronize(); tf.clear()
t_all = time.perf_counter(); ts = []
for i in range(N):
    t0 = time.perf_counter()
    outs = frame.forward(); t1 = time.perf_counter(); torch.autograd.backward(outs, ups); frame.zero_grad()
    ts.append(((t1 - t0) * 1e3, (time.perf_counter() - t1) * 1e3))
torch.cuda.synchronize()
wall = (time.perf_counter() - t_all) / N * 1e3
fw = sorted(x[0] for x in ts); bw = sorted(x[1] for x in ts); tfs = sorted(tf)
print("wall %.3f ms/step | host fwd median %.3f p99 %.3f max %.3f | host bwd median %.3f p99 %.3f max %.3f" % (wall, fw[N//2], fw[int(N*.99)], fw[-1], bw[N//2], bw[int(N*.99)], bw[-1]))
print("make_func_eval: calls/step %.1f median %.3f ms p99 %.3f max %.3f total/step %.3f ms" % (len(tf)/N, tfs[len(tfs)//2], tfs[int(len(tfs)*.99)], tfs[-1], sum(tf)/N))
