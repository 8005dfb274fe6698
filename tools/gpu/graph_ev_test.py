import os, sys
ROOT = "/root/repo"
for p in (ROOT, os.path.join(ROOT, "ad-gs_amd")):
    sys.path.insert(0, p)
import torch, bench
from adgs import _lib, synthetic, graph
device = torch.device("cuda", 0)
cfg = synthetic.CONFIGS["C2"]; sc = bench.build_scene("C2")
pool = bench.camera_pool(cfg, 2); frames = bench.frame_pool(sc, cfg, pool, device, True)
up = synthetic.make_upstream_grads(sc, 0)
ups = [up[k].to(device) for k in ("color", "depth", "img_opacity", "flow", "semantic")]
prof = _lib.StageProfiler(); prof.reserve(256)
f = frames[0]
def fn():
    torch.autograd.backward(f.forward(), ups); g = [p.grad for p in f.parameters()]; f.zero_grad(); return g
for _ in range(3): fn()
torch.cuda.synchronize()
prof.enable(True, stages=["render_bwd"])       # events recorded DURING capture become event-record nodes
step = graph.GraphedStep(fn, warmup=1)
prof.enable(False)
torch.cuda.synchronize()
try:
    print("collect after capture:", prof.collect())
except Exception as e:
    print("collect failed", e)
for _ in range(5): step()
torch.cuda.synchronize()
print("after replays: status", _lib.frame_status())
