import sys, os, time
sys.path.insert(0, '/root/repo'); sys.path.insert(0, '/root/repo/ad-gs_amd')
import torch
from adgs import synthetic, _lib
from diff_gaussian_rasterization import GaussianRasterizationSettings, GaussianRasterizer
name = sys.argv[1] if len(sys.argv) > 1 else "C3"
cfg = synthetic.CONFIGS[name]
sc = synthetic.make_config_scene(name)
d = lambda t: t.cuda()
s = GaussianRasterizationSettings(sc["H"], sc["W"], sc["tanfovx"], sc["tanfovy"], d(sc["bg"]), 1.0, d(sc["viewmatrix"]), d(sc["projmatrix"]), cfg["sh_degree"], d(sc["campos"]), False, True, True)
r = GaussianRasterizer(s)
L = {k: d(sc[k]).clone().requires_grad_(True) for k in ("means3D", "opacities", "shs", "scales", "rotations")}
m2 = torch.zeros(sc["P"], 3, device="cuda", requires_grad=True)
for it in range(3):
    t0 = time.time()
    out = r(means3D=L["means3D"], means2D=m2, opacities=L["opacities"], shs=L["shs"], scales=L["scales"], rotations=L["rotations"], flow_points=d(sc["flow_points"]), semantic=d(sc["semantic"]))
    torch.cuda.synchronize(); t1 = time.time()
    print("fwd ok", it, t1 - t0, _lib.frame_stats(), float(out[0].mean()), float(out[3].mean()), flush=True)
    g = synthetic.make_upstream_grads(sc, 0)
    torch.autograd.backward([out[0], out[2], out[3], out[4], out[5]], [d(g["color"]), d(g["depth"]), d(g["img_opacity"]), d(g["flow"]), d(g["semantic"])])
    torch.cuda.synchronize(); print("bwd ok", it, time.time() - t1, float(L["means3D"].grad.abs().max()), flush=True)
