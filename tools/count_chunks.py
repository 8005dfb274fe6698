import sys, os, torch
sys.path.insert(0, '/root/repo'); sys.path.insert(0, '/root/repo/ad-gs_amd')
from adgs import synthetic
from diff_gaussian_rasterization import _C
cfg = synthetic.CONFIGS["C3"]; sc = synthetic.make_config_scene("C3"); cam = synthetic.make_camera(cfg["W"], cfg["H"], cfg["focal"])
d = lambda t: t.cuda()
out = _C.rasterize_gaussians(d(sc["bg"]), d(sc["means3D"]), torch.empty(0).cuda(), d(sc["opacities"]), d(sc["scales"]), d(sc["rotations"]), 1.0, torch.empty(0).cuda(),
    d(cam["viewmatrix"]), d(cam["projmatrix"]), cam["tanfovx"], cam["tanfovy"], cfg["H"], cfg["W"], d(sc["shs"]), d(sc["flow_points"]), d(sc["semantic"]), 3, d(cam["campos"]), False, True, False)
binning = out[6]
chunks = int(binning[:4].view(torch.int32)[0])
print("R_cells", out[0], "chunks", chunks, "consumed<=", chunks * 64, "per tile", chunks * 64 / 9600)
