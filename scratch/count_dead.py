import sys, torch
sys.path.insert(0, '/root/repo'); sys.path.insert(0, '/root/repo/ad-gs_amd')
from adgs import synthetic
from diff_gaussian_rasterization import _C
cfg = synthetic.CONFIGS["C3"]; sc = synthetic.make_config_scene("C3"); cam = synthetic.make_camera(cfg["W"], cfg["H"], cfg["focal"])
d = lambda t: t.cuda()
P = cfg["P"]
args = (d(sc["bg"]), d(sc["means3D"]), torch.empty(0).cuda(), d(sc["opacities"]), d(sc["scales"]), d(sc["rotations"]), 1.0, torch.empty(0).cuda(),
    d(cam["viewmatrix"]), d(cam["projmatrix"]), cam["tanfovx"], cam["tanfovy"], cfg["H"], cfg["W"], d(sc["shs"]), d(sc["flow_points"]), d(sc["semantic"]), 3, d(cam["campos"]), False, True, False)
out = _C.rasterize_gaussians(*args)
R, color, depth, op, radii, geom, binning, img, flow, sem = out
up = synthetic.make_upstream_grads(sc, 0)
g = _C.rasterize_gaussians_backward(d(sc["bg"]), d(sc["means3D"]), radii, torch.empty(0).cuda(), d(sc["scales"]), d(sc["rotations"]), 1.0, torch.empty(0).cuda(),
    d(cam["viewmatrix"]), d(cam["projmatrix"]), cam["tanfovx"], cam["tanfovy"], d(up["color"]), d(up["depth"]), d(up["flow"]), d(up["semantic"]), d(sc["semantic"]), d(sc["flow_points"]),
    d(sc["shs"]), 3, d(cam["campos"]), geom, R, binning, img, op, d(up["img_opacity"]), True, False)
torch.cuda.synchronize()
import ctypes
from adgs import _lib
buf = (ctypes.c_float * 4)()
_lib.lib().adgs_debug_read(buf)
print("visited", buf[0], "dead", buf[1], "active strips", buf[2], "chunks", int(binning[:4].view(torch.int32)[0]))
