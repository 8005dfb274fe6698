import sys, time, types, importlib.util, torch
sys.path.insert(0, '/root/repo'); sys.path.insert(0, '/root/repo/ad-gs_amd')
spec = importlib.util.spec_from_file_location("ti", "/root/repo/examples/train_iteration.py"); ti = importlib.util.module_from_spec(spec); spec.loader.exec_module(ti)
from adgs import loss, optim
from gaussian_renderer import render
cfgname = sys.argv[1] if len(sys.argv) > 1 else "C5"
cfg, model, cam, env_map, stats, targets = ti.build(cfgname, 8192, torch.device("cuda", 0))
pipe = types.SimpleNamespace(inv_depth=True, debug=False)
def T(): torch.cuda.synchronize(); return time.perf_counter()
for it in range(8):
    t0 = T()
    pkg = render(cam, model, env_map, pipe, flow_pkg=(cam.time + 0.05,) + (None,) * 5, render_objmask=True)
    t1 = T()
    total, l1, ds = loss.photometric_loss(pkg["render"], targets["image"], 0.2)
    total = total + 0.01 * pkg["img_opacity"].mean()
    t2 = T()
    total.backward()
    t3 = T()
    with torch.no_grad():
        optim.add_densification_stats(stats["accum"], stats["denom"], stats["max_r"], pkg["viewspace_points"].grad, pkg["radii"])
        t4 = T()
        model.optimizer.step()
        t5 = T()
        env_map.optimizer.step()
        t6 = T()
        for p in model.parameters(): p.grad = None
        env_map.grid_map.grad = None
    t7 = T()
    if it >= 3: print("render %.2f loss %.2f backward %.2f stats %.2f adam %.2f envadam %.2f zero %.2f ms" % tuple((b - a) * 1e3 for a, b in ((t0, t1), (t1, t2), (t2, t3), (t3, t4), (t4, t5), (t5, t6), (t6, t7))))
