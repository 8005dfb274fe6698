import sys, time, torch
sys.path.insert(0, '/root/repo'); sys.path.insert(0, '/root/repo/ad-gs_amd')
from adgs import synthetic, deform
from adgs.model import SyntheticGaussianModel, DEFAULT_ORDER_ARGS
def T(fn, n=50):
    for _ in range(5): fn()
    torch.cuda.synchronize(); a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n): fn()
    b.record(); torch.cuda.synchronize(); return a.elapsed_time(b) / n * 1e3
sc = synthetic.make_scene(1000000, 1920, 1280, 2050.0, sh_degree=3, seed=2, n_objects=8)
for tag, oa, drop_scene in (("C3", DEFAULT_ORDER_ARGS, False), ("objects only (200k)", DEFAULT_ORDER_ARGS, True),
                            ("objects only, no quaternion spline", dict(DEFAULT_ORDER_ARGS, rotation=[0] * 6), True),
                            ("objects only, no xyz deformation", dict(DEFAULT_ORDER_ARGS, xyz=[0] * 6), True)):
    m = SyntheticGaussianModel.from_scene(sc, "cuda", seed=0, order_args=oa); m.raw_sh = True
    if drop_scene:
        for n in ("_scene_xyz", "_scene_shs_dc", "_scene_shs_rest", "_scene_scaling", "_scene_rotation", "_scene_opacity", "shs_deform_param_scene"):
            setattr(m, n, getattr(m, n).detach()[:0].contiguous().requires_grad_(True))
    def f():
        return deform.get_deformed_pkg(m, 0.37, raw_sh=True, flow_time=0.42)
    def fb():
        for p in m.parameters(): p.grad = None
        pkg = f()
        torch.autograd.backward([pkg["xyz"], pkg["flow_xyz"], pkg["rotation"], pkg["opacity"], pkg["scales"]], [torch.ones_like(pkg[k]) for k in ("xyz", "flow_xyz", "rotation", "opacity", "scales")])
    with torch.no_grad():
        tf = T(f)
    tfb = T(fb)
    print("%-40s fwd %.1f us   fwd+bwd %.1f us" % (tag, tf, tfb))
