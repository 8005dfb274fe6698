"""Timing of the fused deformation forward / backward for variants of the C3 model (which part costs what)."""
import sys, time, torch
sys.path.insert(0, '/root/repo'); sys.path.insert(0, '/root/repo/ad-gs_amd')
from adgs import synthetic, deform
from adgs.model import SyntheticGaussianModel, DEFAULT_ORDER_ARGS
sc = synthetic.make_config_scene("C3")
def run(tag, oa):
    m = SyntheticGaussianModel.from_scene(sc, "cuda", seed=0, order_args=oa)
    m.raw_sh = True
    def f():
        return deform.get_deformed_pkg(m, 0.37, raw_sh=True, flow_time=0.42)
    def fb():
        for p in m.parameters(): p.grad = None
        pkg = f()
        (pkg["xyz"].sum() + pkg["flow_xyz"].sum() + pkg["rotation"].sum() + pkg["opacity"].sum() + pkg["scales"].sum()).backward()
    for fn, name in ((f, "fwd"), (fb, "fwd+bwd")):
        with torch.set_grad_enabled(name != "fwd"):
            for _ in range(5): fn()
            torch.cuda.synchronize(); a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record()
            for _ in range(30): fn()
            b.record(); torch.cuda.synchronize()
            print("%-28s %-8s %.1f us" % (tag, name, a.elapsed_time(b) / 30 * 1e3))
run("default (xyz bs+fft, quat)", dict(DEFAULT_ORDER_ARGS))
run("no quaternion spline", dict(DEFAULT_ORDER_ARGS, rotation=[0] * 6))
run("no xyz deformation", dict(DEFAULT_ORDER_ARGS, xyz=[0] * 6))
run("rotation via fft (no quat)", dict(DEFAULT_ORDER_ARGS, rotation=[0, 0, 0, 3, 0, 0]))
