"""The forward-only render (include/adgs_rasterizer.h: adgs_raster_render / adgs_raster_render_rawsh): what the reference runs under
torch.no_grad() for evaluation and reports as its render FPS (/root/reference/render.py:52-55,86,156).  Same images and radii as the
training forward BIT FOR BIT (it is the same blend loop with the publication of the replay lists compiled out), through the drop-in API:
GaussianRasterizer under no_grad / with inputs that need no gradient, gaussian_renderer.render() with the environment map, frames with
several semantic channels (they publish their lists all the same), HIP-graph replay, and the oracle for good measure."""
import numpy as np
import pytest
import torch

from adgs import _lib, synthetic
from tests.parity import assert_close
from tests.test_gpu_raster import dev, run_oracle

pytestmark = pytest.mark.gpu


def _settings(sc, **kw):
    from diff_gaussian_rasterization import GaussianRasterizationSettings
    return GaussianRasterizationSettings(sc["H"], sc["W"], sc["tanfovx"], sc["tanfovy"], dev(sc["bg"]), kw.get("scale_modifier", 1.0), dev(sc["viewmatrix"]),
                                         dev(sc["projmatrix"]), sc["sh_degree"], dev(sc["campos"]), False, kw.get("inv_depth", True), False)


def _call(rast, sc, grad, semantic=None, colors=None):
    leaf = lambda t: t.cuda().clone().requires_grad_(grad)
    sem = sc["semantic"] if semantic is None else semantic
    return rast(means3D=leaf(sc["means3D"]), means2D=torch.zeros(sc["P"], 3, device="cuda", requires_grad=grad), opacities=leaf(sc["opacities"]),
                shs=None if colors is not None else leaf(sc["shs"]), colors_precomp=None if colors is None else leaf(colors), scales=leaf(sc["scales"]),
                rotations=leaf(sc["rotations"]), flow_points=dev(sc["flow_points"]), semantic=dev(sem))


@pytest.mark.parametrize("P,W,H,focal", [(20000, 320, 208, 300.0), (120000, 1024, 1024, 900.0)])      # half-tile waves / one wave per 16x16 tile
def test_no_grad_and_gradient_free_inputs_take_the_render_path_and_equal_the_training_forward(P, W, H, focal):
    from diff_gaussian_rasterization import GaussianRasterizer
    sc = synthetic.make_scene(P, W, H, focal, seed=91, n_objects=3, scale_mult=0.012 if W > 512 else 1.0)
    rast = GaussianRasterizer(_settings(sc))
    train = _call(rast, sc, True)
    assert train[0].grad_fn is not None                       # the training forward: an autograd node, state kept
    with torch.no_grad():
        a = _call(rast, sc, True)                             # gradients switched off
    b = _call(rast, sc, False)                                # gradients on, nothing asks for one
    for outs in (a, b):
        assert all(o.grad_fn is None and not o.requires_grad for o in outs)
        for x, y in zip(outs, train):
            assert torch.equal(x, y.detach())
    o = run_oracle(sc)
    np.testing.assert_array_equal(a[1].cpu().numpy(), o["radii"])
    ex = o["explained"]
    for k, i in (("color", 0), ("depth", 2), ("img_opacity", 3), ("img_flow", 4), ("img_semantic", 5)):
        assert_close(k, a[i].cpu().numpy(), o[k], explained=ex["pixel"])


def test_colors_precomp_scale_modifier_and_several_semantic_channels():
    from diff_gaussian_rasterization import GaussianRasterizer
    sc = synthetic.make_scene(9000, 256, 160, 200.0, seed=92)
    cols = torch.rand(sc["P"], 3, generator=torch.Generator().manual_seed(3))
    sem4 = torch.rand(sc["P"], 4, generator=torch.Generator().manual_seed(4))
    rast = GaussianRasterizer(_settings(sc, scale_modifier=0.7, inv_depth=False))
    for kw in (dict(colors=cols), dict(semantic=sem4)):       # 4 channels: channels 1.. are a replay of the published lists -> the frame publishes
        train = _call(rast, sc, True, **kw)
        with torch.no_grad():
            ev = _call(rast, sc, True, **kw)
        for x, y in zip(ev, train):
            assert torch.equal(x, y.detach())


def test_render_entry_under_no_grad_equals_the_training_render_with_env_map():
    """gaussian_renderer.render(): deformation (its own no-grad path) -> raw-SH rasterizer entry -> environment map composited in the blend epilogue."""
    from adgs.env import EnvironmentMap
    from adgs.model import SyntheticGaussianModel
    from gaussian_renderer import render
    from tests.test_gpu_full_path import _Pipe
    W, H, focal = 640, 400, 620.0
    sc = synthetic.make_scene(60000, W, H, focal, sh_degree=3, seed=17, n_objects=4)
    cam = synthetic.camera_object(synthetic.make_camera(W, H, focal, cam_seed=5), time=0.61)
    env = EnvironmentMap(256, device="cuda")
    with torch.no_grad():
        env.grid_map.copy_(torch.randn(env.grid_map.shape, generator=torch.Generator().manual_seed(5)).cuda())
    for raw_scene in (False, True):
        model = SyntheticGaussianModel.from_scene(sc, device="cuda", seed=2)
        model.raw_sh, model.raw_scene = True, raw_scene
        kw = dict(flow_pkg=(0.66, None, None, None, None, None), render_objmask=True)
        train = render(cam, model, env, _Pipe(), **kw)
        assert train["render"].grad_fn is not None
        with torch.no_grad():
            ev = render(cam, model, env, _Pipe(), **kw)
        for key in ("render", "depth", "img_opacity", "img_flow", "img_semantic", "radii", "visibility_filter"):
            assert ev[key].grad_fn is None and torch.equal(ev[key], train[key].detach()), key


def test_render_path_replayed_as_a_hip_graph():
    """No host decision inside the forward-only frame either: captured once, replayed as one hipGraphLaunch; an evaluation loop over a
    camera path captures one graph per camera."""
    from adgs import graph
    from diff_gaussian_rasterization import GaussianRasterizer
    sc = synthetic.make_scene(30000, 480, 320, 400.0, seed=93, n_objects=2)
    rast = GaussianRasterizer(_settings(sc))
    t = {k: sc[k].cuda() for k in ("means3D", "opacities", "shs", "scales", "rotations", "flow_points", "semantic")}
    m2 = torch.zeros(sc["P"], 3, device="cuda")

    def fn():
        with torch.no_grad():
            return rast(means3D=t["means3D"], means2D=m2, opacities=t["opacities"], shs=t["shs"], scales=t["scales"], rotations=t["rotations"],
                        flow_points=t["flow_points"], semantic=t["semantic"])
    want = [o.clone() for o in fn()]
    step = graph.GraphedStep(fn)
    for _ in range(3):
        got = step()
    torch.cuda.synchronize()
    assert step.validate(repair=False)
    for x, y in zip(got, want):
        assert torch.equal(x, y)


def test_the_c_abi_exports_the_render_entry_points():
    lib = _lib.lib()
    assert hasattr(lib, "adgs_raster_render") and hasattr(lib, "adgs_raster_render_rawsh")

