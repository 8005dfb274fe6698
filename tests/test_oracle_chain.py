"""CPU pins of the oracle CHAIN (tests/chain_ref.py) that the full-path GPU tests compare against:

  * composition: the chain's raw-parameter gradients (deform_oracle -> raster_oracle's hand-derived backward -> float64 torch
    restatement of the deformation) equal ONE end-to-end float64 autograd pass through the torch restatements of both stages
    (tests/torch_deform_ref.py -> tests/torch_ref.py), wherever the reference's backward is the true derivative
    (grad_img_opacity = 0: the opacity-T quirk has its own pins in tests/test_oracle_raster.py);
  * the environment-map composite: render = fg + (1 - O) bg, the extra -sum_c(g_c bg_c) term on dL/dO, and the map gradient
    against central differences of env_oracle.background.
"""
import numpy as np
import torch

from adgs import synthetic
from adgs.env import fov2focal
from adgs.model import SyntheticGaussianModel
from oracle import env_oracle
from tests import chain_ref, torch_ref
from tests import torch_deform_ref as tr


def _setup(P=400, W=64, H=48, focal=60.0, seed=4, cam_seed=3):
    sc = synthetic.make_scene(P, W, H, focal, sh_degree=3, seed=seed, n_objects=2, scale_mult=0.02)
    m = SyntheticGaussianModel.from_scene(sc, "cpu", seed=1)
    cam = synthetic.make_camera(W, H, focal, cam_seed=cam_seed)
    up = {k: v.numpy() for k, v in synthetic.make_upstream_grads(sc, 2).items()}
    return sc, m, cam, up


def test_chain_equals_end_to_end_float64_autograd():
    sc, m, cam, up = _setup()
    H, W, t, tf = sc["H"], sc["W"], 0.37, 0.42
    up["img_opacity"] = np.zeros_like(up["img_opacity"])
    raw = chain_ref.raw_numpy(m)
    camn = {k: (v.numpy() if torch.is_tensor(v) else v) for k, v in cam.items()}
    ref = chain_ref.run_chain(raw, m.order_args, True, t, tf, camn, H, W, 3, up, semantic=sc["semantic"].numpy(), precision="f64")
    m64 = {k: torch.tensor(np.asarray(v, np.float64), requires_grad=(k != "gs_time")) for k, v in raw.items()}
    pkg = tr.get_deformed_pkg(m64, t, m.order_args, True)
    flow = chain_ref.deformed_xyz64(m64, tf, m.order_args)
    color, radii, depth, op, fl, sem = torch_ref.render_dense(
        pkg["xyz"], None, pkg["opacity"], pkg["shs"], None, pkg["scales"], pkg["rotation"], None, flow, sc["semantic"].double(),
        torch.zeros(3), cam["viewmatrix"], cam["projmatrix"], cam["campos"], cam["tanfovx"], cam["tanfovy"], H, W, 3, 1.0, True)
    np.testing.assert_array_equal(radii.numpy(), ref["radii"])
    assert int((radii > 0).sum()) > 150
    T = lambda a: torch.tensor(a, dtype=torch.float64)
    loss = (color * T(up["color"])).sum() + (depth * T(up["depth"])).sum() + (fl * T(up["flow"])).sum() + (sem * T(up["semantic"])).sum()
    loss.backward()
    checked = 0
    for name, want in ref["raw_grads"].items():
        got = m64[name].grad
        if want is None or got is None:
            assert (want is None or not np.any(want)) and (got is None or not bool(got.abs().max() > 0)), name
            continue
        scale = max(np.abs(got.numpy()).max(), 1e-30)
        # the oracle takes float32 inputs (the activated tensors are rounded once) and its conic backward carries the reference's
        # 1/(det^2 + 1e-7): both are ~1e-5 relative effects
        np.testing.assert_allclose(want, got.numpy(), rtol=2e-4, atol=2e-5 * scale, err_msg=name)
        checked += 1
    assert checked == 16


def test_env_composite_terms_and_map_gradient():
    sc, m, cam, up = _setup(P=300, seed=6, cam_seed=8)
    H, W, t, tf = sc["H"], sc["W"], 0.61, 0.66
    raw = chain_ref.raw_numpy(m)
    camn = {k: (v.numpy() if torch.is_tensor(v) else v) for k, v in cam.items()}
    rng = np.random.RandomState(0)
    env = dict(grid_map=rng.randn(3, 32, 32) * 1.5, focal=fov2focal(cam["fovx"], W), R=cam["viewmatrix"][:3, :3].numpy())
    ups = dict(up, render=up["color"])
    sem = sc["semantic"].numpy()
    a = chain_ref.run_chain(raw, m.order_args, True, t, tf, camn, H, W, 3, ups, semantic=sem, precision="f64", env=env)
    np.testing.assert_allclose(a["render"], a["color"] + (1 - a["img_opacity"]) * a["background"], rtol=0, atol=1e-15)
    assert a["background"].min() > 0 and a["background"].max() < 1 and float((a["img_opacity"] < 0.5).mean()) > 0.05
    # without the map, the same raw gradients must come out of the chain fed with the composite's two upstream terms
    up_b = dict(up, img_opacity=(up["img_opacity"].astype(np.float64) - (up["color"].astype(np.float64) * a["background"]).sum(0, keepdims=True)).astype(np.float32))
    b = chain_ref.run_chain(raw, m.order_args, True, t, tf, camn, H, W, 3, up_b, semantic=sem, precision="f64")
    for k, v in a["raw_grads"].items():
        if v is not None:
            np.testing.assert_allclose(v, b["raw_grads"][k], rtol=1e-8, atol=1e-9 * np.abs(v).max(), err_msg=k)      # same arithmetic; the oracle's OpenMP sums are unordered
    # map gradient: central differences of sum(g (1 - O) bg) along random directions of the map
    w = (1.0 - np.asarray(a["img_opacity"], np.float64)) * up["color"].astype(np.float64)
    f = lambda gm: float((w * env_oracle.background(gm, H, W, env["focal"], env["R"])).sum())
    for s in range(3):
        dirn = np.random.RandomState(10 + s).randn(*env["grid_map"].shape)
        h = 1e-5
        fd = (f(env["grid_map"] + h * dirn) - f(env["grid_map"] - h * dirn)) / (2 * h)
        an = float((a["env_grad"] * dirn).sum())
        assert abs(fd - an) <= 1e-6 * max(abs(an), 1e-12) + 1e-14, (fd, an)
