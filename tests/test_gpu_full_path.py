"""Oracle-chain parity of the WHOLE hot path as bench.py and gaussian_renderer.render() run it (raw parameters -> fused
deformation incl. the flow time stamp -> raw-SH rasterizer -> images; backward into every raw parameter), at the full size
of BASELINE.json's dynamic configs:

  * C3 (the benchmarked configuration): bench.DeformFrame itself, every image and every raw-parameter gradient;
  * C5 (3 M Gaussians, 16 objects): the same chain on one GPU;
  * C4 (3 cameras per iteration over a 1 M scene): accumulation through adgs.dp.FactoredSHExchange against conventional
    accumulation, and one of its cameras against the oracle chain;
  * gaussian_renderer.render() incl. the environment-map composite against env_oracle o deform_oracle o raster_oracle.

The chain is tests/chain_ref.py: deform_oracle (NumPy f32, pinned on the reference's own Python) -> raster_oracle (C++ f32)
-> float64 torch restatement of the deformation for the chain rule (reference gaussian_renderer/__init__.py:57-94,
scene/gaussian_model.py:173-231).  Tolerance 1e-4 (north_star), tests/parity.py.
"""
import os
import sys

import numpy as np
import pytest
import torch

from tests import chain_ref, parity

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _bench():
    if ROOT not in sys.path:
        sys.path.insert(0, ROOT)
    import bench
    return bench


def _check_radii(got, want):
    """The deformation runs in float32 on both sides but not with the same instruction sequence (fma contraction, v_exp_f32 /
    v_sin_f32 vs libm): a position or scale that differs in the last bit can move ceil(3 sqrt(lambda)) across an integer for a
    handful of Gaussians per million.  Everything else is bit-exact."""
    got = np.asarray(got); want = np.asarray(want)
    diff = got != want
    assert diff.mean() <= 2e-5, "radii differ on %d of %d Gaussians" % (diff.sum(), diff.size)
    if diff.any():
        assert np.abs(got[diff].astype(np.int64) - want[diff]).max() <= 1 or ((got[diff] == 0) | (want[diff] == 0)).all()
    return int(diff.sum())


def _frame_vs_chain(config, max_frac_img=2e-5):
    from adgs import synthetic
    from diff_gaussian_rasterization import GaussianRasterizationSettings, GaussianRasterizer
    bench = _bench()
    cfg = synthetic.CONFIGS[config]
    sc = synthetic.make_config_scene(config)
    cam = synthetic.make_camera(cfg["W"], cfg["H"], cfg["focal"])
    dev = torch.device("cuda", 0)
    d = lambda t: t.to(dev)
    settings = GaussianRasterizationSettings(cfg["H"], cfg["W"], cam["tanfovx"], cam["tanfovy"], d(sc["bg"]), 1.0, d(cam["viewmatrix"]),
                                             d(cam["projmatrix"]), cfg["sh_degree"], d(cam["campos"]), False, True, False)
    frame = bench.DeformFrame(sc, GaussianRasterizer(settings), dev, True)      # exactly what bench.py times
    assert frame.model.raw_sh and frame.fused_flow
    up = synthetic.make_upstream_grads(sc, cfg["seed"])
    m = frame.model
    raw = chain_ref.raw_numpy(m)
    ups = {k: up[k].numpy() for k in up}
    ref = chain_ref.run_chain(raw, m.order_args, m.use_time_mask, frame.t, frame.t + 0.05, {k: (v.numpy() if torch.is_tensor(v) else v) for k, v in cam.items()},
                              cfg["H"], cfg["W"], cfg["sh_degree"], ups, semantic=frame.sem.cpu().numpy(), strict=True)
    ex = ref["explained"]
    parity.assert_masked_coverage(ex, what=config)
    outs = frame.forward()
    keys = ("color", "depth", "img_opacity", "flow", "semantic")
    torch.autograd.backward(outs, [d(up[k]) for k in keys], retain_graph=True)
    torch.cuda.synchronize()
    params = {name: getattr(m, chain_ref.attr_of(name)) for name in chain_ref.RAW_NAMES if name != "gs_time"}
    unmasked = {name: (None if p.grad is None else p.grad.detach().clone()) for name, p in params.items()}
    for p in params.values():
        p.grad = None
    masked_up = parity.mask_upstream({k: up[k] for k in keys}, ex["pixel"])
    torch.autograd.backward(outs, [d(masked_up[k]) for k in keys])          # the strict pass: no gradient enters at the gate-flip pixels
    torch.cuda.synchronize()
    n_rad = _check_radii(frame.last_radii.cpu().numpy(), ref["radii"])
    report = ["%s: %d radii differ; gate-flip mask: %.3g of the pixels, %.3g of the Gaussians" % (config, n_rad, ex["frac_pixel"], ex["frac_gauss"])]
    for name, got, want in zip(("color", "depth", "img_opacity", "img_flow", "img_semantic"), outs,
                               (ref["color"], ref["depth"], ref["img_opacity"], ref["img_flow"], ref["img_semantic"])):
        st = parity.assert_close(name, got.detach().cpu().numpy(), np.asarray(want).reshape(tuple(got.shape)), max_frac=max_frac_img)
        report.append(parity.fmt_stats(name, st))
    checked, cond = 0, {}
    for name, p in params.items():
        want, want_s = ref["raw_grads"][name], ref["raw_grads_strict"][name]
        if p.numel() == 0:
            continue
        if want is None or not np.any(want):
            assert unmasked[name] is None or float(unmasked[name].abs().max()) == 0.0, name
            continue
        assert p.grad is not None and unmasked[name] is not None, name
        # THE gradient check: every element, no exemption (tests/parity.py).  n_rad Gaussians whose radius differs by one (the deformation's
        # last bit, _check_radii) cover other tiles on the two sides: their rows and their tile neighbours' are the only permitted outliers
        if n_rad == 0:
            try:
                st = parity.assert_close("grad[strict] " + name, p.grad.cpu().numpy(), want_s, strict=True)
            except AssertionError as exc:
                # ill-conditioned rows (tests/parity.py: assert_rows_conditioned): the float64 chain and a second float32 draw (the raw
                # parameters moved by one float32 ulp) under the same upstream mask, computed once per frame and only when needed
                if "draws" not in cond:
                    kw = dict(semantic=frame.sem.cpu().numpy(), strict_mask=ex["pixel"])
                    camn = {k: (v.numpy() if torch.is_tensor(v) else v) for k, v in cam.items()}
                    prng = np.random.RandomState(4242)
                    raw_p = {k: (v if k == "gs_time" else (v * (1.0 + prng.choice([-1.0, 1.0], size=v.shape).astype(np.float32) * np.float32(2.0 ** -23))).astype(np.float32))
                             for k, v in raw.items()}
                    cond["draws"] = (chain_ref.run_chain(raw, m.order_args, m.use_time_mask, frame.t, frame.t + 0.05, camn, cfg["H"], cfg["W"], cfg["sh_degree"], ups,
                                                         precision="f64", **kw),
                                     chain_ref.run_chain(raw_p, m.order_args, m.use_time_mask, frame.t, frame.t + 0.05, camn, cfg["H"], cfg["W"], cfg["sh_degree"], ups, **kw))
                exact, pert = cond["draws"]
                f32 = [want_s] + ([pert["raw_grads_strict"][name]] if np.array_equal(pert["radii"], ref["radii"]) else [])
                parity.assert_rows_conditioned("grad[strict] " + name, p.grad.cpu().numpy(), f32, exact["raw_grads_strict"][name], context=str(exc).splitlines()[0])
                st = parity.error_stats(p.grad.cpu().numpy(), want_s)
        else:
            st = parity.assert_close("grad[strict] " + name, p.grad.cpu().numpy(), want_s, max_frac=max(2e-5, 40.0 * n_rad / p.numel()))
        report.append(parity.fmt_stats("grad[strict] " + name, st))
        # sanity: the unmasked backward (flips included) -- small tensors: one gate-flipped Gaussian moves the components of its row
        st = parity.assert_close("grad " + name, unmasked[name].cpu().numpy(), want, max_frac=max(2e-4, 4.5 / p.numel()))
        report.append(parity.fmt_stats("grad " + name, st))
        checked += 1
    print("\n".join(report))
    assert checked >= 16, checked          # every raw tensor except the (unused) background row and gs_time
    return frame, ref


def test_c3_benchmarked_frame_vs_oracle_chain():
    """BASELINE.json configs[2], the configuration `bench.py` reports `value` on."""
    frame, ref = _frame_vs_chain("C3")
    assert ref["num_rendered"] > 40_000_000


def test_c5_frame_3m_gaussians_16_objects_vs_oracle_chain():
    """BASELINE.json configs[4] on one GPU: 3 M Gaussians, 16 dynamic objects, 1920x1280."""
    frame, ref = _frame_vs_chain("C5")
    assert frame.model.get_pts_num == 3_000_000 and frame.model.get_obj_pts_num == 600_000


def test_c4_three_camera_accumulation_factored_vs_conventional_and_oracle():
    """BASELINE.json configs[3]: 1 M Gaussians, 3 cameras per iteration.  The gradient of the iteration is the sum over the three
    cameras; on one GPU the factored exchange path (SH gradients as per-camera colour factors, expanded once) must equal conventional
    autograd accumulation, and the conventional sum must equal the sum of three oracle chains."""
    from adgs import dp, synthetic
    from adgs.model import SyntheticGaussianModel
    from tests.test_gpu_exchange import _render_cam
    cfg = synthetic.CONFIGS["C4"]
    sc = synthetic.make_config_scene("C4")
    cams = [synthetic.make_camera(cfg["W"], cfg["H"], cfg["focal"], cam_seed=c) for c in range(3)]
    times = [0.2, 0.37, 0.71]
    up = synthetic.make_upstream_grads(sc, cfg["seed"])
    ups = [up[k].cuda() for k in ("color", "depth", "img_opacity", "flow", "semantic")]
    names = [n for n in chain_ref.RAW_NAMES if n != "gs_time"]

    def run(factored):
        model = SyntheticGaussianModel.from_scene(sc, torch.device("cuda", 0), seed=0)
        model.raw_sh = True
        ex = dp.FactoredSHExchange(model, factor_xyz=True) if factored else None
        for cam, t in zip(cams, times):
            _render_cam(model, cam, t, ups, ex.sink_for if factored else None)
        if factored:
            ex.reduce(times, [c["campos"].tolist() for c in cams], flow_times=[t + 0.05 for t in times])
        torch.cuda.synchronize()
        g = {n: getattr(model, chain_ref.attr_of(n)).grad for n in names}
        return model, {n: v.detach().cpu().numpy() for n, v in g.items() if v is not None}

    model, conv = run(False)
    _, fact = run(True)
    assert set(conv) == set(fact)
    for n in conv:
        # two HIP runs: they differ by the order of the blend kernels' fp32 atomics (measured: 1e-6 of the elements outside 2e-5,
        # relative L2 up to 2.1e-5 on the rotation gradients)
        parity.assert_close("factored vs conventional " + n, fact[n], conv[n], tol=5e-5, max_frac=1e-5, rel_l2=5e-5)
    # the conventional sum against three oracle chains (one per camera)
    raw = chain_ref.raw_numpy(model)
    upn = {k: up[k].numpy() for k in up}
    sem = model.get_obj_mask.float()[:, None].cpu().numpy()
    total = {}
    for cam, t in zip(cams, times):
        ref = chain_ref.run_chain(raw, model.order_args, model.use_time_mask, t, t + 0.05, {k: (v.numpy() if torch.is_tensor(v) else v) for k, v in cam.items()},
                                  cfg["H"], cfg["W"], cfg["sh_degree"], upn, semantic=sem)
        for n, g in ref["raw_grads"].items():
            if g is not None:
                total[n] = total.get(n, 0) + g
    for n in conv:
        if n not in total:                       # a parameter the configuration does not use (obj_rotation under the quaternion spline)
            assert not np.any(conv[n]), n
            continue
        # a sum over three cameras: the per-camera errors add while the summed rows may cancel (and the time mask hides an object
        # at two of the three time stamps), so the per-row bands are three times the single-frame ones
        st = parity.assert_close("3-camera sum " + n, conv[n], total[n], max_frac=max(6e-4, 4.5 / conv[n].size), row_tol=((1e-3, 3e-2), (1e-2, 3e-3)))
        print(parity.fmt_stats("3-camera sum " + n, st))


class _Pipe:
    inv_depth, debug = True, False


@pytest.mark.parametrize("P,W,H,focal,n_objects,res,cam_seed", [(60000, 640, 400, 620.0, 4, 256, 5), (300000, 1242, 375, 721.5, 6, 1024, None)])
def test_render_entry_with_env_map_vs_oracle_chain(P, W, H, focal, n_objects, res, cam_seed):
    """The whole `render()` result -- deformation -> rasterizer -> environment-map composite `fg + (1 - O) bg`
    (gaussian_renderer/__init__.py:93-94, scene/env.py:43-76; composited in the blend epilogue on this path, 'foreground' handed out
    on demand) -- and the gradients of a loss on every returned image, into the
    raw parameters, the screen-space means (viewspace_points.grad, what add_densification_stats reads) and the environment map."""
    from adgs import synthetic
    from adgs.env import EnvironmentMap, fov2focal
    from adgs.model import SyntheticGaussianModel
    from gaussian_renderer import render
    sc = synthetic.make_scene(P, W, H, focal, sh_degree=3, seed=17, n_objects=n_objects)
    camd = synthetic.make_camera(W, H, focal, cam_seed=cam_seed)
    model = SyntheticGaussianModel.from_scene(sc, device="cuda", seed=2)
    model.raw_sh = True
    model.raw_scene = cam_seed is None             # the second case also takes the raw-scene path (activations inside the preprocess)
    cam = synthetic.camera_object(camd, time=0.61)
    env = EnvironmentMap(res, device="cuda")
    with torch.no_grad():
        env.grid_map.copy_(torch.randn(env.grid_map.shape, generator=torch.Generator().manual_seed(5)).cuda() * 1.5)
    t_flow = 0.66
    out = render(cam, model, env, _Pipe(), flow_pkg=(t_flow, None, None, None, None, None), render_objmask=True)
    up = synthetic.make_upstream_grads(sc, 9)
    d = lambda k: up[k].cuda()
    torch.autograd.backward([out["render"], out["depth"], out["img_opacity"], out["img_flow"], out["img_semantic"]],
                            [d("color"), d("depth")[0], d("img_opacity")[0], d("flow"), d("semantic")])
    torch.cuda.synchronize()
    raw = chain_ref.raw_numpy(model)
    ups = dict(render=up["color"].numpy(), depth=up["depth"].numpy(), img_opacity=up["img_opacity"].numpy(), flow=up["flow"].numpy(),
               semantic=up["semantic"].numpy())
    envd = dict(grid_map=env.grid_map.detach().cpu().numpy()[0], focal=fov2focal(cam.FoVx, W), R=camd["viewmatrix"][:3, :3].numpy())
    ref = chain_ref.run_chain(raw, model.order_args, model.use_time_mask, cam.time, t_flow, {k: (v.numpy() if torch.is_tensor(v) else v) for k, v in camd.items()},
                              H, W, 3, ups, semantic=model.get_obj_mask.float()[:, None].cpu().numpy(), env=envd)
    _check_radii(out["radii"].cpu().numpy(), ref["radii"])
    assert torch.equal(out["visibility_filter"], out["radii"] > 0)
    for key, want in (("render", ref["render"]), ("foreground", ref["color"]), ("background", ref["background"]), ("depth", ref["depth"]),
                      ("img_opacity", ref["img_opacity"]), ("img_flow", ref["img_flow"]), ("img_semantic", ref["img_semantic"])):
        got = out[key].detach().cpu().numpy()
        print(parity.fmt_stats(key, parity.assert_close(key, got, np.asarray(want).reshape(got.shape))))
    # the reference returns the deformed state as well (gaussian_renderer/__init__.py:99-115)
    for key, akey in (("xyz", "xyz"), ("rotation", "rotation"), ("opacity", "opacity")):
        parity.assert_close("pkg " + key, out[key].detach().cpu().numpy(), ref["act"][akey], tol=1e-5, max_frac=0, rel_l2=1e-6, row_tol=None)
    st = parity.assert_close("viewspace_points.grad", out["viewspace_points"].grad.cpu().numpy(), ref["act_grads"]["dL_dmeans2D"], max_frac=2e-4)
    print(parity.fmt_stats("viewspace_points.grad", st))
    for name in chain_ref.RAW_NAMES:
        if name == "gs_time":
            continue
        p = getattr(model, chain_ref.attr_of(name))
        want = ref["raw_grads"][name]
        if p.numel() == 0 or want is None or not np.any(want):
            continue
        st = parity.assert_close("grad " + name, p.grad.cpu().numpy(), want, max_frac=max(2e-4, 4.5 / p.numel()))
        print(parity.fmt_stats("grad " + name, st))
    st = parity.assert_close("grad env grid_map", env.grid_map.grad.cpu().numpy()[0], ref["env_grad"], max_frac=2e-5, row_tol=None)
    print(parity.fmt_stats("grad env grid_map", st))
