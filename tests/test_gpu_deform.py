"""GPU parity of the fused HIP deformation kernels: forward vs the NumPy oracle / golden vectors of
the reference's own Python, backward vs golden gradients (reference autograd) and vs the float64
torch restatement.  Tolerance 1e-4 (north_star); observed errors are ~1e-6."""
import os

import numpy as np
import pytest
import torch

from oracle import deform_oracle as do
from tests import torch_deform_ref as tr

pytestmark = pytest.mark.gpu
GOLD = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "deform_golden.npz"))
FUNC_CASES = sorted({k[len("func_"):-len("_order")] for k in GOLD.files if k.startswith("func_") and k.endswith("_order")})
PKG_CASES = sorted({k[len("pkg_"):-len("_ts")] for k in GOLD.files if k.startswith("pkg_") and k.endswith("_ts")})
TOL = 1e-4


def close(name, a, b, tol=TOL):
    a = np.asarray(a, np.float64); b = np.asarray(b, np.float64)
    scale = max(np.abs(b).max(), 1e-30)
    np.testing.assert_allclose(a, b, rtol=tol, atol=tol * scale, err_msg=name)


@pytest.mark.parametrize("name", FUNC_CASES)
def test_get_func_result_forward_backward_vs_reference_golden(name):
    from adgs.deform import get_func_result
    oa = GOLD["func_%s_order" % name].tolist()
    for vi, v in enumerate(GOLD["vs"].tolist()):
        p = torch.tensor(GOLD["func_%s_param" % name], device="cuda", requires_grad=True)
        r = get_func_result(v, p, oa)
        w = torch.linspace(0.5, 1.5, r.numel(), device="cuda").reshape(r.shape)
        (r * w).sum().backward()
        close("%s out v=%g" % (name, v), r.detach().cpu().numpy(), GOLD["func_%s_out_%d" % (name, vi)])
        close("%s grad v=%g" % (name, v), p.grad.cpu().numpy(), GOLD["func_%s_grad_%d" % (name, vi)])


def test_get_func_result_zero_orders_and_cpu_tensor():
    from adgs.deform import get_func_result
    assert get_func_result(0.3, torch.zeros(2, 3, 0, device="cuda"), [0] * 6) == 0.0
    with pytest.raises(RuntimeError):
        get_func_result(0.3, torch.zeros(2, 3, 12), [0, 0, 0, 6, 0, 0])


class _Model:
    pass


def _model_from_gold(tag, device="cuda", requires_grad=True):
    pre = "pkg_%s_" % tag
    m = _Model()
    raw = {}
    for k in GOLD.files:
        if k.startswith(pre + "in_"):
            name = k[len(pre) + 3:]
            raw[name] = GOLD[k]
            t = torch.tensor(GOLD[k], device=device)
            if requires_grad and name != "gs_time":
                t.requires_grad_(True)
            attr = name if name.endswith("deform_param") or name.startswith("shs_deform") or name.startswith("gs_") else "_" + name
            setattr(m, attr, t)
    m.order_args = {k: GOLD[pre + "order_" + k].tolist() for k in ("xyz", "rotation", "shs", "background")}
    m.use_time_mask = bool(GOLD[pre + "use_time_mask"])
    return m, raw, pre


@pytest.mark.parametrize("tag", PKG_CASES)
def test_fused_deformed_pkg_vs_reference_golden_and_autograd(tag):
    from adgs.deform import get_deformed_pkg
    for ti, t in enumerate((0.0, 0.4, 1.0)):
        m, raw, pre = _model_from_gold(tag)
        pkg = get_deformed_pkg(m, t)
        for key in ("xyz", "rotation", "shs", "opacity", "scales"):
            close("%s %s t=%g" % (tag, key, t), pkg[key].detach().cpu().numpy(), GOLD[pre + "t%d_%s" % (ti, key)])
        # backward vs float64 autograd of the torch restatement
        g = torch.Generator().manual_seed(ti)
        ws = {k: torch.randn(pkg[k].shape, generator=g) for k in pkg}
        sum((pkg[k] * ws[k].cuda()).sum() for k in pkg).backward()
        m64 = {k: torch.tensor(v, dtype=torch.float64, requires_grad=(k != "gs_time")) for k, v in raw.items()}
        ref = tr.get_deformed_pkg(m64, t, m.order_args, m.use_time_mask)
        sum((ref[k] * ws[k].double()).sum() for k in ref).backward()
        for name, tt in m64.items():
            if name == "gs_time":
                continue
            attr = name if name.endswith("deform_param") or name.startswith("shs_deform") or name.startswith("gs_") else "_" + name
            got = getattr(m, attr).grad
            want = tt.grad
            if want is None or float(want.abs().max()) == 0.0:
                assert got is None or float(got.abs().max()) == 0.0, name
            else:
                close("%s grad %s t=%g" % (tag, name, t), got.cpu().numpy(), want.numpy(), tol=2e-4)


@pytest.mark.parametrize("Ns,No", [(30000, 7000), (777, 0), (0, 333), (129, 127)])
def test_large_random_model_matches_numpy_oracle(Ns, No):
    """Also the degenerate splits: no object Gaussians, no scene Gaussians, a block straddling nothing (ranges are separate)."""
    from adgs.deform import get_deformed_pkg, get_deformed_xyz, get_param_num
    g = torch.Generator().manual_seed(0)
    oa = dict(xyz=[16, 5, 0, 6, 0, 0], rotation=[0, 0, 0, 0, 16, 5], shs=[0, 0, 0, 6, 0, 0], background=[16, 2, 0, 6, 0, 0])
    r = lambda *s: torch.randn(*s, generator=g)
    raw = dict(scene_xyz=r(Ns, 3) * 10, obj_xyz=r(No, 3) * 10, scene_shs_dc=r(Ns, 1, 3), obj_shs_dc=r(No, 1, 3),
               scene_shs_rest=r(Ns, 15, 3) * 0.1, obj_shs_rest=r(No, 15, 3) * 0.1, scene_scaling=r(Ns, 3) * 0.3 - 2,
               obj_scaling=r(No, 3) * 0.3 - 2, scene_rotation=r(Ns, 4), obj_rotation=r(No, 4), scene_opacity=r(Ns, 1), obj_opacity=r(No, 1),
               xyz_deform_param=r(No, 3, get_param_num(oa["xyz"])) * 0.1, rotation_deform_param=r(No, 4, get_param_num(oa["rotation"])) * 0.3,
               shs_deform_param_scene=r(Ns, 3, 12) * 0.1, shs_deform_param_obj=r(No, 3, 12) * 0.1,
               background_deform_param=r(1, 3, get_param_num(oa["background"])) * 0.1, gs_time=torch.rand(No, 1, generator=g),
               gs_time_sigma=r(No, 2) * 0.3 - 1.5)
    m = _Model()
    for k, v in raw.items():
        attr = k if k.endswith("deform_param") or k.startswith("shs_deform") or k.startswith("gs_") else "_" + k
        setattr(m, attr, v.cuda())
    m.order_args, m.use_time_mask = oa, True
    npm = {k: v.numpy() for k, v in raw.items()}
    npm["order_args"], npm["use_time_mask"] = oa, True
    for t in (0.0, 0.123, 0.77, 1.0):
        pkg = get_deformed_pkg(m, t)
        ref = do.get_deformed_pkg(npm, t)
        for key in ("xyz", "rotation", "shs", "opacity", "scales"):
            close("%s t=%g" % (key, t), pkg[key].cpu().numpy(), ref[key])
        np.testing.assert_array_equal(get_deformed_xyz(m, t).cpu().numpy(), pkg["xyz"].cpu().numpy())
    # and the backward runs on every split (gradients of the existing parameters are finite)
    leaves = [getattr(m, a) for a in dir(m) if torch.is_tensor(getattr(m, a, None)) and getattr(m, a).numel() > 0 and a not in ("gs_time",)]
    for p in leaves:
        p.requires_grad_(True)
    pkg = get_deformed_pkg(m, 0.4, flow_time=0.45)
    sum((v * v).sum() for v in pkg.values() if torch.is_tensor(v) and v.numel()).backward()
    assert all(p.grad is None or torch.isfinite(p.grad).all() for p in leaves)


def test_render_entry_runs_on_fused_pkg():
    """gaussian_renderer.render() (reference signature) over a model whose getters are the fused HIP path."""
    import math
    from adgs import synthetic
    from adgs.model import SyntheticGaussianModel
    from gaussian_renderer import render
    sc = synthetic.make_scene(4000, 160, 96, 120.0, sh_degree=3, seed=3, n_objects=2)
    model = SyntheticGaussianModel.from_scene(sc, device="cuda", seed=0)
    cam = synthetic.camera_object(sc, time=0.37)

    class Pipe:
        inv_depth, debug = True, False
    flow_pkg = (0.42, None, None, None, None, None)
    res = render(cam, model, None, Pipe(), flow_pkg=flow_pkg, render_objmask=True)
    assert res["render"].shape == (3, 96, 160) and res["img_flow"].shape == (3, 96, 160) and res["img_semantic"].shape == (1, 96, 160)
    assert res["visibility_filter"].dtype == torch.bool and int(res["visibility_filter"].sum()) > 100
    loss = res["render"].mean() + res["depth"].mean() + res["img_opacity"].mean() + res["img_flow"].abs().mean() + res["img_semantic"].mean()
    loss.backward()
    assert res["viewspace_points"].grad is not None and float(res["viewspace_points"].grad[:, :2].abs().max()) > 0
    for p in model.parameters():
        assert p.grad is not None and torch.isfinite(p.grad).all()
    assert float(model.xyz_deform_param.grad.abs().max()) > 0 and float(model.rotation_deform_param.grad.abs().max()) > 0


@pytest.mark.parametrize("seed", [0, 1, 2])
def test_raw_sh_path_matches_materialised_sh_path(seed):
    """GaussianRasterizer.forward_rawsh (SH read from / gradients written to the raw scene||object tensors)
    == get_deformed_pkg + GaussianRasterizer.forward, forward and backward."""
    from adgs import synthetic
    sc = synthetic.make_scene(6000, 208, 130, 150.0, sh_degree=3, seed=seed, n_objects=2 if seed < 2 else 0)      # seed 2: a model without objects
    _raw_vs_materialised(sc, seed, max(3 - seed, 2), None, 0.37, 0.42)


def test_raw_sh_path_with_three_semantic_channels():
    """The raw-SH entry points with D_S = 3 (channels 1, 2 are replayed on top of the main blend): == the materialised path, incl. the
    gradient of the semantic values."""
    from adgs import synthetic
    sc = synthetic.make_scene(5000, 208, 130, 150.0, sh_degree=3, seed=4, n_objects=2)
    _raw_vs_materialised(sc, 4, 3, None, 0.37, 0.42, D_S=3)


@pytest.mark.parametrize("seed", range(int(os.environ.get("ADGS_TEST_SEED_BASE", "0")), int(os.environ.get("ADGS_TEST_SEED_BASE", "0")) + int(os.environ.get("ADGS_TEST_RAWSH_SEEDS", "8"))))
def test_raw_sh_path_fuzz(seed):
    """The same over random scene sizes, image shapes, active SH degrees, object counts, basis mixes of every deformation function
    (incl. no SH deformation at all) and time stamps."""
    from adgs import synthetic
    rng = np.random.RandomState(11000 + seed)
    sc = synthetic.make_scene(int(rng.choice([40, 900, 5000])), int(rng.randint(40, 300)), int(rng.randint(30, 200)), float(rng.uniform(80, 250)),
                              sh_degree=3, seed=700 + seed, n_objects=int(rng.randint(0, 4)))
    oa = dict(xyz=_random_order(rng, False), rotation=_random_order(rng, True), shs=_random_order(rng, False), background=_random_order(rng, False))
    if rng.randint(4) == 0:
        oa["shs"] = [0] * 6
    if rng.randint(2):
        oa["background"] = [0] * 6
    _raw_vs_materialised(sc, seed, int(rng.randint(0, 4)), oa, float(rng.rand()), float(rng.rand()))


def _raw_vs_materialised(sc, seed, degree, order_args, t, t_flow, D_S=1):
    from adgs import synthetic, deform
    from adgs.model import SyntheticGaussianModel
    from diff_gaussian_rasterization import GaussianRasterizationSettings, GaussianRasterizer
    g = synthetic.make_upstream_grads(sc, seed, D_S=D_S)
    d = lambda x: x.cuda()
    s = GaussianRasterizationSettings(sc["H"], sc["W"], sc["tanfovx"], sc["tanfovy"], d(sc["bg"]), 1.0, d(sc["viewmatrix"]), d(sc["projmatrix"]),
                                      degree, d(sc["campos"]), False, True, False)
    rast = GaussianRasterizer(s)
    res = []
    for raw in (False, True):
        model = SyntheticGaussianModel.from_scene(sc, device="cuda", seed=3, order_args=order_args)
        pkg = deform.get_deformed_pkg(model, t, raw_sh=raw)
        flow = model.get_deformed_xyz(t_flow)
        m2 = torch.zeros_like(pkg["xyz"], requires_grad=True)
        sem = model.get_obj_mask.float()[:, None].contiguous()
        if D_S > 1:
            sem = torch.cat([sem, torch.rand(sem.shape[0], D_S - 1, generator=torch.Generator().manual_seed(seed)).cuda()], 1).requires_grad_(True)
        if raw:
            assert not torch.is_tensor(pkg["shs"])
            out = rast.forward_rawsh(pkg["xyz"], m2, pkg["opacity"], pkg["shs"], pkg["scales"], pkg["rotation"], flow_points=flow, semantic=sem)
        else:
            out = rast(means3D=pkg["xyz"], means2D=m2, opacities=pkg["opacity"], shs=pkg["shs"], scales=pkg["scales"], rotations=pkg["rotation"],
                       flow_points=flow, semantic=sem)
        color, radii, depth, op, fl, se = out
        torch.autograd.backward([color, depth, op, fl, se], [d(g["color"]), d(g["depth"]), d(g["img_opacity"]), d(g["flow"]), d(g["semantic"])])
        res.append((out, model, m2, sem))
    if D_S > 1:
        close("semantic grad", res[1][3].grad.cpu().numpy(), res[0][3].grad.cpu().numpy(), tol=1e-4)
        assert float(res[1][3].grad[:, 1:].abs().max()) > 0
    (o0, m0, a0, _), (o1, m1, a1, _) = res
    assert torch.equal(o0[1], o1[1])
    for x, y, n in zip(o0, o1, ("color", "radii", "depth", "opacity", "flow", "sem")):
        if n != "radii":
            close(n, y.detach().cpu().numpy(), x.detach().cpu().numpy(), tol=2e-5)
    close("means2D", a1.grad.cpu().numpy(), a0.grad.cpu().numpy(), tol=1e-4)
    for p0, p1, name in zip(m0.parameters(), m1.parameters(), [n for n in __import__("adgs.model", fromlist=["_RAW"])._RAW]):
        assert (p0.grad is None) == (p1.grad is None), name
        if p0.grad is not None:
            close(name, p1.grad.cpu().numpy(), p0.grad.cpu().numpy(), tol=1e-4)


@pytest.mark.parametrize("background", [False, True])
def test_fused_flow_xyz_matches_separate_evaluations(background):
    """get_deformed_pkg(t, flow_time=t2)['flow_xyz'] == get_deformed_xyz(t2) (reference gaussian_renderer/__init__.py:57),
    and the parameter gradients of the fused pass equal the sum autograd forms from the two separate passes."""
    from adgs import synthetic, deform
    from adgs.model import SyntheticGaussianModel, DEFAULT_ORDER_ARGS
    sc = synthetic.make_scene(5003, 208, 130, 150.0, sh_degree=3, seed=5, n_objects=3)
    oa = dict(DEFAULT_ORDER_ARGS)
    if background:
        oa["background"] = [5, 3, 2, 0, 0, 0]
    gen = torch.Generator().manual_seed(11)
    w = {k: torch.randn(sc["P"], c, generator=gen).cuda() for k, c in (("xyz", 3), ("flow", 3), ("rot", 4), ("op", 1), ("sc", 3))}
    res = []
    for fused in (False, True):
        model = SyntheticGaussianModel.from_scene(sc, device="cuda", seed=3, order_args=oa)
        if background:
            with torch.no_grad():
                model.background_deform_param.copy_(torch.randn(model.background_deform_param.shape, generator=torch.Generator().manual_seed(2)) * 0.1)
        if fused:
            pkg = deform.get_deformed_pkg(model, 0.37, want=("xyz", "rotation", "opacity", "scales"), flow_time=0.42)
            flow = pkg["flow_xyz"]
        else:
            pkg = deform.get_deformed_pkg(model, 0.37, want=("xyz", "rotation", "opacity", "scales"))
            flow = deform.get_deformed_xyz(model, 0.42)
        loss = (pkg["xyz"] * w["xyz"]).sum() + (flow * w["flow"]).sum() + (pkg["rotation"] * w["rot"]).sum() + \
               (pkg["opacity"] * w["op"]).sum() + (pkg["scales"] * w["sc"]).sum()
        loss.backward()
        res.append((pkg, flow, model))
    (p0, f0, m0), (p1, f1, m1) = res
    assert torch.equal(f0, f1)
    for k in ("xyz", "rotation", "opacity", "scales"):
        assert torch.equal(p0[k], p1[k]), k
    names = __import__("adgs.model", fromlist=["_RAW"])._RAW
    for a, b, name in zip(m0.parameters(), m1.parameters(), names):
        assert (a.grad is None) == (b.grad is None), name
        if a.grad is not None:
            close(name, b.grad.cpu().numpy(), a.grad.cpu().numpy(), tol=1e-5)
    assert float(m1.xyz_deform_param.grad.abs().max()) > 0


def test_flow_only_gradient_reaches_xyz_parameters():
    """Only the flow points carry an upstream gradient: xyz / xyz_deform_param still get theirs."""
    from adgs import synthetic, deform
    from adgs.model import SyntheticGaussianModel
    sc = synthetic.make_scene(3001, 208, 130, 150.0, sh_degree=3, seed=6, n_objects=2)
    model = SyntheticGaussianModel.from_scene(sc, device="cuda", seed=3)
    ref = SyntheticGaussianModel.from_scene(sc, device="cuda", seed=3)
    w = torch.randn(sc["P"], 3, generator=torch.Generator().manual_seed(1)).cuda()
    pkg = deform.get_deformed_pkg(model, 0.2, want=("xyz",), flow_time=0.7)
    (pkg["flow_xyz"] * w).sum().backward()
    (deform.get_deformed_xyz(ref, 0.7) * w).sum().backward()
    for n in ("_scene_xyz", "_obj_xyz", "xyz_deform_param"):
        close(n, getattr(model, n).grad.cpu().numpy(), getattr(ref, n).grad.cpu().numpy(), tol=1e-6)


def _random_order(rng, quat):
    """A random get_func_result order list: B-spline [n, k], polynomial, Fourier, and (rotation only) a quaternion spline."""
    oa = [0] * 6
    if rng.randint(3) > 0:
        oa[1] = int(rng.randint(1, 6)); oa[0] = int(rng.randint(oa[1] + 1, oa[1] + 12))
    if rng.randint(3) == 0:
        oa[2] = int(rng.randint(1, 5))
    if rng.randint(2) == 0:
        oa[3] = int(rng.randint(1, 7))
    if quat and rng.randint(4) > 0:
        oa[5] = int(rng.randint(1, 6)); oa[4] = int(rng.randint(oa[5] + 1, oa[5] + 10))
    return oa


@pytest.mark.parametrize("seed", range(int(os.environ.get("ADGS_TEST_SEED_BASE", "0")), int(os.environ.get("ADGS_TEST_SEED_BASE", "0")) + int(os.environ.get("ADGS_TEST_DEFORM_SEEDS", "16"))))
def test_random_order_args_forward_vs_oracle_and_backward_vs_float64_autograd(seed):
    """Random basis mixes / row lengths / scene-object splits / time stamps (incl. both ends of the sequence): forward against the
    NumPy oracle, the fused flow points against the oracle's positions at the flow time, every gradient against float64 autograd."""
    from adgs.deform import get_deformed_pkg, get_param_num
    rng = np.random.RandomState(7000 + seed)
    Ns, No = int(rng.choice([0, 5, 300, 1111])), int(rng.choice([1, 64, 257, 700]))
    oa = dict(xyz=_random_order(rng, False), rotation=_random_order(rng, True), shs=_random_order(rng, False), background=_random_order(rng, False))
    use_mask = bool(rng.randint(2))
    g = torch.Generator().manual_seed(seed)
    r = lambda *s: torch.randn(*s, generator=g)
    raw = dict(scene_xyz=r(Ns, 3) * 10, obj_xyz=r(No, 3) * 10, scene_shs_dc=r(Ns, 1, 3), obj_shs_dc=r(No, 1, 3),
               scene_shs_rest=r(Ns, 15, 3) * 0.1, obj_shs_rest=r(No, 15, 3) * 0.1, scene_scaling=r(Ns, 3) * 0.3 - 2,
               obj_scaling=r(No, 3) * 0.3 - 2, scene_rotation=r(Ns, 4), obj_rotation=r(No, 4), scene_opacity=r(Ns, 1), obj_opacity=r(No, 1),
               xyz_deform_param=r(No, 3, get_param_num(oa["xyz"])) * 0.1, rotation_deform_param=r(No, 4, get_param_num(oa["rotation"])) * 0.3,
               shs_deform_param_scene=r(Ns, 3, get_param_num(oa["shs"])) * 0.1, shs_deform_param_obj=r(No, 3, get_param_num(oa["shs"])) * 0.1,
               background_deform_param=r(1, 3, get_param_num(oa["background"])) * 0.1, gs_time=torch.rand(No, 1, generator=g),
               gs_time_sigma=r(No, 2) * 0.3 - 1.0)
    attr_of = lambda k: k if k.endswith("deform_param") or k.startswith("shs_deform") or k.startswith("gs_") else "_" + k
    npm = {k: v.numpy() for k, v in raw.items()}
    npm["order_args"], npm["use_time_mask"] = oa, use_mask
    for t, tf in ((float(rng.rand()), float(rng.rand())), (float(rng.choice([0.0, 1.0])), None)):
        m = _Model()
        for k, v in raw.items():
            setattr(m, attr_of(k), v.cuda().requires_grad_(k != "gs_time"))
        m.order_args, m.use_time_mask = oa, use_mask
        pkg = get_deformed_pkg(m, t, flow_time=tf)
        ref = do.get_deformed_pkg(npm, t)
        for key in ("xyz", "rotation", "shs", "opacity", "scales"):
            close("%s %s t=%g" % (oa, key, t), pkg[key].detach().cpu().numpy(), ref[key])
        if tf is not None:
            close("%s flow_xyz t=%g" % (oa, tf), pkg["flow_xyz"].detach().cpu().numpy(), do.get_deformed_pkg(npm, tf)["xyz"])
        keys = [k for k in ("xyz", "rotation", "shs", "opacity", "scales", "flow_xyz") if k in pkg and torch.is_tensor(pkg[k])]
        ws = {k: torch.randn(pkg[k].shape, generator=g) for k in keys}
        sum((pkg[k] * ws[k].cuda()).sum() for k in keys).backward()
        m64 = {k: v.double().requires_grad_(k != "gs_time") for k, v in raw.items()}
        r64 = tr.get_deformed_pkg(m64, t, oa, use_mask)
        if tf is not None:
            r64["flow_xyz"] = tr.get_deformed_pkg(m64, tf, oa, use_mask)["xyz"]
        sum((r64[k] * ws[k].double()).sum() for k in keys).backward()
        for name, t64 in m64.items():
            if name == "gs_time" or t64.numel() == 0:
                continue
            got, want = getattr(m, attr_of(name)).grad, t64.grad
            if want is None or float(want.abs().max()) == 0.0:
                assert got is None or float(got.abs().max()) == 0.0, (name, oa)
            else:
                assert got is not None, (name, oa)
                close("%s grad %s t=%g" % (oa, name, t), got.cpu().numpy(), want.numpy(), tol=2e-4)


@pytest.mark.parametrize("seed", range(int(os.environ.get("ADGS_TEST_SEED_BASE", "0")), int(os.environ.get("ADGS_TEST_SEED_BASE", "0")) + int(os.environ.get("ADGS_TEST_RENDER_SEEDS", "8"))))
def test_render_entry_fuzz(seed):
    """gaussian_renderer.render() under random option combinations (flow target, object mask, colour override, environment map,
    scale modifier, inverse depth, time mask, models without objects, random basis mixes): the raw-SH model and the model that
    materialises the SH tensor must give the same package and the same gradients on every parameter and on the map."""
    import types
    from adgs import synthetic
    from adgs.env import EnvironmentMap
    from adgs.model import SyntheticGaussianModel
    from gaussian_renderer import render
    rng = np.random.RandomState(14000 + seed)
    sc = synthetic.make_scene(int(rng.choice([60, 1500, 5000])), int(rng.randint(40, 280)), int(rng.randint(30, 180)), float(rng.uniform(80, 250)),
                              sh_degree=int(rng.randint(0, 4)), seed=900 + seed, n_objects=int(rng.randint(0, 4)))
    oa = dict(xyz=_random_order(rng, False), rotation=_random_order(rng, True), shs=_random_order(rng, False), background=_random_order(rng, False))
    if rng.randint(2):
        oa["background"] = [0] * 6
    t = float(rng.rand())
    cam = synthetic.camera_object(sc, time=t)
    flow_pkg = ((float(rng.rand()),) + (None,) * 5) if rng.randint(2) else None
    objmask, use_env, use_mask = bool(rng.randint(2)), bool(rng.randint(2)), bool(rng.randint(2))
    override = torch.rand(sc["P"], 3, generator=torch.Generator().manual_seed(seed)).cuda() if rng.randint(4) == 0 else None
    pipe = types.SimpleNamespace(inv_depth=bool(rng.randint(2)), debug=False)
    smod = float(rng.choice([1.0, 0.8]))
    gen = torch.Generator().manual_seed(100 + seed)
    H, W = sc["H"], sc["W"]
    ws = dict(render=torch.randn(3, H, W, generator=gen).cuda(), depth=torch.randn(H, W, generator=gen).cuda(), img_opacity=torch.randn(H, W, generator=gen).cuda(),
              img_flow=torch.randn(3, H, W, generator=gen).cuda(), img_semantic=torch.randn(1, H, W, generator=gen).cuda())
    res = []
    for raw in (False, True):
        model = SyntheticGaussianModel.from_scene(sc, device="cuda", seed=3, order_args=oa, use_time_mask=use_mask)
        model.raw_sh = raw
        env = None
        if use_env:
            env = EnvironmentMap(64, 3)
            with torch.no_grad():
                env.grid_map.copy_(torch.randn(env.grid_map.shape, generator=torch.Generator().manual_seed(7)).cuda())
        pkg = render(cam, model, env, pipe, scaling_modifier=smod, override_color=override, flow_pkg=flow_pkg, render_objmask=objmask)
        assert (pkg["img_flow"] is not None) == (flow_pkg is not None) and (pkg["img_semantic"] is not None) == objmask
        loss = sum((pkg[k] * w).sum() for k, w in ws.items() if pkg.get(k) is not None)
        loss.backward()
        res.append((pkg, model, env))
    (p0, m0, e0), (p1, m1, e1) = res
    assert torch.equal(p0["radii"], p1["radii"])
    for k in ("render", "depth", "img_opacity", "img_flow", "img_semantic", "foreground", "background"):
        if p0.get(k) is not None:
            close(k, p1[k].detach().cpu().numpy(), p0[k].detach().cpu().numpy(), tol=2e-5)
    close("means2D", p1["viewspace_points"].grad.cpu().numpy(), p0["viewspace_points"].grad.cpu().numpy(), tol=1e-4)
    names = __import__("adgs.model", fromlist=["_RAW"])._RAW
    for name in names:
        a, b = getattr(m0, name, None), getattr(m1, name, None)
        if a is None or a.numel() == 0:
            continue
        ga, gb = a.grad, b.grad
        za = ga is None or float(ga.abs().max()) == 0.0
        zb = gb is None or float(gb.abs().max()) == 0.0
        assert za == zb, (name, za, zb)
        if not za:
            close(name, gb.cpu().numpy(), ga.cpu().numpy(), tol=1e-4)
    if use_env:
        close("grid_map", e1.grid_map.grad.cpu().numpy(), e0.grid_map.grad.cpu().numpy(), tol=1e-4)


@pytest.mark.parametrize("seed,n_objects,flow,use_arena", [(0, 2, True, False), (1, 3, False, False), (2, 2, True, True)])
def test_raw_scene_path_matches_materialised_path(seed, n_objects, flow, use_arena):
    """model.raw_scene (the deformation pass covers the object range only; the rasterizer's preprocess applies exp / normalize /
    sigmoid to the RAW scene tensors and its backward writes their gradients, scene/gaussian_model.py:89-152) == the fully
    materialised path: images, radii and every raw-parameter gradient; render() hides the unwritten scene rows behind lazy entries."""
    from adgs import synthetic, dp
    from adgs.model import SyntheticGaussianModel
    from gaussian_renderer import render
    sc = synthetic.make_scene(9000, 240, 152, 180.0, sh_degree=3, seed=40 + seed, n_objects=n_objects)
    cam = synthetic.camera_object(synthetic.make_camera(240, 152, 180.0, cam_seed=seed), time=0.43)
    g = synthetic.make_upstream_grads(sc, seed)
    d = lambda x: x.cuda()

    class Pipe:
        inv_depth, debug = True, False
    res = []
    for raw_scene in (False, True):
        m = SyntheticGaussianModel.from_scene(sc, device="cuda", seed=5)
        m.raw_sh, m.raw_scene = True, raw_scene
        ex = dp.FactoredSHExchange(m, factor_xyz=False) if (use_arena and raw_scene) else None      # installs the gradient arena
        out = render(cam, m, None, Pipe(), flow_pkg=(0.5,) + (None,) * 5 if flow else None, render_objmask=True,
                     sh_factor_sink=ex.sink_for if ex is not None else None)
        outs = [out["render"], out["depth"], out["img_opacity"], out["img_semantic"]] + ([out["img_flow"]] if flow else [])
        ups = [d(g["color"]), d(g["depth"])[0], d(g["img_opacity"])[0], d(g["semantic"])] + ([d(g["flow"])] if flow else [])
        torch.autograd.backward(outs, ups)
        if ex is not None:
            assert ex.arena.holds("scene_xyz", m._scene_xyz.grad) and ex.arena.holds("scene_opacity", m._scene_opacity.grad)
            ex.reduce([cam.time], [cam.camera_center.tolist()], flow_times=[0.5 if flow else None])
        torch.cuda.synchronize()
        res.append((out, m))
    (o0, m0), (o1, m1) = res
    assert torch.equal(o0["radii"], o1["radii"])
    for k in ("render", "depth", "img_opacity", "img_semantic") + (("img_flow",) if flow else ()):
        close(k, o1[k].detach().cpu().numpy(), o0[k].detach().cpu().numpy(), tol=2e-5)
    close("viewspace", o1["viewspace_points"].grad.cpu().numpy(), o0["viewspace_points"].grad.cpu().numpy(), tol=1e-4)
    from adgs.model import _RAW
    for name in _RAW:
        p0, p1 = getattr(m0, name), getattr(m1, name)
        if p0.numel() == 0:
            continue
        assert (p0.grad is None) == (p1.grad is None), name
        if p0.grad is not None:
            close(name, p1.grad.cpu().numpy(), p0.grad.cpu().numpy(), tol=1e-4)
    # the reference's result entries are still there, produced on demand
    assert "xyz" in o1 and "rotation" in o1
    close("lazy xyz", o1["xyz"].detach().cpu().numpy(), o0["xyz"].detach().cpu().numpy(), tol=1e-6)
    close("lazy opacity", o1["opacity"].detach().cpu().numpy(), o0["opacity"].detach().cpu().numpy(), tol=1e-6)
    # override_color goes through the plain entry: the raw-scene model materialises full rows for it
    col = torch.rand(sc["P"], 3, device="cuda")
    a = render(cam, m0, None, Pipe(), override_color=col)["render"]
    b = render(cam, m1, None, Pipe(), override_color=col)["render"]
    close("override_color", b.detach().cpu().numpy(), a.detach().cpu().numpy(), tol=2e-5)
