"""Fused Adam against the optimizer the reference actually uses: torch.optim.Adam(l, lr=0.0, eps=1e-15)
(scene/gaussian_model.py:370), run on the CPU in float64 (tight bound) and float32 (same-precision sanity)."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _groups(seed, dtype, device):
    g = torch.Generator().manual_seed(seed)
    shapes = [((1000, 3), 0.0), ((1000, 1, 3), 2.5e-3), ((1000, 15, 3), 2.5e-3 / 20), ((1000, 1), 0.05), ((1000, 3), 5e-3), ((1000, 4), 1e-3),
              ((257, 3, 18), 1.6e-4), ((257, 4, 6), 1e-3), ((7,), 1e-2), ((0, 3), 1e-2)]
    ps = [torch.randn(*s, generator=g).to(dtype).to(device).requires_grad_(True) for s, _ in shapes]
    return [{"params": [p], "lr": lr, "name": "g%d" % i} for i, (p, (_, lr)) in enumerate(zip(ps, shapes))]


def _run(opt_cls, groups, dtype, device, steps, **kw):
    opt = opt_cls(groups, lr=0.0, eps=1e-15, **kw)
    g = torch.Generator().manual_seed(99)
    for it in range(steps):
        for gi, group in enumerate(opt.param_groups):
            p = group["params"][0]
            if gi == 3 and it % 2 == 1:
                p.grad = None                      # a parameter without gradient is skipped (no state update)
            else:
                p.grad = (torch.randn(*p.shape, generator=g) * (10.0 ** (gi % 4 - 2))).to(dtype).to(device)
            if gi == 4:
                group["lr"] = 5e-3 * (0.9 ** it)   # schedulers write group['lr'] (train.py / update_learning_rate)
        opt.step()
    return opt


@pytest.mark.parametrize("steps", [1, 7])
def test_fused_adam_matches_torch_adam(steps):
    from adgs.optim import FusedAdam
    dev = torch.device("cuda")
    fused = _run(FusedAdam, _groups(0, torch.float32, dev), torch.float32, dev, steps)
    ref64 = _run(torch.optim.Adam, _groups(0, torch.float64, "cpu"), torch.float64, "cpu", steps)
    ref32 = _run(torch.optim.Adam, _groups(0, torch.float32, "cpu"), torch.float32, "cpu", steps)
    for gf, g64, g32 in zip(fused.param_groups, ref64.param_groups, ref32.param_groups):
        pf, p64, p32 = gf["params"][0], g64["params"][0], g32["params"][0]
        if pf.numel() == 0:
            continue
        a = pf.detach().cpu().double().numpy(); b = p64.detach().numpy(); c = p32.detach().double().numpy()
        scale = max(np.abs(b).max(), 1e-30)
        err, err32 = np.abs(a - b).max() / scale, np.abs(c - b).max() / scale
        assert err <= max(3e-6, 4 * err32), (gf["name"], err, err32)
        sf, s64 = fused.state[pf], ref64.state[p64]
        assert float(sf["step"]) == float(s64["step"])
        for k in ("exp_avg", "exp_avg_sq"):
            x, y = sf[k].cpu().double().numpy(), s64[k].numpy()
            np.testing.assert_allclose(x, y, rtol=2e-5, atol=2e-6 * max(np.abs(y).max(), 1e-30), err_msg=gf["name"] + k)


def test_fused_adam_zero_grad_and_state_keys():
    from adgs.optim import FusedAdam
    dev = torch.device("cuda")
    p = torch.randn(1001, 5, device=dev, requires_grad=True)
    q = torch.randn(33, device=dev, requires_grad=True)
    opt = FusedAdam([{"params": [p], "lr": 1e-2, "name": "p"}, {"params": [q], "lr": 0.0, "name": "q"}], lr=0.0, eps=1e-15)
    p.grad, q.grad = torch.randn_like(p), torch.randn_like(q)
    q0 = q.detach().clone()
    opt.step(zero_grad="zeros")                                      # zero_grad(set_to_none=False): zero-filled by the same kernel pass
    assert float(p.grad.abs().max()) == 0.0 and float(q.grad.abs().max()) == 0.0
    p.grad.copy_(torch.randn_like(p))
    q.grad = None
    steps = (float(opt.state[p]["step"]), float(opt.state[q]["step"]))
    opt.step(zero_grad=True)                                         # the reference's zero_grad(set_to_none=True), train.py:165
    assert p.grad is None and q.grad is None
    # a parameter without gradient is skipped like torch.optim.Adam skips it: no step count, no moment-driven update
    assert (float(opt.state[p]["step"]), float(opt.state[q]["step"])) == (steps[0] + 1, steps[1])
    with pytest.raises(ValueError):
        opt.step(zero_grad="none")
    p.grad, q.grad = torch.randn_like(p), torch.randn_like(q)
    q0 = q.detach().clone()
    opt.step()
    assert torch.equal(q.detach(), q0)                               # lr = 0 leaves the parameter untouched, but the moments move
    st = opt.state[q]
    assert set(st.keys()) == {"step", "exp_avg", "exp_avg_sq"} and float(st["exp_avg"].abs().max()) > 0
    sd = opt.state_dict()                                            # torch.optim.Optimizer plumbing (checkpointing) works
    opt.load_state_dict(sd)
    cpu_p = torch.zeros(3, requires_grad=True)
    cpu_p.grad = torch.ones(3)
    with pytest.raises(RuntimeError):                                # no CPU fallback
        FusedAdam([cpu_p], lr=1e-3).step()


def test_densification_stats_match_the_reference_lines():
    """train.py:148-150 and scene/gaussian_model.py:863-867 restated with the reference's own torch expressions on the CPU."""
    from adgs.optim import add_densification_stats
    g = torch.Generator().manual_seed(5)
    N = 100003
    radii = torch.randint(-1, 40, (N,), generator=g, dtype=torch.int32).clamp_min(0)
    grad = torch.randn(N, 3, generator=g)
    accum, denom, maxr = torch.rand(N, 1, generator=g), torch.randint(0, 5, (N, 1), generator=g).float(), torch.randint(0, 30, (N,), generator=g).float()
    d_acc, d_den, d_max = accum.cuda(), denom.cuda(), maxr.cuda()
    for _ in range(2):
        add_densification_stats(d_acc, d_den, d_max, grad.cuda(), radii.cuda())
        vis = radii > 0
        maxr[vis] = torch.max(maxr[vis], radii[vis])
        accum[vis] += torch.norm(grad[vis, :2], dim=-1, keepdim=True)
        denom[vis] += 1
    assert torch.equal(d_den.cpu(), denom) and torch.equal(d_max.cpu(), maxr)
    assert torch.allclose(d_acc.cpu(), accum, rtol=1e-6, atol=1e-7)


def test_fused_adam_survives_the_references_optimizer_state_surgery():
    """scene/gaussian_model.py:560-582 (_prune_optimizer) and :616-638 (cat_tensors_to_optimizer) edit optimizer.state and
    param_groups in place; the same edits on FusedAdam (GPU) and torch.optim.Adam (CPU, float64) must keep them in step."""
    from torch import nn
    from adgs.optim import FusedAdam

    def surgery(opt, mask, extension):
        for group in opt.param_groups:
            st = opt.state.get(group["params"][0], None)
            # prune
            st["exp_avg"], st["exp_avg_sq"] = st["exp_avg"][mask.to(st["exp_avg"].device)], st["exp_avg_sq"][mask.to(st["exp_avg"].device)]
            del opt.state[group["params"][0]]
            group["params"][0] = nn.Parameter(group["params"][0][mask.to(group["params"][0].device)].requires_grad_(True))
            opt.state[group["params"][0]] = st
            # densify
            ext = extension.to(group["params"][0].device, group["params"][0].dtype)
            st = opt.state.get(group["params"][0], None)
            st["exp_avg"] = torch.cat((st["exp_avg"], torch.zeros_like(ext)), dim=0)
            st["exp_avg_sq"] = torch.cat((st["exp_avg_sq"], torch.zeros_like(ext)), dim=0)
            del opt.state[group["params"][0]]
            group["params"][0] = nn.Parameter(torch.cat((group["params"][0], ext), dim=0).requires_grad_(True))
            opt.state[group["params"][0]] = st

    g = torch.Generator().manual_seed(3)
    p0 = torch.randn(500, 3, generator=g)
    grads = [torch.randn(500, 3, generator=g), torch.randn(470, 3, generator=g), torch.randn(470, 3, generator=g)]
    mask = torch.rand(500, generator=g) > 0.1
    n_keep = int(mask.sum())
    ext = torch.randn(470 - n_keep, 3, generator=g) if n_keep < 470 else torch.zeros(0, 3)
    grads[1], grads[2] = grads[1][: n_keep + ext.shape[0]], grads[2][: n_keep + ext.shape[0]]
    results = []
    for cls, dev, dt in ((FusedAdam, "cuda", torch.float32), (torch.optim.Adam, "cpu", torch.float64)):
        p = nn.Parameter(p0.to(dev, dt).clone())
        opt = cls([{"params": [p], "lr": 1e-2, "name": "scene_xyz"}], lr=0.0, eps=1e-15)
        opt.param_groups[0]["params"][0].grad = grads[0].to(dev, dt)
        opt.step()
        surgery(opt, mask, ext)
        for gr in grads[1:]:
            opt.param_groups[0]["params"][0].grad = gr.to(dev, dt)
            opt.step()
        results.append(opt.param_groups[0]["params"][0].detach().cpu().double())
    assert results[0].shape == results[1].shape
    assert torch.allclose(results[0], results[1], rtol=1e-5, atol=1e-6)


@pytest.mark.parametrize("seed", range(int(__import__("os").environ.get("ADGS_TEST_SEED_BASE", "0")), int(__import__("os").environ.get("ADGS_TEST_SEED_BASE", "0")) + int(__import__("os").environ.get("ADGS_TEST_ADAM_SEEDS", "8"))))
def test_fused_adam_fuzz(seed):
    """Random tensor shapes (empty, one element, sizes around the kernel's vector width), learning rates incl. 0, betas,
    gradients that are None on random steps: parameters and both moments against float64 torch.optim.Adam."""
    from adgs.optim import FusedAdam
    rng = np.random.default_rng(2000 + seed)
    n_groups = int(rng.integers(1, 7))
    shapes = [tuple(int(v) for v in rng.choice([0, 1, 2, 3, 4, 5, 63, 64, 65, 255, 1000], size=int(rng.integers(1, 4)))) for _ in range(n_groups)]
    lrs = [float(rng.choice([0.0, 1e-4, 1e-2, 0.3])) for _ in range(n_groups)]
    betas = (float(rng.choice([0.9, 0.5, 0.0])), float(rng.choice([0.999, 0.9])))
    steps = int(rng.integers(1, 6))
    init = [rng.normal(size=s) for s in shapes]
    grads = [[None if rng.random() < 0.2 else rng.normal(size=s) * 10.0 ** float(rng.integers(-3, 3)) for s in shapes] for _ in range(steps)]

    def run(cls, dtype, device):
        ps = [torch.tensor(a, dtype=dtype, device=device).requires_grad_(True) for a in init]
        opt = cls([{"params": [p], "lr": lr, "name": "g%d" % i} for i, (p, lr) in enumerate(zip(ps, lrs))], lr=0.0, eps=1e-15, betas=betas)
        for gs in grads:
            for p, g in zip(ps, gs):
                p.grad = None if g is None else torch.tensor(g, dtype=dtype, device=device)
            opt.step()
        return ps, opt
    pf, of = run(FusedAdam, torch.float32, "cuda")
    p64, o64 = run(torch.optim.Adam, torch.float64, "cpu")
    p32, _ = run(torch.optim.Adam, torch.float32, "cpu")
    for a, b, c, s in zip(pf, p64, p32, shapes):
        if a.numel() == 0:
            continue
        x, y, z = a.detach().cpu().double().numpy(), b.detach().numpy(), c.detach().double().numpy()
        scale = max(np.abs(y).max(), 1e-30)
        err, err32 = np.abs(x - y).max() / scale, np.abs(z - y).max() / scale
        assert err <= max(3e-6, 4 * err32), (s, err, err32)
        assert (a in of.state) == (b in o64.state), s
        if b in o64.state:
            assert float(of.state[a]["step"]) == float(o64.state[b]["step"])
            # the moments are updated as m += (1 - beta) (g - m) (torch's lerp): in fp32 the absolute error is a few ulp of the
            # LARGEST gradient the moment has seen, not of its final value
            idx = shapes.index(s) if shapes.count(s) == 1 else [j for j, q in enumerate(pf) if q is a][0]
            hist = max([float(np.abs(gs[idx]).max()) for gs in grads if gs[idx] is not None] + [1e-30])
            for k, bound in (("exp_avg", hist), ("exp_avg_sq", hist * hist)):
                u, v = of.state[a][k].cpu().double().numpy(), o64.state[b][k].numpy()
                np.testing.assert_allclose(u, v, rtol=2e-5, atol=1e-6 * bound, err_msg=str(s) + k)


def test_dormant_tiles_are_skipped_exactly():
    """skip_dormant_tiles: a tile of 256 elements (several kernel blocks, a partial last tile) that has never seen a non-zero gradient is left untouched -- bit for bit what the
    dense step does to it -- and wakes up for good with its first non-zero gradient; moments from elsewhere switch the shortcut off."""
    from adgs.optim import FusedAdam, ADAM_TILE
    torch.manual_seed(3)
    n = 37 * ADAM_TILE + 123
    p0 = torch.randn(n, device="cuda")
    a = torch.nn.Parameter(p0.clone()); b = torch.nn.Parameter(p0.clone())
    oa = FusedAdam([{"params": [a], "lr": 1e-2}], eps=1e-15, skip_dormant_tiles=True)
    ob = FusedAdam([{"params": [b], "lr": 1e-2}], eps=1e-15)
    touched = [set(), {1}, {1}, {1, 3}, {5, 20}, set(), {3, 37}]          # tiles with a non-zero gradient in step i
    seen = set()
    for step, tiles in enumerate(touched):
        g = torch.zeros(n, device="cuda")
        for t in tiles:
            lo, hi = t * ADAM_TILE, min((t + 1) * ADAM_TILE, n)
            g[lo + 7:hi:97] = torch.randn(len(range(lo + 7, hi, 97)), device="cuda")
        seen |= tiles
        a.grad, b.grad = g.clone(), g.clone()
        oa.step(); ob.step()
        assert torch.equal(a, b), step
        for k in ("exp_avg", "exp_avg_sq"):
            assert torch.equal(oa.state[a][k], ob.state[b][k]), (step, k)
        amap = oa._tile_maps[id(a)][0].cpu().tolist()
        assert [i for i, v in enumerate(amap) if v] == sorted(seen), (step, amap)
    assert torch.equal(a[:ADAM_TILE], p0[:ADAM_TILE]) and torch.equal(a[2 * ADAM_TILE:3 * ADAM_TILE], p0[2 * ADAM_TILE:3 * ADAM_TILE])   # never touched
    # moments replaced behind the optimizer's back (what densification / load_state_dict do): every tile is treated as active
    oa.state[a]["exp_avg"] = oa.state[a]["exp_avg"].clone() + 0.5
    ob.state[b]["exp_avg"] = ob.state[b]["exp_avg"].clone() + 0.5
    a.grad, b.grad = torch.zeros(n, device="cuda"), torch.zeros(n, device="cuda")
    oa.step(); ob.step()
    assert torch.equal(a, b) and not torch.equal(a[:ADAM_TILE], p0[:ADAM_TILE])
    assert all(oa._tile_maps[id(a)][0].cpu().tolist())


# ---------------------------------------------------------------- the Adam step inside the rasterizer's backward (adgs_sh_adam)
def _render_model(P, n_objects, seed, dev, in_backward, W=176, H=112):
    from adgs import synthetic
    from adgs.model import SyntheticGaussianModel
    import bench
    sc = synthetic.to_z_up_world(synthetic.make_scene(P, W, H, 140.0, sh_degree=3, seed=seed, n_objects=n_objects))
    model = SyntheticGaussianModel.from_scene(sc, device=dev, seed=seed)
    model.raw_sh = model.raw_scene = True
    lrs = {"scene_shs_rest": 1.25e-4, "obj_shs_rest": 2.5e-4, "deform_shs_scene": 1e-3, "deform_shs_obj": 2e-3, "scene_shs_dc": 2.5e-3, "obj_shs_dc": 2.5e-3}
    model.training_setup(lrs=lrs, scene_extent=20.0, object_extent=4.0, near_num=0, adam_in_backward=in_backward)
    cams = []
    for cam, t in bench.camera_pool(dict(P=P, W=W, H=H, focal=140.0, sh_degree=3, n_objects=n_objects, seed=seed), 3):
        c = synthetic.camera_object(synthetic.camera_to_z_up(cam), time=t)
        for name in ("world_view_transform", "full_proj_transform", "camera_center"):
            setattr(c, name, getattr(c, name).to(dev))
        cams.append(c)
    return model, cams


def _one_iteration(model, cam, weights, arm):
    import types
    from gaussian_renderer import render
    pkg = render(cam, model, None, types.SimpleNamespace(inv_depth=True, debug=False), render_objmask=True)
    loss = (pkg["render"] * weights[0]).sum() + (pkg["depth"] * weights[1]).sum() + (pkg["img_opacity"] * weights[2]).sum()
    if arm:
        model.optimizer.arm_backward()
    loss.backward()
    fused = {g["name"]: g["params"][0].grad is None for g in model.optimizer.param_groups}
    model.optimizer.step(zero_grad=True)
    return fused


@pytest.mark.parametrize("P,n_objects,seed,W,H", [(5000, 2, 3, 32, 4), (4099, 3, 4, 32, 4), (700, 1, 5, 32, 4)])
def test_adam_in_the_backward_is_the_two_kernel_step_bit_for_bit(P, n_objects, seed, W, H):
    """FusedAdam(in_backward=True) + arm_backward(): the rasterizer's backward applies the step of the SH rest tensors (preprocess
    backward, from its LDS gradient rows) and of the SH deformation rows (the pass that expands dL/d(dc)) itself.  Against the same
    model stepped by adgs_adam_step from materialised gradients: parameters, both moments and the step counters of EVERY group
    bit-identical after every iteration -- armed and unarmed iterations mixed, three cameras, scene / object boundaries that are and
    are not multiples of the block size and of the 16-byte quads (the straddling block goes element by element), Gaussians outside
    the frustum (zero gradient rows: Adam still moves them by their moments).

    Two runs of the blend backward agree bit for bit only where no Gaussian sums more than two tiles (its per-(tile, Gaussian)
    float atomics commute, they do not associate): the 32 x 4 frames have two tiles.  Full-size frames on both paths against the CPU
    trajectory: tests/test_gpu_trajectory.py."""
    dev = torch.device("cuda", 0)
    exact, same = True, torch.equal
    a, cams = _render_model(P, n_objects, seed, dev, True, W, H)
    b, _ = _render_model(P, n_objects, seed, dev, False, W, H)
    g = torch.Generator().manual_seed(seed)
    H, W = int(cams[0].image_height), int(cams[0].image_width)
    names = ("scene_shs_rest", "obj_shs_rest", "deform_shs_scene", "deform_shs_obj")
    for it in range(7):
        cam = cams[it % len(cams)]
        weights = [torch.randn(3, H, W, generator=g).to(dev), torch.randn(H, W, generator=g).to(dev) * 0.1, torch.randn(H, W, generator=g).to(dev) * 0.1]
        arm = it not in (2, 5)                       # unarmed iterations go through the gradient tensors on both models
        if it == 4:                                  # schedulers write group['lr'] between iterations
            for m in (a, b):
                for grp in m.optimizer.param_groups:
                    grp["lr"] *= 0.7
        fa = _one_iteration(a, cam, weights, arm)
        fb = _one_iteration(b, cam, weights, False)
        for n in names:
            assert fa[n] == arm and not fb[n], (it, n, fa, fb)      # the fused tensors never had a gradient tensor
        assert all(fa[k] == fb[k] for k in fa if k not in names), (fa, fb)
        torch.cuda.synchronize()
        for ga, gb in zip(a.optimizer.param_groups, b.optimizer.param_groups):
            pa, pb = ga["params"][0], gb["params"][0]
            if pa.numel() == 0 or (pa.grad is None and pa not in a.optimizer.state):
                continue
            assert same(pa.detach(), pb.detach()), (it, ga["name"], float((pa.detach() - pb.detach()).abs().max()))
            sa, sb = a.optimizer.state[pa], b.optimizer.state[pb]
            assert int(sa["step"]) == int(sb["step"]) == it + 1, (it, ga["name"])
            assert same(sa["exp_avg"], sb["exp_avg"]) and same(sa["exp_avg_sq"], sb["exp_avg_sq"]), (it, ga["name"])
    if exact:
        seen = a.optimizer.state[a._scene_shs_rest]["exp_avg_sq"].flatten(1).amax(1)
        assert bool((seen > 0).any()) and bool((seen == 0).any()), "the frames should leave Gaussians inside and outside the frustum"


def test_adam_in_the_backward_refuses_what_it_cannot_do_exactly():
    from adgs.optim import FusedAdam
    dev = torch.device("cuda", 0)
    m, cams = _render_model(900, 1, 8, dev, True)
    H, W = int(cams[0].image_height), int(cams[0].image_width)
    w = [torch.ones(3, H, W, device=dev), torch.zeros(H, W, device=dev), torch.zeros(H, W, device=dev)]
    # a second armed backward before step()
    import types
    from gaussian_renderer import render
    pipe = types.SimpleNamespace(inv_depth=True, debug=False)
    m.optimizer.arm_backward()
    render(cams[0], m, None, pipe)["render"].sum().backward()
    with pytest.raises(RuntimeError, match="has not been followed by step"):
        m.optimizer.arm_backward()
    # a gradient from elsewhere on a tensor the backward has already stepped
    m._scene_shs_rest.grad = torch.zeros_like(m._scene_shs_rest)
    with pytest.raises(RuntimeError, match="received a gradient from elsewhere"):
        m.optimizer.step(zero_grad=True)
    m._scene_shs_rest.grad = None
    m.optimizer.step(zero_grad=True)
    # an optimizer built without in_backward
    plain = FusedAdam([{"params": [torch.zeros(4, device=dev, requires_grad=True)], "lr": 1e-3}], lr=0.0)
    with pytest.raises(RuntimeError, match="in_backward=True"):
        plain.arm_backward()
    # unarmed: nothing changes about the ordinary path
    fused = _one_iteration(m, cams[1], w, False)
    assert not any(fused[n] for n in ("scene_shs_rest", "obj_shs_rest", "deform_shs_scene", "deform_shs_obj")), fused
