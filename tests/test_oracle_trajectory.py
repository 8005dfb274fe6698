"""The CPU reference trajectory (tests/trajectory_ref.py, the checker of tests/test_gpu_trajectory.py) checked on its own: its Adam
against torch.optim.Adam in float64, and the gradient of its composed total loss (chain rule through deformation, rasterizer,
environment map, six image losses, three regularisers) against central differences of that loss along random directions."""
import numpy as np
import torch

from tests import trajectory_ref as tj


def test_adam_step_equals_torch_adam_float64():
    rng = np.random.default_rng(0)
    p0 = rng.normal(size=(50, 3)).astype(np.float32)
    t = torch.nn.Parameter(torch.tensor(p0, dtype=torch.float64))
    opt = torch.optim.Adam([{"params": [t], "lr": 3e-3}], lr=0.0, eps=1e-15)
    p, m, v = p0.astype(np.float64), np.zeros_like(p0, dtype=np.float64), np.zeros_like(p0, dtype=np.float64)
    for step in range(1, 8):
        g = rng.normal(size=p0.shape) * 10.0 ** rng.integers(-8, 1)
        t.grad = torch.tensor(g)
        opt.step()
        # float64 parameter here (the trainer rounds to float32 each step, as the reference's float32 parameters do)
        m = tj.BETA1 * m + (1 - tj.BETA1) * g
        v = tj.BETA2 * v + (1 - tj.BETA2) * g * g
        p = p - (3e-3 / (1 - tj.BETA1 ** step)) * m / (np.sqrt(v) / np.sqrt(1 - tj.BETA2 ** step) + tj.EPS)
        np.testing.assert_allclose(t.detach().numpy(), p, rtol=1e-12, atol=1e-14)
        q, m2, v2 = tj.adam_step(p0 if step == 1 else q, g, (m - (1 - tj.BETA1) * g) / tj.BETA1, (v - (1 - tj.BETA2) * g * g) / tj.BETA2, step, 3e-3)
        np.testing.assert_allclose(m2, m, rtol=1e-12); np.testing.assert_allclose(v2, v, rtol=1e-12)
        np.testing.assert_allclose(q, p, rtol=2e-6, atol=1e-7)          # the float32 rounding of the stored parameter


def _trainer(case, env_res=16, opacity_shift=0.0):
    cfg, sc, model, cams = case
    from tests import chain_ref
    raw = chain_ref.raw_numpy(model)
    raw["scene_opacity"] = raw["scene_opacity"] + np.float32(opacity_shift); raw["obj_opacity"] = raw["obj_opacity"] + np.float32(opacity_shift)
    env = (np.random.default_rng(3).normal(size=(1, 3, env_res, env_res)) * 0.5).astype(np.float32)
    tr = tj.RefTrainer(raw, tj.LRS, model.order_args, model.use_time_mask, tj.WEIGHTS, 3, 20.0, 4.0, 0.01, 0.02, 8, env_grid=env, env_lr=1e-2)
    tr.set_obj_near_idx(np.random.default_rng(1).permutation(tr.n_obj))
    return tr


def _directional(tr, args, group, h, rng):
    """(central difference of the total loss along a random +-1 direction of one parameter group, analytic directional derivative)."""
    base = tr.loss_and_grads(*args)
    if group == "ENV":
        x0, g = tr.env["p"].copy(), base["env_grad"][None]
    else:
        x0, g = tr.st["p"][group].copy(), base["grads"][group].reshape(tr.st["p"][group].shape)
    d = rng.choice([-1.0, 1.0], size=x0.shape)
    vals = []
    for sgn in (+1, -1):
        x = (x0.astype(np.float64) + sgn * h * d).astype(np.float32)
        if group == "ENV":
            tr.env["p"] = x
        else:
            tr.st["p"][group] = x
        vals.append(tr.loss_and_grads(*args)["total"])
    if group == "ENV":
        tr.env["p"] = x0
    else:
        tr.st["p"][group] = x0
    return (vals[0] - vals[1]) / (2 * h), float((g * d).sum()), base


def test_total_loss_gradient_against_central_differences():
    """What this can and cannot pin: the total loss is only piecewise smooth in the geometry and the opacities (the rasterizer's hard
    gates -- alpha >= 1/255 cuts every footprint off at a jump, the BCE clip has a slope of 1000 below 0.999 and 0 above), and the
    float32 deformation adds 1e-7 of rounding noise to every evaluation, so a finite step along those directions measures the jumps.
    Exact where the loss is smooth (colours, environment map, regularisers).  The paths through opacity and geometry are covered by the
    per-stage pins (every loss oracle, the deformation and the environment map on the reference's own autograd, tests/golden; the
    rasterizer's backward on torch.autograd of tests/torch_ref.py) and by tests/test_gpu_trajectory.py, where an independent
    implementation of the same iteration (the HIP path) must reproduce this trajectory."""
    case = tj.build_case(P=300, W=64, H=40, focal=50.0, n_objects=2, seed=3, n_cameras=1)
    c = case[3][0]
    args = (c["cam"], c["time"], c["flow_pkg"], c["targets"], c["env_cam"])
    rng = np.random.default_rng(5)
    tr = _trainer(case, opacity_shift=-2.0)
    tr.precision = "f64"          # the float32 rasterizer's rounding noise (1e-7 of the loss) is of the size of the differences taken here
    tr.st["p"]["time_sigma"] = (tr.st["p"]["time_sigma"] + rng.normal(size=tr.st["p"]["time_sigma"].shape) * 0.3).astype(np.float32)
    base = tr.loss_and_grads(*args)
    assert np.isfinite(base["total"]) and all(np.isfinite(v) for v in base["terms"].values())
    assert min(base["terms"][k] for k in ("l1", "dssim", "depth", "flow", "obj", "sky", "reg", "sigma", "reg_sigma")) > 0
    # every term at once, smooth directions
    for group, h, tol in (("scene_shs_dc", 2e-3, 0.03), ("obj_shs_dc", 2e-3, 0.03), ("obj_shs_rest", 1e-3, 0.05), ("ENV", 1e-2, 0.03)):
        fd, an, _ = _directional(tr, args, group, h, rng)
        assert abs(fd - an) <= tol * max(abs(an), abs(fd)) + 2e-6, (group, fd, an)
    # one term at a time
    full = dict(tr.w)
    for only, group, h, tol in ((("reg",), "deform_xyz", 1e-3, 1e-4), (("sigma", "reg_sigma"), "time_sigma", 1e-3, 1e-4),
                                (("l1",), "scene_shs_dc", 2e-3, 0.08), (("dssim",), "scene_shs_dc", 2e-3, 0.08)):
        tr.w = {k: (v if k in only else 0.0) for k, v in full.items()}
        fd, an, _ = _directional(tr, args, group, h, rng)
        assert an != 0.0 and abs(fd - an) <= tol * max(abs(an), abs(fd)) + 1e-9, (only, group, fd, an)
    tr.w = full


def test_reference_trajectory_runs_and_learns():
    """A few iterations incl. a densification and an opacity reset: finite, point count changes, losses fall on a reachable target."""
    case = tj.build_case(P=400, W=64, H=40, focal=50.0, n_objects=2, seed=4, n_cameras=1)
    tr = _trainer(case, env_res=8)
    c = case[3][0]
    rng = np.random.default_rng(2)
    totals = []
    n0 = tr.n_scene + tr.n_obj
    for it in range(6):
        r = tr.loss_and_grads(c["cam"], c["time"], c["flow_pkg"], c["targets"], c["env_cam"])
        totals.append(r["total"])
        tr.add_densification_stats(r["chain"])
        if it == 3:
            acc = (tr.st["xyz_gradient_accum"] / np.maximum(tr.st["denom"], 1)).reshape(-1)
            thr = tj.gap_threshold(acc, 0.8)
            sel = tr.split_parents(thr, thr)
            tr.densify_and_prune(thr, thr, 0.005, False, rng.normal(size=(2 * int(sel["scene"].sum()), 3)), rng.normal(size=(2 * int(sel["obj"].sum()), 3)))
            tr.set_obj_near_idx(rng.permutation(tr.n_obj))
            tr.reset_opacity()
        tr.optimizer_step(r["grads"], r["env_grad"])
    assert all(np.isfinite(t) for t in totals) and tr.n_scene + tr.n_obj != n0
    assert totals[2] < totals[0]
    assert tr.st["m"]["scene_xyz"].shape == tr.st["p"]["scene_xyz"].shape and tr.st["xyz_gradient_accum"].shape[0] == tr.n_scene + tr.n_obj
    assert float(dz_sigmoid(tr.st["p"]["scene_opacity"]).max()) <= 0.0100001 + 0.06          # reset to <= 0.01, then at most one Adam step of lr 0.05 in logit space


def dz_sigmoid(x):
    return 1.0 / (1.0 + np.exp(-np.asarray(x, np.float64)))
