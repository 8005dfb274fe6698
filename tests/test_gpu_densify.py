"""GPU parity of the HIP densification (adgs.densify, include/adgs_densify.h) against golden vectors produced by the
REFERENCE's own GaussianModel.densify_and_prune / reset_opacity (tests/golden/make_densify_golden.py) and, at larger random
sizes, against the NumPy oracle (oracle/densify_oracle.py, itself pinned on the same goldens).  Row selection, row order and
the moved data (parameters, Adam moments, gs_time) are bit-exact; the two computed quantities (split positions, split
scales) are within 1e-5 (north_star: 1e-4)."""
import os

import numpy as np
import pytest
import torch

from oracle import densify_oracle as do

pytestmark = pytest.mark.gpu
GOLD = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "densify_golden.npz"))
GROUPS = do.SCENE_GROUPS + do.OBJ_GROUPS + ["deform_background"]


class _Model:
    pass


def _model_from_state(st, args, fused):
    from adgs.densify import GROUP_ATTR
    from adgs.optim import FusedAdam
    m = _Model()
    groups = []
    for g in GROUPS:
        p = torch.nn.Parameter(torch.tensor(st["p"][g], device="cuda"))
        setattr(m, GROUP_ATTR[g], p)
        groups.append({"params": [p], "lr": 1e-3, "name": g})
    m.optimizer = (FusedAdam if fused else torch.optim.Adam)(groups, lr=0.0, eps=1e-15)
    for g in GROUPS:
        if g in st["m"]:
            p = getattr(m, GROUP_ATTR[g])
            m.optimizer.state[p] = dict(step=torch.tensor(float(st.get("step", {}).get(g, 1.0))), exp_avg=torch.tensor(st["m"][g], device="cuda"),
                                        exp_avg_sq=torch.tensor(st["v"][g], device="cuda"))
    m.gs_time = torch.tensor(st["gs_time"], device="cuda")
    for k in ("xyz_gradient_accum", "denom", "max_radii2D"):
        setattr(m, k, torch.tensor(st[k], device="cuda"))
    m.percent_dense, m.scene_extent, m.object_extent = args["percent_dense"], args["scene_extent"], args["object_extent"]
    return m


def _state_of_model(m):
    from adgs.densify import GROUP_ATTR
    st = dict(p={}, m={}, v={})
    for g in GROUPS:
        p = getattr(m, GROUP_ATTR[g])
        st["p"][g] = p.detach().cpu().numpy()
        s = m.optimizer.state.get(p, None)
        if s is not None and "exp_avg" in s:
            st["m"][g], st["v"][g] = s["exp_avg"].cpu().numpy(), s["exp_avg_sq"].cpu().numpy()
    st["gs_time"] = m.gs_time.cpu().numpy()
    for k in ("xyz_gradient_accum", "denom", "max_radii2D"):
        st[k] = getattr(m, k).cpu().numpy()
    return st


def _load_gold(pre):
    st = dict(p={}, m={}, v={}, step={})
    for g in GROUPS:
        st["p"][g] = GOLD[pre + "p_" + g].copy()
        if pre + "m_" + g in GOLD.files:
            st["m"][g], st["v"][g], st["step"][g] = GOLD[pre + "m_" + g].copy(), GOLD[pre + "v_" + g].copy(), float(GOLD[pre + "step_" + g])
    for k in ("gs_time", "xyz_gradient_accum", "denom", "max_radii2D"):
        st[k] = GOLD[pre + k].copy()
    return st


def _args(tag):
    a = GOLD["dp_%s_args" % tag]
    return dict(max_scene_grad=a[0], max_obj_grad=a[1], min_opacity=a[2], prune_big_points=bool(a[3]), percent_dense=a[4], scene_extent=a[5],
                object_extent=a[6])


def _compare(got, want, tol=1e-5):
    for g in GROUPS:
        assert got["p"][g].shape == want["p"][g].shape, (g, got["p"][g].shape, want["p"][g].shape)
        if (g.endswith("_xyz") and not g.startswith("deform")) or g.endswith("_scaling"):
            np.testing.assert_allclose(got["p"][g], want["p"][g], rtol=tol, atol=tol, err_msg=g)
        else:
            assert np.array_equal(got["p"][g], want["p"][g]), g
        assert (g in got["m"]) == (g in want["m"]), g
        if g in want["m"]:
            assert np.array_equal(got["m"][g], want["m"][g]) and np.array_equal(got["v"][g], want["v"][g]), g
    for k in ("gs_time", "xyz_gradient_accum", "denom", "max_radii2D"):
        assert np.array_equal(got[k], want[k]), k


class _FixedNormal:
    """Stands in for torch.normal: hands out the samples the reference drew, checking the std it is asked for."""

    def __init__(self, samples):
        self.samples, self.calls, self.real = list(samples), 0, torch.normal

    def __call__(self, mean=0.0, std=None, **kw):
        s = torch.tensor(self.samples[self.calls], device=std.device).reshape(std.shape)
        self.calls += 1
        assert mean == 0.0 and bool((std > 0).all())
        return s


@pytest.mark.parametrize("fused", [False, True])
@pytest.mark.parametrize("tag", ["small", "big", "none_selected"])
def test_densify_and_prune_vs_reference_golden(tag, fused, monkeypatch):
    from adgs import densify
    a = _args(tag)
    m = _model_from_state(_load_gold("dp_%s_in_" % tag), a, fused)
    fake = _FixedNormal([GOLD["dp_%s_samples_scene" % tag], GOLD["dp_%s_samples_obj" % tag]])
    monkeypatch.setattr(torch, "normal", fake)
    info = densify.densify_and_prune(m, a["max_scene_grad"], a["max_obj_grad"], a["min_opacity"], a["prune_big_points"])
    monkeypatch.undo()
    assert fake.calls == 2
    assert 2 * info["scene"][1] == GOLD["dp_%s_samples_scene" % tag].shape[0] and 2 * info["obj"][1] == GOLD["dp_%s_samples_obj" % tag].shape[0]
    _compare(_state_of_model(m), _load_gold("dp_%s_out_" % tag))
    # the optimizer keeps working on the new tensors (state shapes, group wiring) and step counters survive
    for g in m.optimizer.param_groups:
        p = g["params"][0]
        p.grad = torch.ones_like(p)
    m.optimizer.step()
    for g in ("scene_xyz", "obj_scaling"):
        p = getattr(m, densify.GROUP_ATTR[g])
        assert float(m.optimizer.state[p]["step"]) == 2.0 and m.optimizer.state[p]["exp_avg"].shape == p.shape


def test_reset_opacity_vs_reference_golden():
    from adgs import densify
    m = _model_from_state(_load_gold("ro_in_"), dict(percent_dense=0.01, scene_extent=20.0, object_extent=8.0), True)
    densify.reset_opacity(m)
    got, want = _state_of_model(m), _load_gold("ro_out_")
    for g in GROUPS:
        if g in ("scene_opacity", "obj_opacity"):
            np.testing.assert_allclose(got["p"][g], want["p"][g], rtol=1e-5, atol=1e-5)
            assert not got["m"][g].any() and not got["v"][g].any()
        else:
            assert np.array_equal(got["p"][g], want["p"][g])
            if g in want["m"]:
                assert np.array_equal(got["m"][g], want["m"][g])


@pytest.mark.parametrize("Ns,No,big,seed", [(20000, 6000, True, 0), (7001, 0, False, 1), (0, 4099, True, 2), (50000, 50000, True, 3)])
def test_densify_and_prune_random_sizes_vs_oracle(Ns, No, big, seed):
    _random_sizes_case(Ns, No, big, seed, True)


@pytest.mark.parametrize("seed", range(int(os.environ.get("ADGS_TEST_SEED_BASE", "0")), int(os.environ.get("ADGS_TEST_SEED_BASE", "0")) + int(os.environ.get("ADGS_TEST_DENSIFY_SEEDS", "6"))))
def test_densify_and_prune_fuzz(seed):
    """Random (small, odd, one-sided) sizes: row maps, parameters and both Adam moments bit-exact against the oracle."""
    rng = np.random.default_rng(4000 + seed)
    Ns, No = int(rng.choice([0, 1, 63, 64, 65, 500, 3001])), int(rng.choice([0, 1, 31, 257, 1000, 2049]))
    if Ns + No == 0:
        Ns = 17
    _random_sizes_case(Ns, No, bool(rng.integers(2)), 100 + seed, False)


def _random_sizes_case(Ns, No, big, seed, expect_all):
    from adgs import densify
    rng = np.random.default_rng(seed)
    f = lambda *s: rng.normal(size=s).astype(np.float32)
    p = dict(scene_xyz=f(Ns, 3) * 5, scene_shs_dc=f(Ns, 1, 3), scene_shs_rest=f(Ns, 15, 3), scene_opacity=f(Ns, 1) * 2.5 - 1, scene_scaling=f(Ns, 3) * 0.8 - 1.5,
             scene_rotation=f(Ns, 4), deform_shs_scene=f(Ns, 3, 12), obj_xyz=f(No, 3), obj_shs_dc=f(No, 1, 3), obj_shs_rest=f(No, 15, 3),
             obj_opacity=f(No, 1) * 2.5 - 1, obj_scaling=f(No, 3) * 0.8 - 2, obj_rotation=f(No, 4), deform_xyz=f(No, 3, 18), deform_rotation=f(No, 4, 6),
             deform_shs_obj=f(No, 3, 12), time_sigma=f(No, 2), deform_background=np.zeros((1, 3, 0), np.float32))
    st = dict(p=p, m={k: f(*v.shape) for k, v in p.items() if k != "deform_background"}, v={k: np.abs(f(*v.shape)) for k, v in p.items() if k != "deform_background"},
              gs_time=rng.random((No, 1)).astype(np.float32), xyz_gradient_accum=(rng.random((Ns + No, 1)) * 3e-3).astype(np.float32),
              denom=rng.integers(0, 4, (Ns + No, 1)).astype(np.float32), max_radii2D=(rng.random(Ns + No) * 30).astype(np.float32))
    st["xyz_gradient_accum"] *= (st["denom"] > 0) | (rng.random((Ns + No, 1)) < 0.5)        # a few x/0 = inf rows, many 0/0 = NaN rows
    a = dict(max_scene_grad=8e-4, max_obj_grad=6e-4, min_opacity=0.005, prune_big_points=big, percent_dense=0.01, scene_extent=20.0, object_extent=8.0)
    m = _model_from_state(st, a, True)
    drawn, real = [], torch.normal

    def rec(*args, **kw):
        s = real(*args, **kw); drawn.append(s.cpu().numpy()); return s
    torch.manual_seed(seed)
    torch.normal = rec
    try:
        info = densify.densify_and_prune(m, a["max_scene_grad"], a["max_obj_grad"], a["min_opacity"], a["prune_big_points"])
    finally:
        torch.normal = real
    want = {k: ({kk: vv.copy() for kk, vv in v.items()} if isinstance(v, dict) else v.copy()) for k, v in st.items()}
    do.densify_and_prune(want, a, drawn[0], drawn[1])
    assert (info["scene"][2], info["obj"][2]) == (want["p"]["scene_xyz"].shape[0], want["p"]["obj_xyz"].shape[0])
    if Ns and No and expect_all:
        assert min(info["scene"][0], info["scene"][1], info["obj"][0], info["obj"][1]) > 0
    _compare(_state_of_model(m), want)
    # same seed, same draws: the sample stream is the reference's own torch.normal call
    torch.manual_seed(seed)
    again = [real(mean=0.0, std=torch.tensor(np.abs(d) * 0 + 1.0, device="cuda")) for d in drawn]
    assert all(x.shape == torch.Size(d.shape) for x, d in zip(again, drawn))
