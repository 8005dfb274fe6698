"""GPU parity tests: HIP rasterizer (through the drop-in Python API -> C ABI) vs the CPU
oracle on identical seeded inputs.  Tolerance: 1e-4 (BASELINE.json north_star), integer
outputs (radii, num_rendered) bit-exact.

Threshold note: `alpha < 1/255`, `power > 0` and `T*(1-alpha) < 1e-4` are hard gates on
values that come out of exp(); v_exp/libm differ in the last ulp, so a (pixel, Gaussian)
pair that sits exactly on a gate can flip (the CUDA reference has the same property across
GPUs).  An element outside the tolerance therefore passes only if the oracle's own gate margins
EXPLAIN it (tests/parity.py: explained_masks / raster_oracle.cpp: gate_margins): the pixel's walk came
within float32 rounding error of a gate, or the Gaussian is fed by such a pixel.  Anything else fails.
"""
import math

import numpy as np
import pytest
import torch

from adgs import _lib, synthetic
from oracle import oracle

pytestmark = pytest.mark.gpu

TOL = 1e-4


from tests.parity import assert_close, assert_masked_coverage, explained_masks, mask_upstream  # noqa: E402  (element-wise 1e-4 + relative L2 + per-row relative error; tests/parity.py)


def dev(t):
    return None if t is None else t.cuda()


def run_hip(sc, colors=None, cov3D=None, use_sh=True, flow=True, sem=True, inv_depth=True, scale_modifier=1.0, degree=None,
            semantic=None, bg=None, grads=None, debug=False, strict_mask=None):
    """strict_mask ([H, W] bool, the oracle's gate-flip pixels): a SECOND backward over the same forward state with the upstream gradients
    zeroed there -> res["grads_strict"] (tests/parity.py: the strict gradient pass)."""
    from diff_gaussian_rasterization import GaussianRasterizationSettings, GaussianRasterizer
    settings = GaussianRasterizationSettings(
        image_height=sc["H"], image_width=sc["W"], tanfovx=sc["tanfovx"], tanfovy=sc["tanfovy"],
        bg=dev(sc["bg"] if bg is None else bg), scale_modifier=scale_modifier, viewmatrix=dev(sc["viewmatrix"]),
        projmatrix=dev(sc["projmatrix"]), sh_degree=sc["sh_degree"] if degree is None else degree, campos=dev(sc["campos"]),
        prefiltered=False, inv_depth=inv_depth, debug=debug)
    rast = GaussianRasterizer(settings)
    leaf = lambda t: None if t is None else t.cuda().clone().requires_grad_(True)
    L = dict(means3D=leaf(sc["means3D"]), means2D=torch.zeros(sc["P"], 3, device="cuda", requires_grad=True),
             opacities=leaf(sc["opacities"]), shs=leaf(sc["shs"]) if (use_sh and colors is None) else None,
             colors=leaf(colors), scales=leaf(sc["scales"]) if cov3D is None else None,
             rotations=leaf(sc["rotations"]) if cov3D is None else None, cov3D=leaf(cov3D),
             flow=leaf(sc["flow_points"]) if flow else None,
             sem=leaf(sc["semantic"] if semantic is None else semantic) if sem else None)
    out = rast(means3D=L["means3D"], means2D=L["means2D"], opacities=L["opacities"], shs=L["shs"], colors_precomp=L["colors"],
               scales=L["scales"], rotations=L["rotations"], cov3D_precomp=L["cov3D"], flow_points=L["flow"], semantic=L["sem"])
    color, radii, depth, img_opacity, img_flow, img_sem = out
    res = dict(color=color, radii=radii, depth=depth, img_opacity=img_opacity, img_flow=img_flow, img_semantic=img_sem)
    if grads is not None:
        def total(g):
            loss = (color * dev(g["color"])).sum() + (depth * dev(g["depth"])).sum() + (img_opacity * dev(g["img_opacity"])).sum()
            if flow:
                loss = loss + (img_flow * dev(g["flow"])).sum()
            if sem:
                loss = loss + (img_sem * dev(g["semantic"])).sum()
            return loss
        total(grads).backward(retain_graph=strict_mask is not None)
        res["grads"] = {k: (v.grad if v is not None else None) for k, v in L.items()}
        if strict_mask is not None:
            for v in L.values():
                if v is not None:
                    v.grad = None
            total(mask_upstream(grads, strict_mask)).backward()
            res["grads_strict"] = {k: (v.grad if v is not None else None) for k, v in L.items()}
    torch.cuda.synchronize()
    return res


def run_oracle(sc, colors=None, cov3D=None, use_sh=True, flow=True, sem=True, inv_depth=True, scale_modifier=1.0, degree=None,
               semantic=None, bg=None, grads=None, precision="f32", strict=False, strict_mask=None):
    """strict: also out["grads_strict"], the backward with the upstream gradients zeroed at the gate-flip pixels (`strict_mask`, default
    this run's own out["explained"]["pixel"])."""
    semt = (sc["semantic"] if semantic is None else semantic) if sem else None
    fwd_args = (sc["bg"] if bg is None else bg, sc["means3D"], colors, sc["opacities"],
                None if cov3D is not None else sc["scales"], None if cov3D is not None else sc["rotations"], scale_modifier, cov3D,
                sc["viewmatrix"], sc["projmatrix"], sc["tanfovx"], sc["tanfovy"], sc["H"], sc["W"],
                sc["shs"] if (use_sh and colors is None) else None, sc["flow_points"] if flow else None, semt,
                sc["sh_degree"] if degree is None else degree, sc["campos"], False, inv_depth)
    # Full-size scenes (C2 / C3: 3 - 26 s of oracle time per run on the box's host cores) are run by several tests of a session with
    # identical inputs: the forward state (the oracle object) and every backward over it are memoised on a content hash of the inputs
    # (the GPU suite stood at 48 % of its time limit).  Small scenes are not cached.
    entry = None
    if sc["P"] >= _ORACLE_CACHE_MIN_P:
        key = (precision,) + tuple(_content_key(a) for a in fwd_args)
        entry = _ORACLE_CACHE.get(key)
        if entry is None:
            while len(_ORACLE_CACHE) >= _ORACLE_CACHE_ENTRIES:
                _ORACLE_CACHE.pop(next(iter(_ORACLE_CACHE)))
            entry = _ORACLE_CACHE[key] = dict(o=None, fwd=None, back={})
        else:
            _ORACLE_CACHE[key] = _ORACLE_CACHE.pop(key)          # most recently used last
    if entry is None or entry["o"] is None:
        o = oracle.RasterOracle(precision)
        fwd = o.forward(*fwd_args)
        fwd["explained"] = explained_masks(o.gate_margins())      # which deviations a gate flip may explain (tests/parity.py)
        if entry is not None:
            entry["o"], entry["fwd"] = o, fwd
    else:
        o, fwd = entry["o"], entry["fwd"]
    out = dict(fwd)                                               # callers add keys: never into the cached dictionary
    if grads is not None:
        # unused outputs receive materialised ZERO grads from autograd
        H, W = sc["H"], sc["W"]

        def back(g):
            args = (g["color"], g["depth"], g["flow"] if flow else np.zeros((3, H, W), np.float32), g["semantic"] if sem else None, g["img_opacity"])
            if entry is None:
                return o.backward(*args)
            bkey = tuple(_content_key(a) for a in args)
            if bkey not in entry["back"]:
                entry["back"][bkey] = o.backward(*args)
            return dict(entry["back"][bkey])
        out["grads"] = back(grads)
        if strict or strict_mask is not None:
            gm = mask_upstream({k: (v.numpy() if torch.is_tensor(v) else v) for k, v in grads.items()},
                               out["explained"]["pixel"] if strict_mask is None else strict_mask)
            out["grads_strict"] = back(gm)
    return out


_ORACLE_CACHE, _ORACLE_CACHE_ENTRIES, _ORACLE_CACHE_MIN_P = {}, 3, 200_000


def _content_key(a):
    """Hashable identity of an oracle argument BY CONTENT (tensors / arrays: shape, dtype and a 64-bit hash of the bytes)."""
    if a is None or isinstance(a, (bool, int, float, str)):
        return a
    import xxhash
    arr = np.ascontiguousarray(a.detach().cpu().numpy() if torch.is_tensor(a) else np.asarray(a))
    return (arr.shape, str(arr.dtype), xxhash.xxh64(arr.view(np.uint8).reshape(-1).data).hexdigest())


GRAD_PAIRS = [("means3D", "dL_dmeans3D"), ("means2D", "dL_dmeans2D"), ("opacities", "dL_dopacity"), ("shs", "dL_dsh"),
              ("colors", "dL_dcolors"), ("scales", "dL_dscales"), ("rotations", "dL_drotations"), ("cov3D", "dL_dcov3D"),
              ("flow", "dL_dflow_points"), ("sem", "dL_dsemantic")]


def conditioning_draws(sc, grads, mask, seed=0, **okw):
    """What the fallback of the strict pass needs (tests/parity.py: assert_rows_conditioned), from the oracle alone: the float64 run
    (the exact result) and a second float32 draw -- the float32 oracle on inputs moved by ONE float32 ulp, what any float32 evaluation
    may legitimately mistake the inputs for -- both with the strict pass's upstream mask.  Returns (exact run, perturbed float32 run or
    None when the perturbation changed a cull / radius decision)."""
    exact = run_oracle(sc, grads=grads, precision="f64", strict_mask=mask, **okw)
    prng = np.random.RandomState(77000 + seed)
    sc_p = dict(sc)
    for k in ("means3D", "scales", "rotations", "opacities"):
        sgn = torch.tensor(prng.choice([-1.0, 1.0], size=tuple(sc[k].shape)).astype(np.float32))
        sc_p[k] = (sc[k] * (1.0 + sgn * 2.0 ** -23)).float().contiguous()
    pert = run_oracle(sc_p, grads=grads, strict_mask=mask, **okw)
    return exact, pert


def compare_strict_grads(h, o, label="", draws=None, key="grads_strict"):
    """THE gradient parity check (tests/parity.py): the backward with the upstream gradients zeroed at the oracle's gate-flip pixels, on both
    sides -- every element of every gradient tensor, no exemption.  `draws` (a callable returning conditioning_draws(...)): a tensor that
    fails is re-examined row by row against the float64 oracle with the float32 oracle's own error as the yardstick -- ill-conditioned
    rows (rotation gradients of nearly isotropic Gaussians, ...) are then accepted, anything the float32 oracle itself gets right is not."""
    g, og = h[key], o[key]
    n, failed = 0, []
    for hk, ok in GRAD_PAIRS:
        if g.get(hk) is None:
            continue
        try:
            assert_close(label + "grad_" + hk, g[hk].cpu().numpy(), np.asarray(og[ok]).reshape(g[hk].shape), strict=True)
        except AssertionError as exc:
            if draws is None:
                raise
            failed.append((hk, ok, str(exc).splitlines()[0]))
        n += 1
    if failed:
        exact, pert = draws()
        if not np.array_equal(np.asarray(exact["radii"]), np.asarray(o["radii"])):
            raise AssertionError("%s: %s (and the float64 oracle culls differently: no common ground truth)" % (label, failed[0][2]))
        for hk, ok, msg in failed:
            a = g[hk].cpu().numpy()
            f32 = [np.asarray(og[ok]).reshape(a.shape)]
            if pert is not None and np.array_equal(np.asarray(pert["radii"]), np.asarray(o["radii"])):
                f32.append(np.asarray(pert[key][ok]).reshape(a.shape))
            from tests.parity import assert_rows_conditioned
            assert_rows_conditioned(label + "grad_" + hk, a, f32, np.asarray(exact[key][ok]).reshape(a.shape), context=msg)
    return n


def compare(sc, coverage=None, **kw):
    grads = kw.get("grads")
    okw = dict(kw); okw.pop("debug", None)
    o = run_oracle(sc, strict=grads is not None, **okw)
    ex = o["explained"]
    assert_masked_coverage(ex, **({} if coverage is None else dict(limit=coverage)))
    h = run_hip(sc, strict_mask=ex["pixel"] if grads is not None else None, **kw)
    np.testing.assert_array_equal(h["radii"].cpu().numpy(), o["radii"])
    for k in ("color", "depth", "img_opacity", "img_flow", "img_semantic"):
        assert_close(k, h[k].detach().cpu().numpy(), o[k], explained=ex["pixel"])
    if grads is not None:
        dkw = {k: v for k, v in okw.items() if k != "grads"}
        assert compare_strict_grads(h, o, draws=lambda: conditioning_draws(sc, grads, ex["pixel"], **dkw)) >= 1
        # sanity only (bounds nothing for the rows a flagged pixel feeds -- most rows at full size): the unmasked backward, flips included,
        # at twice the element tolerance (the strict pass above is the gate, with its conditioning fallback; this one has none: fuzz seed
        # 7125 has one SH-gradient element at 1.5 x the tolerance that the strict pass accepts through the float64 yardstick)
        g, og = h["grads"], o["grads"]
        for hk, ok in GRAD_PAIRS:
            if g.get(hk) is None:
                continue
            assert_close("grad_" + hk, g[hk].cpu().numpy(), og[ok].reshape(g[hk].shape), tol=2 * TOL, explained=ex["gauss"])
    return h, o


def test_device_is_gfx950_and_library_loaded():
    from adgs import _lib
    assert _lib.lib().adgs_device_check() == 0, _lib.last_error()


@pytest.mark.parametrize("seed,degree,inv_depth", [(0, 3, True), (1, 2, False), (2, 1, True), (3, 0, True)])
def test_forward_backward_small(seed, degree, inv_depth):
    sc = synthetic.make_scene(3000, 200, 136, 150.0, sh_degree=3, seed=seed, n_objects=2)
    compare(sc, degree=degree, inv_depth=inv_depth, grads=synthetic.make_upstream_grads(sc, seed))


def test_num_rendered_matches_oracle(monkeypatch):
    """In the stage-by-stage "classic" mode even the opaque integer num_rendered equals the reference's;
    the default coarse-binned pipeline returns its (much smaller) number of (cell, Gaussian) pairs."""
    from diff_gaussian_rasterization import _C
    monkeypatch.setenv("ADGS_RASTER_MODE", "classic")
    sc = synthetic.make_scene(5000, 320, 200, 200.0, seed=7)
    e = torch.Tensor([])
    r = _C.rasterize_gaussians(dev(sc["bg"]), dev(sc["means3D"]), e, dev(sc["opacities"]), dev(sc["scales"]), dev(sc["rotations"]), 1.0, e,
                               dev(sc["viewmatrix"]), dev(sc["projmatrix"]), sc["tanfovx"], sc["tanfovy"], sc["H"], sc["W"], dev(sc["shs"]),
                               e, e, 3, dev(sc["campos"]), False, True, False)
    o = run_oracle(sc, flow=False, sem=False)
    assert r[0] == o["num_rendered"] and r[0] > 0
    assert r[5].dtype == torch.uint8 and r[5].numel() > 0 and r[6].numel() > 0 and r[7].numel() > 0
    assert tuple(r[9].shape) == (0, sc["H"], sc["W"])
    monkeypatch.delenv("ADGS_RASTER_MODE")
    r2 = _C.rasterize_gaussians(dev(sc["bg"]), dev(sc["means3D"]), e, dev(sc["opacities"]), dev(sc["scales"]), dev(sc["rotations"]), 1.0, e,
                                dev(sc["viewmatrix"]), dev(sc["projmatrix"]), sc["tanfovx"], sc["tanfovy"], sc["H"], sc["W"], dev(sc["shs"]),
                                e, e, 3, dev(sc["campos"]), False, True, False)
    assert 0 < r2[0] < r[0]
    assert torch.equal(r2[4], r[4])                       # radii do not depend on the pipeline
    assert_close("color v2 vs classic", r2[1].cpu().numpy(), r[1].cpu().numpy(), tol=1e-6)


@pytest.mark.parametrize("mode", ["classic", "v2"])
def test_both_pipelines_match_oracle(monkeypatch, mode):
    monkeypatch.setenv("ADGS_RASTER_MODE", mode)
    sc = synthetic.make_scene(8000, 330, 210, 220.0, seed=21, n_objects=2)
    compare(sc, grads=synthetic.make_upstream_grads(sc, 21))


def test_offscreen_and_transparent_gaussians_v2_rect_shrink():
    """Gaussians whose opacity-aware footprint is off-screen / empty while the reference rectangle is not,
    and opacities around the 1/255 visibility threshold."""
    sc = synthetic.make_scene(6000, 200, 120, 90.0, seed=22, scale_mult=0.02)
    sc["means3D"][:1500, 0] *= 1.6                        # push a quarter far off-axis (left/right of the frustum)
    sc["opacities"][1500:3000] = torch.linspace(0.0005, 0.02, 1500)[:, None]
    compare(sc, grads=synthetic.make_upstream_grads(sc, 22))


def test_c1_config_full_size():
    """BASELINE.json configs[0]: 10k static Gaussians, 400x300, SH degree 0."""
    sc = synthetic.make_config_scene("C1")
    compare(sc, grads=synthetic.make_upstream_grads(sc, 0))


def test_long_tile_lists_multi_batch():
    """Tile lists longer than one 256-entry batch, many contributors per pixel."""
    sc = synthetic.make_scene(20000, 96, 64, 80.0, sh_degree=1, seed=5, scale_mult=0.01)
    sc["opacities"] = sc["opacities"] * 0.05 + 0.01     # transparent: nothing saturates early
    h, o = compare(sc, grads=synthetic.make_upstream_grads(sc, 5))
    assert o["num_rendered"] / (6 * 4) > 600


def test_colors_precomp_and_scale_modifier_and_bg():
    sc = synthetic.make_scene(2500, 160, 120, 120.0, seed=9)
    colors = torch.rand(sc["P"], 3, generator=torch.Generator().manual_seed(1))
    compare(sc, colors=colors, scale_modifier=0.7, bg=torch.tensor([0.2, 0.5, 0.9]), grads=synthetic.make_upstream_grads(sc, 9))


def test_cov3d_precomp():
    sc = synthetic.make_scene(2000, 160, 120, 120.0, seed=10)
    q = sc["rotations"].double()
    r, x, y, z = q[:, 0], q[:, 1], q[:, 2], q[:, 3]
    R = torch.stack([1 - 2 * (y * y + z * z), 2 * (x * y - r * z), 2 * (x * z + r * y), 2 * (x * y + r * z), 1 - 2 * (x * x + z * z),
                     2 * (y * z - r * x), 2 * (x * z - r * y), 2 * (y * z + r * x), 1 - 2 * (x * x + y * y)], 1).reshape(-1, 3, 3)
    Mm = R @ torch.diag_embed(sc["scales"].double())
    Sg = Mm @ Mm.transpose(1, 2)
    cov3D = torch.stack([Sg[:, 0, 0], Sg[:, 0, 1], Sg[:, 0, 2], Sg[:, 1, 1], Sg[:, 1, 2], Sg[:, 2, 2]], 1).float().contiguous()
    compare(sc, cov3D=cov3D, grads=synthetic.make_upstream_grads(sc, 10))


def test_no_flow_no_semantic_no_colour():
    sc = synthetic.make_scene(2000, 130, 70, 100.0, seed=11)      # ragged image size (not a multiple of 16)
    compare(sc, flow=False, sem=False, grads=synthetic.make_upstream_grads(sc, 11))
    h = run_hip(sc, use_sh=False, flow=False, sem=False)           # neither shs nor colours: allowed, colour stays 0
    assert float(h["color"].abs().max()) == 0.0 and float(h["img_opacity"].max()) > 0


@pytest.mark.parametrize("mode", ["default", "classic"])
def test_multi_channel_semantic(monkeypatch, mode):
    if mode == "classic":
        monkeypatch.setenv("ADGS_RASTER_MODE", "classic")
    sc = synthetic.make_scene(1500, 128, 96, 100.0, seed=12)
    sem = torch.rand(sc["P"], 5, generator=torch.Generator().manual_seed(3))
    compare(sc, semantic=sem, grads=synthetic.make_upstream_grads(sc, 12, D_S=5))
    sem32 = torch.rand(sc["P"], 32, generator=torch.Generator().manual_seed(4))
    compare(sc, semantic=sem32, grads=synthetic.make_upstream_grads(sc, 13, D_S=32))


@pytest.mark.parametrize("D_S", [2, 3, 4])
def test_semantic_channels_on_default_pipeline(monkeypatch, D_S):
    """Several semantic channels on the default (v2) pipeline: channel 0 in the main blend kernels, the others by replaying the
    published lists (forward: Horner form of forward.cu:372; backward: one more replay per channel, backward.cu:597-603, whose
    share of dL/dalpha adds into the same per-Gaussian sums).  Against the oracle, and against the classic kernels."""
    from adgs import _lib
    assert _lib.lib().adgs_raster_needs_zero_init(D_S) == 0          # = the default pipeline serves this D_S
    sc = synthetic.make_scene(4000, 260, 150, 180.0, seed=40 + D_S, n_objects=2)
    sem = torch.rand(sc["P"], D_S, generator=torch.Generator().manual_seed(D_S)) * 2 - 0.5
    grads = synthetic.make_upstream_grads(sc, 40 + D_S, D_S=D_S)
    h, _ = compare(sc, semantic=sem, grads=grads)
    assert float(h["grads"]["sem"][:, 1:].abs().max()) > 0
    monkeypatch.setenv("ADGS_RASTER_MODE", "classic")
    c = run_hip(sc, semantic=sem, grads=grads)
    assert_close("sem_vs_classic", h["img_semantic"].detach().cpu().numpy(), c["img_semantic"].detach().cpu().numpy())
    for k in ("means3D", "opacities", "scales", "rotations", "sem"):
        assert_close("grad_vs_classic_" + k, h["grads"][k].cpu().numpy(), c["grads"][k].cpu().numpy(), max_frac=max(2e-4, 4.5 / h["grads"][k].numel()))


def test_semantic_channels_limit_raised_and_semantic_only_backward(monkeypatch):
    """ADGS_V2_MAX_SEMANTIC moves the boundary to the classic kernels; 7 channels = two forward replays of 4 + 2 channels; a backward
    that carries only a semantic gradient (every other upstream gradient zero) exercises the extra replays on their own."""
    from adgs import _lib
    assert _lib.lib().adgs_raster_needs_zero_init(32) == 0
    monkeypatch.setenv("ADGS_V2_MAX_SEMANTIC", "8")
    assert _lib.lib().adgs_raster_needs_zero_init(7) == 0 and _lib.lib().adgs_raster_needs_zero_init(9) == 1
    sc = synthetic.make_scene(2500, 200, 120, 150.0, seed=51)
    sem = torch.rand(sc["P"], 7, generator=torch.Generator().manual_seed(5))
    grads = synthetic.make_upstream_grads(sc, 51, D_S=7)
    compare(sc, semantic=sem, grads=grads)
    only = {k: (v if k == "semantic" else torch.zeros_like(v)) for k, v in grads.items()}
    compare(sc, semantic=sem, grads=only)


def test_empty_and_all_culled():
    from diff_gaussian_rasterization import GaussianRasterizationSettings, GaussianRasterizer
    sc = synthetic.make_scene(64, 64, 48, 60.0, seed=13)
    settings = GaussianRasterizationSettings(48, 64, sc["tanfovx"], sc["tanfovy"], dev(sc["bg"]), 1.0, dev(sc["viewmatrix"]),
                                             dev(sc["projmatrix"]), 3, dev(sc["campos"]), False, True, False)
    rast = GaussianRasterizer(settings)
    z = lambda *s: torch.zeros(*s, device="cuda")
    # P == 0 -> zero outputs (rasterize_points.cu:99)
    out = rast(means3D=z(0, 3), means2D=z(0, 3), opacities=z(0, 1), shs=z(0, 16, 3), scales=z(0, 3), rotations=z(0, 4))
    assert out[0].shape == (3, 48, 64) and float(out[0].abs().sum()) == 0 and out[1].numel() == 0
    # everything behind the camera -> num_rendered == 0, backward must still run
    m = sc["means3D"].clone(); m[:, 2] = -m[:, 2].abs() - 1
    sc2 = dict(sc); sc2["means3D"] = m
    h = run_hip(sc2, grads=synthetic.make_upstream_grads(sc2, 1))
    assert int(h["radii"].max()) == 0 and float(h["color"].abs().max()) == 0
    assert float(h["grads"]["means3D"].abs().max()) == 0


def test_argument_validation_matches_reference():
    from diff_gaussian_rasterization import GaussianRasterizationSettings, GaussianRasterizer
    sc = synthetic.make_scene(16, 32, 32, 30.0, seed=1)
    s = GaussianRasterizationSettings(32, 32, sc["tanfovx"], sc["tanfovy"], dev(sc["bg"]), 1.0, dev(sc["viewmatrix"]), dev(sc["projmatrix"]),
                                      3, dev(sc["campos"]), False, True, False)
    r = GaussianRasterizer(s)
    m, o, sh, sc_, ro = dev(sc["means3D"]), dev(sc["opacities"]), dev(sc["shs"]), dev(sc["scales"]), dev(sc["rotations"])
    with pytest.raises(Exception):
        r(means3D=m, means2D=m, opacities=o, shs=sh, colors_precomp=m, scales=sc_, rotations=ro)
    with pytest.raises(Exception):
        r(means3D=m, means2D=m, opacities=o, shs=sh, scales=sc_)            # rotations missing
    with pytest.raises(Exception):
        r(means3D=m, means2D=m, opacities=o, shs=sh, scales=sc_, rotations=ro, cov3D_precomp=torch.zeros(16, 6, device="cuda"))
    with pytest.raises(RuntimeError):
        r(means3D=m[:, :2], means2D=m, opacities=o, shs=sh, scales=sc_, rotations=ro)
    with pytest.raises(RuntimeError):                                      # CPU tensors: loud failure, no fallback
        r(means3D=sc["means3D"], means2D=sc["means3D"], opacities=sc["opacities"], shs=sc["shs"], scales=sc["scales"], rotations=sc["rotations"])


def test_mark_visible_and_debug_mode():
    from diff_gaussian_rasterization import GaussianRasterizationSettings, GaussianRasterizer
    sc = synthetic.make_scene(4000, 96, 64, 80.0, seed=14, near_frac=0.3)
    s = GaussianRasterizationSettings(64, 96, sc["tanfovx"], sc["tanfovy"], dev(sc["bg"]), 1.0, dev(sc["viewmatrix"]), dev(sc["projmatrix"]),
                                      3, dev(sc["campos"]), False, True, True)
    vis = GaussianRasterizer(s).markVisible(dev(sc["means3D"]))
    ref = oracle.RasterOracle("f32").mark_visible(sc["means3D"], sc["viewmatrix"], sc["projmatrix"])
    assert vis.dtype == torch.bool
    np.testing.assert_array_equal(vis.cpu().numpy(), ref)
    compare(sc, debug=True, grads=synthetic.make_upstream_grads(sc, 14))


def test_opacity_only_gradient_quirk_matches():
    """Only grad_img_opacity non-zero: exercises the reference's T-scaled opacity term (backward.cu:612-614)."""
    sc = synthetic.make_scene(3000, 128, 96, 100.0, seed=15)
    g = synthetic.make_upstream_grads(sc, 15)
    for k in ("color", "depth", "flow", "semantic"):
        g[k] = torch.zeros_like(g[k])
    compare(sc, grads=g)


def test_deterministic_forward_and_stable_backward():
    sc = synthetic.make_scene(6000, 160, 120, 120.0, seed=16)
    g = synthetic.make_upstream_grads(sc, 16)
    a = run_hip(sc, grads=g)
    b = run_hip(sc, grads=g)
    for k in ("color", "depth", "img_opacity", "img_flow", "img_semantic", "radii"):
        assert torch.equal(a[k], b[k]), k                        # forward is bit-reproducible
    for k, v in a["grads"].items():
        if v is not None:                                        # backward sums with fp32 atomics: order may vary
            assert_close("rerun_" + k, b["grads"][k].cpu().numpy(), v.cpu().numpy(), tol=1e-5, max_frac=0)


def test_c2_sized_scene_properties_and_parity():
    """BASELINE.json configs[1] (300k Gaussians, 1242x375, SH 3): oracle parity at full size."""
    sc = synthetic.make_config_scene("C2")
    g = synthetic.make_upstream_grads(sc, 1)
    h, o = compare(sc, grads=g)
    # size-independent properties
    op = h["img_opacity"]
    assert float(op.min()) >= 0 and float(op.max()) <= 1.0
    assert float(h["depth"].min()) >= 0
    # linearity of the backward in the upstream gradient (same forward state)
    g2 = {k: 2.0 * v for k, v in g.items()}
    h2 = run_hip(sc, grads=g2)
    for k in ("means3D", "opacities", "shs", "scales", "rotations"):
        # two separate backward runs: the per-Gaussian sums are accumulated with float atomics in a different order each time, so
        # "exactly twice" holds to accumulation rounding only (observed 2.3e-5 of the tensor's scale on 1 of 1.2 M elements)
        assert_close("lin_" + k, h2["grads"][k].cpu().numpy(), 2.0 * h["grads"][k].cpu().numpy(), tol=5e-5, max_frac=0)


@pytest.mark.parametrize("mode", ["classic", "v2"])
@pytest.mark.parametrize("used", [("color",), ("depth",), ("img_opacity", "img_flow"), ("img_semantic",)])
def test_outputs_the_loss_does_not_use(monkeypatch, mode, used):
    """A loss built from a subset of the six outputs: the unused ones reach the kernels as absent (NULL) gradients -- no zero
    fills -- and the result equals the oracle's with zero upstream gradients there."""
    from diff_gaussian_rasterization import GaussianRasterizationSettings, GaussianRasterizer
    monkeypatch.setenv("ADGS_RASTER_MODE", mode)
    sc = synthetic.make_scene(1500, 112, 80, 90.0, sh_degree=3, seed=21)
    g = synthetic.make_upstream_grads(sc, 4)
    s = GaussianRasterizationSettings(sc["H"], sc["W"], sc["tanfovx"], sc["tanfovy"], dev(sc["bg"]), 1.0, dev(sc["viewmatrix"]), dev(sc["projmatrix"]), 3,
                                      dev(sc["campos"]), False, True, False)
    L = {k: sc[k].cuda().clone().requires_grad_(True) for k in ("means3D", "opacities", "shs", "scales", "rotations", "flow_points", "semantic")}
    m2 = torch.zeros(sc["P"], 3, device="cuda", requires_grad=True)
    out = GaussianRasterizer(s)(means3D=L["means3D"], means2D=m2, opacities=L["opacities"], shs=L["shs"], scales=L["scales"], rotations=L["rotations"],
                                flow_points=L["flow_points"], semantic=L["semantic"])
    outs = dict(zip(("color", "radii", "depth", "img_opacity", "img_flow", "img_semantic"), out))
    up = dict(color=g["color"], depth=g["depth"], img_opacity=g["img_opacity"], img_flow=g["flow"], img_semantic=g["semantic"])
    sum((outs[k] * up[k].cuda()).sum() for k in used).backward()
    torch.cuda.synchronize()
    z = lambda k: up[k].numpy() if k in used else np.zeros_like(up[k].numpy())
    o = oracle.RasterOracle("f32")
    o.forward(sc["bg"], sc["means3D"], None, sc["opacities"], sc["scales"], sc["rotations"], 1.0, None, sc["viewmatrix"], sc["projmatrix"], sc["tanfovx"],
              sc["tanfovy"], sc["H"], sc["W"], sc["shs"], sc["flow_points"], sc["semantic"], 3, sc["campos"], False, True)
    og = o.backward(z("color"), z("depth"), z("img_flow"), z("img_semantic"), z("img_opacity"))
    for hk, ok in (("means3D", "dL_dmeans3D"), ("opacities", "dL_dopacity"), ("shs", "dL_dsh"), ("scales", "dL_dscales"), ("rotations", "dL_drotations"),
                   ("flow_points", "dL_dflow_points"), ("semantic", "dL_dsemantic")):
        got = L[hk].grad
        want = og[ok].reshape(L[hk].shape)
        if got is None:
            assert not want.any(), hk
        else:
            assert_close("grad_" + hk, got.cpu().numpy(), want, max_frac=2e-4)
    assert_close("grad_means2D", m2.grad.cpu().numpy(), og["dL_dmeans2D"], max_frac=2e-4)


def test_c3_full_size_two_independent_pipelines_agree(monkeypatch):
    """BASELINE.json configs[2] at full size (1 M Gaussians, 1920x1280, flow + semantic): the coarse-binned v2 pipeline and the
    classic reference-order pipeline (full (tile | depth) sort, per-pixel atomics) are independent implementations of the same
    operator; images, radii and every gradient must agree, and the pixel budget is conserved (img_opacity = 1 - T in [0, 1])."""
    sc = synthetic.make_config_scene("C3")
    g = synthetic.make_upstream_grads(sc, 2)
    res = {}
    for mode in ("v2", "classic"):
        monkeypatch.setenv("ADGS_RASTER_MODE", mode)
        res[mode] = run_hip(sc, grads=g)
    a, b = res["v2"], res["classic"]
    assert torch.equal(a["radii"], b["radii"]) and int((a["radii"] > 0).sum()) > 800_000
    for k in ("color", "depth", "img_opacity", "img_flow", "img_semantic"):
        assert_close(k, a[k].detach().cpu().numpy(), b[k].detach().cpu().numpy(), max_frac=5e-6)
    for k in ("means3D", "means2D", "opacities", "shs", "scales", "rotations", "flow", "sem"):
        assert_close("grad_" + k, a["grads"][k].cpu().numpy(), b["grads"][k].cpu().numpy(), max_frac=2e-4)
    op = a["img_opacity"]
    assert float(op.min()) >= 0.0 and float(op.max()) <= 1.0 + 1e-6


def test_c3_full_size_vs_oracle():
    """BASELINE.json configs[2] at full size against the CPU oracle (the line-by-line restatement of the reference kernels):
    radii bit-exact, the five images and all gradients within 1e-4 (about 10 s of oracle time on the GPU box's host cores)."""
    sc = synthetic.make_config_scene("C3")
    h, o = compare(sc, grads=synthetic.make_upstream_grads(sc, 2))
    assert o["num_rendered"] > 40_000_000                       # the reference's (tile, Gaussian) pair count of this frame


def test_backward_uses_the_forwards_configuration_not_the_environment(monkeypatch):
    """Pipeline, cell size and pixels per lane are read from the environment by the FORWARD and remembered with its state buffers:
    changing the environment between forward and backward must not change how the backward carves them."""
    from diff_gaussian_rasterization import GaussianRasterizationSettings, GaussianRasterizer
    sc = synthetic.make_scene(6000, 320, 200, 300.0, seed=23, n_objects=2)
    g = synthetic.make_upstream_grads(sc, 23)
    ref = run_hip(sc, grads=g)
    s = GaussianRasterizationSettings(sc["H"], sc["W"], sc["tanfovx"], sc["tanfovy"], dev(sc["bg"]), 1.0, dev(sc["viewmatrix"]), dev(sc["projmatrix"]), 3,
                                      dev(sc["campos"]), False, True, False)
    L = {k: sc[k].cuda().clone().requires_grad_(True) for k in ("means3D", "opacities", "shs", "scales", "rotations")}
    m2 = torch.zeros(sc["P"], 3, device="cuda", requires_grad=True)
    out = GaussianRasterizer(s)(means3D=L["means3D"], means2D=m2, opacities=L["opacities"], shs=L["shs"], scales=L["scales"], rotations=L["rotations"],
                                flow_points=dev(sc["flow_points"]), semantic=dev(sc["semantic"]))
    monkeypatch.setenv("ADGS_CELL_TILES", "3")
    monkeypatch.setenv("ADGS_V2_PPL", "4")
    monkeypatch.setenv("ADGS_RASTER_MODE", "classic")
    monkeypatch.setenv("ADGS_TILE_ORDER", "0")
    monkeypatch.setenv("ADGS_NO_SH_STAGING", "1")
    reads = _lib.lib().adgs_test_env_reads()
    torch.autograd.backward([out[0], out[2], out[3], out[4], out[5]], [dev(g["color"]), dev(g["depth"]), dev(g["img_opacity"]), dev(g["flow"]), dev(g["semantic"])])
    torch.cuda.synchronize()
    assert _lib.lib().adgs_test_env_reads() == reads, "the backward read the environment: the forward decides, the frame state carries it"
    for k in ("means3D", "opacities", "shs", "scales", "rotations"):
        assert_close("grad_" + k, L[k].grad.cpu().numpy(), ref["grads"][k].cpu().numpy(), tol=2e-5, max_frac=1e-4, rel_l2=2e-5)      # two runs: the order of the float atomics differs


def test_backward_over_cloned_state_reads_the_forwards_configuration_from_the_state(monkeypatch):
    """Saved tensors that reach the backward as COPIES (offloaded / cloned state buffers) are unknown to the library's frame table.  The
    forward's preprocess kernel wrote the frame's configuration into the header of the image state: the backward reads that word back --
    not the environment, which has changed in between -- and carves the copies the way the forward carved the originals."""
    from diff_gaussian_rasterization import _C
    sc = synthetic.make_scene(5000, 300, 190, 250.0, seed=31, n_objects=2)
    g = synthetic.make_upstream_grads(sc, 31)
    e = torch.Tensor([])
    args_f = lambda: (dev(sc["bg"]), dev(sc["means3D"]), e, dev(sc["opacities"]), dev(sc["scales"]), dev(sc["rotations"]), 1.0, e, dev(sc["viewmatrix"]),
                      dev(sc["projmatrix"]), sc["tanfovx"], sc["tanfovy"], sc["H"], sc["W"], dev(sc["shs"]), dev(sc["flow_points"]), dev(sc["semantic"]), 3,
                      dev(sc["campos"]), False, True, False)

    def backward(r, geom, binning, img):
        return _C.rasterize_gaussians_backward(dev(sc["bg"]), dev(sc["means3D"]), r[4], e, dev(sc["scales"]), dev(sc["rotations"]), 1.0, e, dev(sc["viewmatrix"]),
                                               dev(sc["projmatrix"]), sc["tanfovx"], sc["tanfovy"], dev(g["color"]), dev(g["depth"]), dev(g["flow"]), dev(g["semantic"]),
                                               dev(sc["semantic"]), dev(sc["flow_points"]), dev(sc["shs"]), 3, dev(sc["campos"]), geom, r[0], binning, img, r[3],
                                               dev(g["img_opacity"]), True, False)
    for env in (dict(ADGS_CELL_TILES="5", ADGS_V2_PPL="4"), dict(ADGS_RASTER_MODE="classic"), {}):
        for k, v in env.items():
            monkeypatch.setenv(k, v)
        r = _C.rasterize_gaussians(*args_f())
        for k in env:
            monkeypatch.delenv(k)
        want = backward(r, r[5], r[6], r[7])                       # the forward's own buffers: frame table
        reads = _lib.lib().adgs_test_env_reads()
        got = backward(r, r[5].clone(), r[6].clone(), r[7].clone())      # copies: the header word
        assert _lib.lib().adgs_test_env_reads() == reads
        torch.cuda.synchronize()
        for a, b in zip(got, want):
            assert a.shape == b.shape
            assert_close("cloned state " + str(env), a.cpu().numpy(), b.cpu().numpy(), tol=2e-5, max_frac=1e-4, rel_l2=2e-5)      # two runs: the order of the float atomics differs
    junk = torch.zeros_like(r[7])
    with pytest.raises(RuntimeError):                              # an image state no forward wrote: refused, not mis-carved
        backward(r, r[5].clone(), r[6].clone(), junk)


def test_forward_tile_order_hint_changes_nothing_but_the_schedule(monkeypatch):
    """Round 5: the blend forward walks its tiles longest-first by the PREVIOUS render of the same camera (the library keeps one hint per
    camera, recognised by the device addresses of its matrices; api.hip: OrderHints).  Any order must give the same frame bit for bit --
    first render of a camera (bottom-up), second render (its own hint), a hint left by a different scene behind the same camera, and
    ADGS_FWD_ORDER=1 / 0 -- and the backward (which gets its order from the forward now) the same gradients."""
    from diff_gaussian_rasterization import GaussianRasterizationSettings, GaussianRasterizer
    sc = synthetic.make_scene(60000, 1024, 1024, 900.0, seed=81, n_objects=3, scale_mult=0.006)       # 4096 wave tiles: the ordered path
    other = synthetic.make_scene(60000, 1024, 1024, 900.0, seed=82, n_objects=3, scale_mult=0.012)
    g = synthetic.make_upstream_grads(sc, 81)
    view, proj, campos, bg = dev(sc["viewmatrix"]), dev(sc["projmatrix"]), dev(sc["campos"]), dev(sc["bg"])       # ONE camera: the same matrices for every render below
    settings = GaussianRasterizationSettings(sc["H"], sc["W"], sc["tanfovx"], sc["tanfovy"], bg, 1.0, view, proj, 3, campos, False, True, False)
    rast = GaussianRasterizer(settings)

    def render(scene, with_grads=True):
        L = {k: scene[k].cuda().clone().requires_grad_(True) for k in ("means3D", "opacities", "shs", "scales", "rotations")}
        m2 = torch.zeros(scene["P"], 3, device="cuda", requires_grad=True)
        out = rast(means3D=L["means3D"], means2D=m2, opacities=L["opacities"], shs=L["shs"], scales=L["scales"], rotations=L["rotations"],
                   flow_points=dev(scene["flow_points"]), semantic=dev(scene["semantic"]))
        grads = None
        if with_grads:
            torch.autograd.backward([out[0], out[2], out[3], out[4], out[5]], [dev(g["color"]), dev(g["depth"]), dev(g["img_opacity"]), dev(g["flow"]), dev(g["semantic"])])
            grads = {k: v.grad.clone() for k, v in L.items()}
        torch.cuda.synchronize()
        return [o.detach().clone() for o in out], grads
    first, g_first = render(sc)                       # no hint yet: bottom-up
    assert _lib.frame_stats()["tiles"] >= 2048
    second, g_second = render(sc)                     # the camera's own hint
    render(other, with_grads=False)                   # another scene behind the same camera leaves ITS order as the hint ...
    third, g_third = render(sc)                       # ... which is a poor guess and still only a schedule
    runs = [(second, g_second), (third, g_third)]
    for mode in ("1", "0"):
        monkeypatch.setenv("ADGS_FWD_ORDER", mode)
        runs.append(render(sc))
    monkeypatch.delenv("ADGS_FWD_ORDER")
    for outs, grads in runs:
        for a, b in zip(outs, first):
            assert torch.equal(a, b)                  # images, radii: bit for bit
        for k, v in g_first.items():                  # gradients: float atomics in another order
            assert_close("grad_" + k, grads[k].cpu().numpy(), v.cpu().numpy(), tol=2e-5, max_frac=1e-4, rel_l2=2e-5)


def test_repeated_backward_over_one_forward_state():
    """retain_graph: the per-Gaussian accumulator lines of a forward are consumed by its first backward and zeroed again (by the
    library, api.hip: note_backward) before every further one -- three backward passes over one forward give the same gradients
    (up to the summation order of the atomics), and so does a backward whose forward has dropped out of the library's frame table."""
    from diff_gaussian_rasterization import GaussianRasterizationSettings, GaussianRasterizer
    sc = synthetic.make_scene(30000, 320, 208, 300.0, seed=61, n_objects=2)
    g = synthetic.make_upstream_grads(sc, 61)
    settings = GaussianRasterizationSettings(
        image_height=sc["H"], image_width=sc["W"], tanfovx=sc["tanfovx"], tanfovy=sc["tanfovy"], bg=dev(sc["bg"]), scale_modifier=1.0,
        viewmatrix=dev(sc["viewmatrix"]), projmatrix=dev(sc["projmatrix"]), sh_degree=sc["sh_degree"], campos=dev(sc["campos"]),
        prefiltered=False, inv_depth=True, debug=False)
    rast = GaussianRasterizer(settings)
    leaf = lambda t: t.cuda().clone().requires_grad_(True)
    L = [leaf(sc["means3D"]), torch.zeros(sc["P"], 3, device="cuda", requires_grad=True), leaf(sc["opacities"]), leaf(sc["shs"]),
         leaf(sc["scales"]), leaf(sc["rotations"]), leaf(sc["flow_points"]), leaf(sc["semantic"])]

    def forward():
        color, radii, depth, op, fl, sem = rast(means3D=L[0], means2D=L[1], opacities=L[2], shs=L[3], scales=L[4], rotations=L[5],
                                               flow_points=L[6], semantic=L[7])
        return ((color * dev(g["color"])).sum() + (depth * dev(g["depth"])).sum() + (op * dev(g["img_opacity"])).sum() +
                (fl * dev(g["flow"])).sum() + (sem * dev(g["semantic"])).sum())
    loss = forward()
    runs = [torch.autograd.grad(loss, L, retain_graph=True) for _ in range(3)]
    for later in runs[1:]:
        for a, b in zip(runs[0], later):
            assert torch.allclose(a, b, rtol=2e-4, atol=1e-6), float((a - b).abs().max())
    assert float(runs[0][0].abs().sum()) > 0
    # more forwards than the frame table holds in between: the backward of the first one no longer finds its entry
    loss0 = forward()
    for _ in range(260):                      # (training forwards: under no_grad they would be forward-only renders, which leave the table alone)
        forward()
    late = torch.autograd.grad(loss0, L)
    for a, b in zip(runs[0], late):
        assert torch.allclose(a, b, rtol=2e-4, atol=1e-6), float((a - b).abs().max())


def test_many_chunks_per_tile_at_four_pixels_per_lane(monkeypatch):
    """A translucent scene on a 4096-tile image (the 16x16-tile, four-pixels-per-lane kernels): nothing saturates, every tile blends
    and publishes hundreds of entries -- far more than the four chunk slots a tile owns, so most slots come in blocks from the shared
    cursor -- and the backward replays all of them.  The reference-order pipeline (independent binning, blend and backward) must agree."""
    sc = synthetic.make_scene(120000, 1024, 1024, 900.0, seed=71, n_objects=3, scale_mult=0.006)
    sc["opacities"] = (sc["opacities"] * 0.05 + 0.01).contiguous()
    g = synthetic.make_upstream_grads(sc, 71)
    res = {}
    for mode in ("v2", "classic"):
        monkeypatch.setenv("ADGS_RASTER_MODE", mode)
        res[mode] = run_hip(sc, grads=g)
        if mode == "v2":
            assert _lib.frame_stats()["tiles"] == 4096
    a, b = res["v2"], res["classic"]
    assert torch.equal(a["radii"], b["radii"])
    for k in ("color", "depth", "img_opacity", "img_flow", "img_semantic"):
        assert_close(k, a[k].detach().cpu().numpy(), b[k].detach().cpu().numpy(), max_frac=5e-6)
    for k in ("means3D", "means2D", "opacities", "shs", "scales", "rotations", "flow", "sem"):
        assert_close("grad_" + k, a["grads"][k].cpu().numpy(), b["grads"][k].cpu().numpy(), max_frac=2e-4)
    # the walk did not stop early: the mean opacity stays well below saturation and the classic pair count per tile is in the hundreds
    assert float(a["img_opacity"].detach().mean()) < 0.995
    monkeypatch.setenv("ADGS_RASTER_MODE", "classic")
    run_hip(sc)
    assert _lib.frame_stats()["num_rendered"] / 4096 > 300          # classic mode: (tile, Gaussian) pairs


def test_opacities_above_the_alpha_clamp():
    """Opacities in (0.99, 1]: the min(0.99, .) clamp of forward.cu:357 bites (alpha = 0.99 at the centre) for 40 % of the Gaussians."""
    sc = synthetic.make_scene(12000, 256, 160, 220.0, seed=73, n_objects=2)
    g = torch.Generator().manual_seed(3)
    op = sc["opacities"].clone()
    hot = torch.rand(op.shape[0], generator=g) < 0.4
    op[hot] = 0.99 + 0.01 * torch.rand(int(hot.sum()), 1, generator=g)
    op[0] = 1.0
    sc["opacities"] = op.contiguous()
    compare(sc, grads=synthetic.make_upstream_grads(sc, 73))
