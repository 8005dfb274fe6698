"""HIP environment-map background (adgs.env) through the C ABI against the reference's golden vectors and the NumPy oracle."""
import os
import types

import numpy as np
import pytest
import torch

from oracle import env_oracle

pytestmark = pytest.mark.gpu
GOLD = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "env_golden.npz"))
CASES = sorted({k.split("/")[0] for k in GOLD.files})


@pytest.mark.parametrize("case", CASES)
def test_background_and_gradient_match_reference_golden(case):
    from adgs import env
    gm = torch.tensor(GOLD[case + "/grid_map"])[None].cuda().requires_grad_(True)
    H, W = [int(v) for v in GOLD[case + "/HW"]]
    bg = env.image_background(gm, H, W, float(GOLD[case + "/focal"]), GOLD[case + "/R"].tolist())
    ref = GOLD[case + "/bg"]
    assert np.abs(bg.detach().cpu().numpy() - ref).max() <= 3e-5
    (bg * torch.tensor(GOLD[case + "/w"]).cuda()).sum().backward()
    gref = GOLD[case + "/g_grid"]
    assert np.abs(gm.grad.cpu().numpy()[0] - gref).max() <= 1e-4 * max(np.abs(gref).max(), 1.0)


def test_environment_map_class_full_resolution_and_adam():
    """1920x1280 over a 2048^2 map: values in (0,1), gradient mass conservation (sum of the bilinear weights is 1 inside the map),
    camera cache, and one fused Adam step on the map."""
    from adgs import env
    e = env.EnvironmentMap(2048, 3)
    with torch.no_grad():
        e.grid_map.copy_(torch.randn(e.grid_map.shape, generator=torch.Generator().manual_seed(0)).cuda())
    w2v = torch.eye(4); w2v[:3, :3] = torch.tensor([[0.8, 0.0, 0.6], [0.0, 1.0, 0.0], [-0.6, 0.0, 0.8]])
    cam = types.SimpleNamespace(FoVx=0.87, image_width=1920, image_height=1280, world_view_transform=w2v.cuda(), cam_id=3)
    bg = e.get_image_background(cam)
    assert bg.shape == (3, 1280, 1920) and float(bg.min()) > 0 and float(bg.max()) < 1
    assert 3 in e._cam_cache
    small = env_oracle.background(e.grid_map.detach().cpu().numpy()[0], 1280, 1920, env.fov2focal(0.87, 1920), w2v[:3, :3].numpy())[:, ::97, ::131]
    # fp32 ray arithmetic: ~1e-4 texel on a 2048 map, times the texel-to-texel contrast of a white-noise map
    assert np.abs(bg.detach().cpu().numpy()[:, ::97, ::131] - small).max() <= 2e-3
    up = torch.ones_like(bg)
    bg.backward(up)
    mass = float((bg.detach() * (1 - bg.detach())).double().sum())
    assert abs(float(e.grid_map.grad.double().sum()) - mass) <= 1e-4 * mass
    e.training_setup(types.SimpleNamespace(env_lr=1e-2))
    before = e.grid_map.detach().clone()
    e.optimizer.step(zero_grad="zeros")
    assert float((e.grid_map.detach() - before).abs().max()) > 0 and float(e.grid_map.grad.abs().max()) == 0.0
    e.optimizer.step(zero_grad=True)
    assert e.grid_map.grad is None


@pytest.mark.parametrize("seed", range(int(os.environ.get("ADGS_TEST_SEED_BASE", "0")), int(os.environ.get("ADGS_TEST_SEED_BASE", "0")) + int(os.environ.get("ADGS_TEST_ENV_SEEDS", "10"))))
def test_random_maps_cameras_and_image_shapes_vs_oracle(seed):
    """Random map resolutions (down to 2x2, non-square), image shapes, focal lengths and rotations (incl. rays that look
    backwards / straight up): forward and the map gradient against the NumPy oracle."""
    from adgs import env
    rng = np.random.default_rng(8000 + seed)
    Hm, Wm = int(rng.choice([2, 3, 7, 16, 33, 128])), int(rng.choice([2, 3, 5, 16, 64, 129]))      # (maps below 2x2 are rejected)
    H, W = int(rng.choice([1, 3, 16, 37, 90])), int(rng.choice([1, 5, 16, 33, 121]))
    q = rng.normal(size=4); q /= np.linalg.norm(q)
    w, x, y, z = q
    R = np.array([[1 - 2 * (y * y + z * z), 2 * (x * y - z * w), 2 * (x * z + y * w)],
                  [2 * (x * y + z * w), 1 - 2 * (x * x + z * z), 2 * (y * z - x * w)],
                  [2 * (x * z - y * w), 2 * (y * z + x * w), 1 - 2 * (x * x + y * y)]], np.float32)
    focal = float(rng.uniform(0.3, 3.0) * W)
    gm_np = rng.normal(size=(3, Hm, Wm)).astype(np.float32)
    gm = torch.tensor(gm_np)[None].cuda().requires_grad_(True)
    bg = env.image_background(gm, H, W, focal, R.tolist())
    wts = rng.normal(size=(3, H, W)).astype(np.float32)
    (bg * torch.tensor(wts).cuda()).sum().backward()
    ref = env_oracle.background(gm_np, H, W, focal, R)
    gref = env_oracle.background_grad(gm_np, H, W, focal, R, wts)
    # fp32 ray arithmetic moves a sample by ~1e-6 of the map; on a white-noise map that is ~1e-6 * resolution * contrast
    tol = 3e-5 + 4e-6 * max(Hm, Wm)
    got = bg.detach().cpu().numpy()
    bad = np.abs(got - ref) > tol
    assert bad.sum() <= max(3, 1e-3 * bad.size), (Hm, Wm, H, W, int(bad.sum()), float(np.abs(got - ref).max()))   # a sample on a texel edge may round across it
    gbad = np.abs(gm.grad.cpu().numpy()[0] - gref) > 1e-4 * max(np.abs(gref).max(), 1.0) + tol * np.abs(wts).max() * 4
    assert gbad.sum() <= max(6, 2e-3 * gbad.size), (Hm, Wm, H, W, int(gbad.sum()))
