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


def _env_cam(yaw, pitch, cid, W=640, H=400):
    cy, sy, cp, sp = np.cos(yaw), np.sin(yaw), np.cos(pitch), np.sin(pitch)
    R = np.array([[0.0, 0.0, 1.0], [-1.0, 0.0, 0.0], [0.0, -1.0, 0.0]]) @ np.array([[cy, 0, sy], [0, 1, 0], [-sy, 0, cy]]) @ np.array([[1, 0, 0], [0, cp, -sp], [0, sp, cp]])
    w2v = torch.eye(4); w2v[:3, :3] = torch.tensor(R, dtype=torch.float32)
    return types.SimpleNamespace(FoVx=0.9, image_width=W, image_height=H, world_view_transform=w2v.cuda(), cam_id=cid)


@pytest.mark.parametrize("fused_zero", [True, False])
def test_sparse_gradient_protocol_equals_the_dense_path(fused_zero):
    """EnvironmentMap(sparse_grad=True): the backward marks the optimizer's tiles and reuses the buffer the optimizer zeroes.  Over 9
    steps and three cameras (one at the azimuth seam, whose workgroups take the direct-atomics path) the map and its moments equal,
    bit for bit, those of a plain dense FusedAdam fed the SAME gradient tensors (the float atomics of two backward runs differ in
    the last bit, and Adam with eps = 1e-15 is discontinuous at g = 0: the two paths are compared on identical gradients);
    every tile holding a non-zero gradient is marked; a gradient that arrives summed with a second term (a new tensor) falls back
    to the dense handling by itself."""
    from adgs import env
    from adgs.optim import FusedAdam, ADAM_TILE
    e = env.EnvironmentMap(1024, 3, sparse_grad=True)
    with torch.no_grad():
        e.grid_map.copy_(torch.randn(e.grid_map.shape, generator=torch.Generator().manual_seed(1)).cuda() * 0.3)
    e.training_setup(types.SimpleNamespace(env_lr=1e-2))
    ref = torch.nn.Parameter(e.grid_map.detach().clone())
    ref_opt = FusedAdam([{"params": [ref], "lr": 1e-2}], lr=0.0, eps=1e-15)
    cams = [_env_cam(0.3, 0.05, 0), _env_cam(-0.2, -0.1, 1), _env_cam(np.pi / 2 + 1.57, 0.02, 2)]
    g = torch.Generator().manual_seed(5)
    ws = [torch.randn(3, 400, 640, generator=g).cuda() for _ in range(3)]
    adopted = 0
    for it in range(9):
        bg = e.get_image_background(cams[it % 3])
        loss = (bg * ws[it % 3]).sum()
        if it == 6:
            loss = loss + (e.grid_map ** 2).sum() * 1e-3          # a second gradient source: autograd hands over gg + other
        loss.backward()
        mg = e.optimizer.marked_gradient(e.grid_map)
        own = mg.is_live(e.grid_map.grad)
        assert own == (it != 6), it
        adopted += own
        ref.grad = e.grid_map.grad.detach().clone()
        if own:          # every tile that holds a gradient is marked
            nz = (ref.grad.reshape(-1, ADAM_TILE) != 0).any(1)
            assert bool(mg.marks[nz].all()), it
        if fused_zero:
            e.optimizer.step(zero_grad=True)
            buf = mg.buffer
            assert (buf is not None) == own
            if buf is not None:
                assert float(buf.abs().max()) == 0.0      # handed back, all zero
            del buf          # a second reference would make autograd clone the next gradient instead of adopting it (-> dense handling)
        else:
            e.optimizer.step()
            e.optimizer.zero_grad(set_to_none=True)
        ref_opt.step(zero_grad=True)
        st, rst = e.optimizer.state[e.grid_map], ref_opt.state[ref]
        assert torch.equal(e.grid_map.detach(), ref.detach()), it
        assert torch.equal(st["exp_avg"], rst["exp_avg"]) and torch.equal(st["exp_avg_sq"], rst["exp_avg_sq"]), it
    assert adopted == 8 and float(st["exp_avg"].abs().max()) > 0


def test_in_place_accumulation_into_the_marked_buffer_is_handled_densely():
    """Round-3 advisor finding: with grad mode off autograd accumulates IN PLACE (`p.grad += new` for a second backward before the
    step), so a non-marking source writes into unmarked tiles of the producer's buffer while its address still matches.  The optimizer
    must see that (version counter) and scan the gradient densely: the regulariser's share reaches every texel, the step equals a plain
    dense FusedAdam on the same gradient, and the buffer that comes back later is all zero (no residue in recycled tiles)."""
    from adgs import env
    from adgs.optim import FusedAdam
    e = env.EnvironmentMap(512, 3, sparse_grad=True)
    with torch.no_grad():
        e.grid_map.copy_(torch.randn(e.grid_map.shape, generator=torch.Generator().manual_seed(2)).cuda() * 0.3)
    e.training_setup(types.SimpleNamespace(env_lr=1e-2))
    ref = torch.nn.Parameter(e.grid_map.detach().clone())
    ref_opt = FusedAdam([{"params": [ref], "lr": 1e-2}], lr=0.0, eps=1e-15)
    cam = _env_cam(0.3, 0.05, 0)
    w = torch.randn(3, 400, 640, generator=torch.Generator().manual_seed(6)).cuda()
    mg = e.optimizer.marked_gradient(e.grid_map)
    for it in range(4):
        (e.get_image_background(cam) * w).sum().backward()
        ptr = e.grid_map.grad.data_ptr()
        assert mg.is_live(e.grid_map.grad)
        if it == 1:
            ((e.grid_map ** 2).sum() * 1e-3).backward()          # a second backward: AccumulateGrad adds into p.grad in place
            assert e.grid_map.grad.data_ptr() == ptr              # same address ...
            assert not mg.is_live(e.grid_map.grad)                # ... but no longer the buffer that was issued
            assert float((e.grid_map.grad != 0).float().mean()) > 0.99      # the regulariser reaches texels no camera ray marks
        ref.grad = e.grid_map.grad.detach().clone()
        e.optimizer.step(zero_grad=True)
        ref_opt.step(zero_grad=True)
        assert torch.equal(e.grid_map.detach(), ref.detach()), it
        buf = mg.buffer
        assert (buf is None) == (it == 1)                          # the touched buffer is not recycled
        if buf is not None:
            assert float(buf.abs().max()) == 0.0
        del buf
