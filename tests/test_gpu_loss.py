"""HIP fused L1 + SSIM (adgs.loss) through the C ABI against the reference's golden vectors and the NumPy oracle."""
import os

import numpy as np
import pytest
import torch

from oracle import loss_oracle

pytestmark = pytest.mark.gpu
GOLD = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "loss_golden.npz"))
CASES = sorted({k.split("/")[0] for k in GOLD.files if not k.startswith("depth")})
DEPTH_CASES = sorted({k.split("/")[0] for k in GOLD.files if k.startswith("depth")})


@pytest.mark.parametrize("case", CASES)
def test_l1_ssim_matches_reference_golden(case):
    from adgs import loss
    img = torch.tensor(GOLD[case + "/img"]).cuda().requires_grad_(True)
    gt = torch.tensor(GOLD[case + "/gt"]).cuda()
    l1, s = loss.l1_ssim(img, gt)
    assert abs(float(l1) - float(GOLD[case + "/l1"])) <= 1e-6 and abs(float(s) - float(GOLD[case + "/ssim"])) <= 1e-5
    (g1,) = torch.autograd.grad(l1, img, retain_graph=True)
    (g2,) = torch.autograd.grad(s, img)
    np.testing.assert_allclose(g1.cpu().numpy(), GOLD[case + "/g_l1"], rtol=1e-6, atol=1e-9)
    ref = GOLD[case + "/g_ssim"]
    np.testing.assert_allclose(g2.cpu().numpy(), ref, rtol=0, atol=1e-4 * np.abs(ref).max())


def test_reference_named_wrappers_and_combined_loss():
    from adgs import loss
    rng = np.random.default_rng(3)
    gt_np = rng.random((3, 70, 131)).astype(np.float32)
    img_np = np.clip(gt_np + 0.2 * rng.standard_normal(gt_np.shape), 0, 1).astype(np.float32)
    img = torch.tensor(img_np).cuda().requires_grad_(True); gt = torch.tensor(gt_np).cuda()
    lam = 0.2
    total, l1, dssim = loss.photometric_loss(img, gt, lam)
    total.backward()
    o_l1, o_s, og_l1, og_s = loss_oracle.l1_ssim(img_np, gt_np)
    assert abs(float(l1) - o_l1) <= 1e-6 and abs(float(dssim) - (1 - o_s)) <= 1e-5
    want = (1 - lam) * og_l1 - lam * og_s
    np.testing.assert_allclose(img.grad.cpu().numpy(), want, rtol=0, atol=1e-4 * np.abs(want).max())
    assert abs(float(loss.l1_loss(img, gt)) - o_l1) <= 1e-6 and abs(float(loss.ssim(img, gt)) - o_s) <= 1e-5
    with pytest.raises(RuntimeError):
        loss.l1_ssim(torch.zeros(3, 8, 8), torch.zeros(3, 8, 8))          # no CPU path


def test_full_resolution_properties():
    """1920x1280: identical images give L1 = 0, SSIM = 1; the SSIM gradient of a constant offset sums to ~0 per the
    symmetry of the window (size-independent checks at BASELINE.json's resolution)."""
    from adgs import loss
    g = torch.Generator().manual_seed(0)
    gt = torch.rand(3, 1280, 1920, generator=g).cuda()
    l1, s = loss.l1_ssim(gt.clone().requires_grad_(True), gt)
    assert float(l1) == 0.0 and abs(float(s) - 1.0) < 1e-6
    img = (gt + 0.05 * torch.randn(3, 1280, 1920, generator=g).cuda()).requires_grad_(True)
    l1, s = loss.l1_ssim(img, gt)
    (l1 + s).backward()
    assert 0.03 < float(l1) < 0.05 and 0.0 < float(s) < 1.0 and torch.isfinite(img.grad).all()
    # linearity of the backward in the upstream gradients
    img2 = img.detach().clone().requires_grad_(True)
    l1b, sb = loss.l1_ssim(img2, gt)
    (2.0 * l1b + 3.0 * sb).backward()
    ga = torch.autograd.grad(loss.l1_ssim(img, gt)[0], img)[0]
    assert torch.allclose(img2.grad, 3.0 * img.grad - ga, rtol=1e-4, atol=1e-10)


@pytest.mark.parametrize("case", DEPTH_CASES)
def test_depth_loss_matches_reference_golden(case):
    from adgs import loss
    pred = torch.tensor(GOLD[case + "/pred"]).cuda().requires_grad_(True)
    gt = torch.tensor(GOLD[case + "/gt"]).cuda()
    mask = torch.tensor(GOLD[case + "/mask"]).cuda() if GOLD[case + "/mask"].size else None
    val = loss.get_depth_loss(pred, gt, mask)
    ref = float(GOLD[case + "/loss"])
    assert abs(float(val) - ref) <= 2e-5 * max(1.0, abs(ref))
    (gp,) = torch.autograd.grad(val, pred)
    gref = GOLD[case + "/g_pred"]
    np.testing.assert_allclose(gp.cpu().numpy(), gref, rtol=0, atol=2e-4 * max(np.abs(gref).max(), 1e-12))


def test_depth_loss_full_resolution_against_oracle():
    from adgs import loss
    rng = np.random.default_rng(5)
    gt = (rng.random((1280, 1920)) * 60 + 1).astype(np.float32)
    pred = (0.02 * gt + 0.3 + 0.1 * rng.standard_normal(gt.shape)).astype(np.float32)
    mask = (rng.random(gt.shape) > 0.25).astype(np.float32)
    p = torch.tensor(pred).cuda().requires_grad_(True)
    val = loss.get_depth_loss(p, torch.tensor(gt).cuda(), torch.tensor(mask).cuda())
    val.backward()
    o_loss, o_grad = loss_oracle.depth_loss(pred, gt, mask)
    assert abs(float(val) - o_loss) <= 1e-4 * o_loss
    np.testing.assert_allclose(p.grad.cpu().numpy(), o_grad, rtol=0, atol=1e-3 * np.abs(o_grad).max())


GOLD2 = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "loss2_golden.npz"))


@pytest.mark.parametrize("name", ["flow_24x40", "flow_noopacity_17x23", "flow_none_selected_8x8"])
def test_flow_loss_vs_reference_golden(name):
    from adgs.loss import get_flow_loss
    g = lambda k: GOLD2[name + "/" + k]
    d = lambda a: torch.tensor(a, device="cuda")
    f = d(g("img_flow")).requires_grad_(True)
    op = d(g("opacity")).requires_grad_(True) if g("opacity").size else None
    loss = get_flow_loss(f, (None, d(g("K")), d(g("R")), d(g("T")), d(g("flow")), d(g("vis"))), op, dist=0.02)
    (loss * 1.0).backward()
    np.testing.assert_allclose(float(loss), float(g("loss")), rtol=1e-5, atol=1e-8)
    scale = max(np.abs(g("g_img_flow")).max(), 1e-30)
    np.testing.assert_allclose(f.grad.cpu().numpy(), g("g_img_flow"), rtol=1e-4, atol=1e-4 * scale)
    if op is not None:
        np.testing.assert_allclose(op.grad.cpu().numpy(), g("g_opacity"), rtol=1e-4, atol=1e-4 * max(np.abs(g("g_opacity")).max(), 1e-30))


def test_flow_loss_full_size_vs_oracle_and_bce_terms():
    from adgs.loss import get_flow_loss, obj_loss, sky_loss
    from oracle import loss_oracle as lo
    rng = np.random.default_rng(5)
    H, W = 320, 480
    K = np.array([[400.0, 0, W / 2], [0, 410.0, H / 2], [0, 0, 1]], np.float32); R = np.eye(3, dtype=np.float32); T = np.array([0.2, 0.0, 0.1], np.float32)
    pts = (rng.normal(size=(3, H, W)) * [[[3.0]], [[2.0]], [[6.0]]] + [[[0.0]], [[0.0]], [[8.0]]]).astype(np.float32)
    flow = np.stack([rng.random((H, W)) * (W + 20) - 10, rng.random((H, W)) * (H + 20) - 10]).astype(np.float32)
    vis, op = rng.random((H, W)).astype(np.float32), rng.random((H, W)).astype(np.float32)
    d = lambda a: torch.tensor(a, device="cuda")
    f, o = d(pts).requires_grad_(True), d(op).requires_grad_(True)
    loss = get_flow_loss(f, (None, d(K), d(R), d(T), d(flow), d(vis)), o, dist=0.02)
    (loss * 3.0).backward()
    want, g_f, g_o = lo.flow_loss(pts, flow, vis, op, K, R, T, 0.02)
    np.testing.assert_allclose(float(loss), want, rtol=1e-5)
    np.testing.assert_allclose(f.grad.cpu().numpy(), 3.0 * g_f, rtol=1e-4, atol=1e-4 * np.abs(g_f).max() * 3)
    np.testing.assert_allclose(o.grad.cpu().numpy(), 3.0 * g_o, rtol=1e-4, atol=1e-4 * np.abs(g_o).max() * 3)
    # the two clipped BCE terms of train.py:95-103 vs golden values from the torch expression
    g = lambda k: GOLD2["bce_19x31/" + k]
    p1 = d(g("pred")).requires_grad_(True)
    l1 = obj_loss(p1[None], d(g("gt_sem"))); l1.backward()
    np.testing.assert_allclose(float(l1), float(g("obj")), rtol=1e-5); np.testing.assert_allclose(p1.grad.cpu().numpy(), g("g_obj"), rtol=1e-4, atol=1e-8)
    p2 = d(g("pred")).requires_grad_(True)
    l2 = sky_loss(p2, d(g("gt_sky"))); l2.backward()
    np.testing.assert_allclose(float(l2), float(g("sky")), rtol=1e-5); np.testing.assert_allclose(p2.grad.cpu().numpy(), g("g_sky"), rtol=1e-4, atol=1e-8)
    with pytest.raises(RuntimeError):
        sky_loss(torch.zeros(4, 4), torch.zeros(4, 4))


def test_flow_camera_reallocated_every_iteration_is_never_stale():
    """train.py:70 `flow_pkg = [a.cuda() ...]`: with the dataset on the CPU every iteration hands in FRESH GPU tensors for K / R / T,
    and the caching allocator re-issues the same few addresses (version counter 0, same shape).  Each call must use the camera it was
    given -- device camera and host camera agree bit for bit, forward and backward, standalone term and the fused image_losses node."""
    from adgs import loss
    from oracle import loss_oracle as lo
    rng = np.random.default_rng(11)
    H, W = 40, 56
    pts = (rng.normal(size=(3, H, W)) * [[[3.0]], [[2.0]], [[6.0]]] + [[[0.0]], [[0.0]], [[8.0]]]).astype(np.float32)
    flow = np.stack([rng.random((H, W)) * (W - 1), rng.random((H, W)) * (H - 1)]).astype(np.float32)
    vis, op = (rng.random((H, W)) * 0.5 + 0.5).astype(np.float32), rng.random((H, W)).astype(np.float32)
    d = lambda a: torch.tensor(a, device="cuda")
    seen = []

    def iteration(it):      # a function: every temporary is gone when it returns, only the allocator's free lists remember it
        K = np.array([[300.0 + 37 * it, 0, W / 2], [0, 310.0 - 11 * it, H / 2], [0, 0, 1]], np.float32)
        a = 0.1 * it
        R = np.array([[np.cos(a), -np.sin(a), 0], [np.sin(a), np.cos(a), 0], [0, 0, 1]], np.float32)
        T = np.array([0.2 - 0.1 * it, 0.05 * it, 0.1], np.float32)
        Kd, Rd, Td = d(K), d(R), d(T)                         # fresh allocations, freed at the end of the iteration
        seen.append(Kd.data_ptr())
        want, g_f, g_o = lo.flow_loss(pts, flow, vis, op, K, R, T, 0.02)
        res = []
        for cam in ((Kd, Rd, Td), (torch.tensor(K), torch.tensor(R), torch.tensor(T))):
            f, o = d(pts).requires_grad_(True), d(op).requires_grad_(True)
            l = loss.get_flow_loss(f, (None,) + cam + (d(flow), d(vis)), o, dist=0.02)
            l.backward()
            np.testing.assert_allclose(float(l), want, rtol=1e-5, err_msg="iteration %d" % it)
            np.testing.assert_allclose(f.grad.cpu().numpy(), g_f, rtol=1e-4, atol=1e-4 * np.abs(g_f).max())
            np.testing.assert_allclose(o.grad.cpu().numpy(), g_o, rtol=1e-4, atol=1e-4 * np.abs(g_o).max())
            res.append((l.detach().clone(), f.grad.clone(), o.grad.clone()))
        for x, y in zip(*res):
            assert torch.equal(x, y)                           # device camera == host camera, bit for bit
        f = d(pts).requires_grad_(True)
        z = lambda *s: torch.rand(*s, device="cuda")
        terms = loss.image_losses(z(3, H, W), z(3, H, W), z(H, W), z(H, W), f, (None, Kd, Rd, Td, d(flow), d(vis)), d(op), z(1, H, W), z(H, W), z(H, W), dist=0.02)
        assert torch.equal(terms[3], res[0][0])
        terms[3].backward()
        assert torch.equal(f.grad, res[0][1])

    # at least six cameras, and on until the allocator has re-issued an address at least twice (which address a fresh tensor gets depends on
    # what the tests before this one left in the allocator's free lists: with six iterations flat the suite once saw six different ones)
    it = 0
    while it < 6 or (len(seen) - len(set(seen)) < 2 and it < 24):
        iteration(it)
        torch.cuda.synchronize()
        it += 1
    if len(set(seen)) == len(seen):
        pytest.skip("every camera was correct, but the allocator re-issued no address in %d iterations: the stale-cache case was not exercised" % it)
    assert not hasattr(loss, "_HOST_FLOATS")


@pytest.mark.parametrize("seed", range(int(os.environ.get("ADGS_TEST_SEED_BASE", "0")), int(os.environ.get("ADGS_TEST_SEED_BASE", "0")) + int(os.environ.get("ADGS_TEST_LOSS_SEEDS", "12"))))
def test_random_image_shapes_vs_oracle(seed):
    """L1 + SSIM, the depth loss and the flow loss on random image shapes (smaller than the 11x11 window, one row / one column,
    sizes that are no multiple of any block shape) against the NumPy oracle: values and gradients."""
    from adgs import loss
    rng = np.random.default_rng(9000 + seed)
    H = int(rng.choice([1, 2, 5, 11, 12, 16, 17, 31, 33, 64, 100, 131]))
    W = int(rng.choice([1, 3, 10, 11, 16, 31, 32, 33, 63, 65, 128, 257]))
    gt_np = rng.random((3, H, W)).astype(np.float32)
    img_np = np.clip(gt_np + float(rng.choice([0.0, 0.02, 0.3])) * rng.standard_normal(gt_np.shape), 0, 1).astype(np.float32)
    img = torch.tensor(img_np).cuda().requires_grad_(True)
    l1, s = loss.l1_ssim(img, torch.tensor(gt_np).cuda())
    a, b = float(rng.uniform(0.5, 2)), float(rng.uniform(-2, 2))
    (a * l1 + b * s).backward()
    o_l1, o_s, og_l1, og_s = loss_oracle.l1_ssim(img_np, gt_np)
    assert abs(float(l1) - o_l1) <= 1e-6 and abs(float(s) - o_s) <= 1e-5, (H, W)
    want = a * og_l1 + b * og_s
    # identical images: the exact gradient is 0 and fp32 leaves rounding noise -> the floor is the scale of a mean's gradient
    np.testing.assert_allclose(img.grad.cpu().numpy(), want, rtol=0, atol=1e-4 * max(np.abs(want).max(), 1.0 / gt_np.size), err_msg=str((H, W)))
    # depth loss: scale/shift-invariant fit over a random mask (incl. the empty mask and the degenerate single-pixel fit)
    gtd = (rng.random((H, W)) * 60 + 1).astype(np.float32)
    pred = (0.02 * gtd + 0.3 + 0.1 * rng.standard_normal(gtd.shape)).astype(np.float32)
    mask = (rng.random(gtd.shape) > float(rng.choice([0.0, 0.3, 0.9]))).astype(np.float32)
    mask[0, 0] = 1.0                                   # an empty mask is 0/0 in the reference as well
    use_mask = bool(rng.random() < 0.8)
    p = torch.tensor(pred).cuda().requires_grad_(True)
    val = loss.get_depth_loss(p, torch.tensor(gtd).cuda(), torch.tensor(mask).cuda() if use_mask else None)
    val.backward()
    if (mask.sum() if use_mask else H * W) < 8:         # (near-)rank-deficient fits: conditioning, not parity -- only "does not trap"
        return
    o_loss, o_grad = loss_oracle.depth_loss(pred, gtd, mask if use_mask else None)
    assert np.isfinite(o_loss) and torch.isfinite(val) and torch.isfinite(p.grad).all(), (H, W)
    assert abs(float(val) - o_loss) <= 2e-4 * max(abs(o_loss), 1e-3), (H, W, float(val), o_loss)
    # |s p + t - g| has a kink: a residual within rounding of zero takes the other sign in fp32 (seen once in 300 seeds)
    bad = np.abs(p.grad.cpu().numpy() - o_grad) > 2e-3 * max(np.abs(o_grad).max(), 1e-9)
    assert bad.sum() <= max(2, 1e-4 * bad.size), (H, W, int(bad.sum()))


GOLD3 = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "loss3_golden.npz"))
REG_CASES = sorted({k.split("/")[0] for k in GOLD3.files})


@pytest.mark.parametrize("case", REG_CASES)
def test_regularisers_match_the_train_py_expressions(case):
    """reg_loss, sigma_loss, reg_sigma_loss (train.py:104-113): values and gradients against the golden vectors generated from
    the reference's expressions (tests/golden/make_loss_golden.py)."""
    from adgs import loss
    x = torch.tensor(GOLD3[case + "/xyz_deform_param"]).cuda().requires_grad_(True)
    s = torch.tensor(GOLD3[case + "/gs_time_sigma"]).cuda().requires_grad_(True)
    idx = torch.tensor(GOLD3[case + "/obj_near_idx"]).cuda()
    gap = float(GOLD3[case + "/frame_gap"])
    for fn, args, wrt, key, gkey in ((loss.reg_loss, (x, idx), x, "reg_loss", "g_reg"), (loss.reg_sigma_loss, (s, idx), s, "reg_sigma_loss", "g_reg_sigma"),
                                     (loss.sigma_loss, (s, gap), s, "sigma_loss", "g_sigma")):
        val = fn(*args)
        ref = float(GOLD3[case + "/" + key])
        assert abs(float(val) - ref) <= 3e-6 * max(1.0, abs(ref)), (key, float(val), ref)
        (g,) = torch.autograd.grad(2.5 * val, wrt)
        ref_g = GOLD3[case + "/" + gkey]
        np.testing.assert_allclose(g.cpu().numpy() / 2.5, ref_g, rtol=3e-5, atol=3e-6 * np.abs(ref_g).max())


def test_regularisers_at_training_size_vs_oracle():
    """C3's object range: 200 k Gaussians, K = 8 neighbours, 18 position-deformation parameters per axis."""
    from adgs import loss
    rng = np.random.default_rng(9)
    No, K, C = 200_000, 8, 18
    x_np = (0.05 * rng.standard_normal((No, 3, C))).astype(np.float32)
    s_np = (np.log(0.02) + 0.3 * rng.standard_normal((No, 2))).astype(np.float32)
    idx_np = rng.integers(0, No, size=(No // K, K), dtype=np.int64)
    x = torch.tensor(x_np).cuda().requires_grad_(True); s = torch.tensor(s_np).cuda().requires_grad_(True); idx = torch.tensor(idx_np).cuda()
    total = 0.5 * loss.reg_loss(x, idx) + 0.5 * loss.reg_sigma_loss(s, idx) + 0.01 * loss.sigma_loss(s, 0.02)
    total.backward()
    l1, g1 = loss_oracle.group_var_loss(x_np, idx_np)
    l2, g2 = loss_oracle.group_var_loss(s_np, idx_np)
    l3, g3 = loss_oracle.sigma_loss(s_np, 0.02)
    want = 0.5 * l1 + 0.5 * l2 + 0.01 * l3
    assert abs(float(total) - want) <= 2e-6 * abs(want)
    np.testing.assert_allclose(x.grad.cpu().numpy(), 0.5 * g1, rtol=2e-4, atol=1e-6 * np.abs(g1).max())
    np.testing.assert_allclose(s.grad.cpu().numpy(), 0.5 * g2 + 0.01 * g3, rtol=2e-4, atol=1e-6 * np.abs(g2).max())


def test_regulariser_argument_errors():
    from adgs import loss
    x = torch.zeros(8, 3, 4, device="cuda")
    with pytest.raises(ValueError):
        loss.reg_loss(x, torch.zeros(2, 3, dtype=torch.int32, device="cuda"))
    with pytest.raises(RuntimeError):
        loss.reg_loss(x, torch.zeros(2, 1, dtype=torch.int64, device="cuda"))       # K = 1: torch.var is undefined there
    with pytest.raises(ValueError):
        loss.sigma_loss(torch.zeros(8, 3, device="cuda"), 0.1)


def test_regulariser_indices_follow_fancy_indexing_and_never_leave_the_tensor():
    """`param[obj_near_idx]` in the reference (train.py:104-113): a negative index counts from the end; an index outside [-N, N) is an
    IndexError there.  Here (round-3 advisor finding: the kernels indexed unchecked) -1 means the last row, and an out-of-range index
    -- a stale obj_near_idx from before a prune -- gives a NaN loss instead of an out-of-bounds read / atomic, with no gradient written
    for that row."""
    from adgs import loss
    g = torch.Generator().manual_seed(3)
    x = torch.randn(50, 3, 4, generator=g).cuda().requires_grad_(True)
    idx = torch.randint(0, 50, (20, 8), generator=g).cuda()
    neg = idx.clone(); neg[3, 2] = -1; neg[7, 0] = -50
    pos = idx.clone(); pos[3, 2] = 49; pos[7, 0] = 0
    a = loss.reg_loss(x, neg); ga, = torch.autograd.grad(a, x)
    b = loss.reg_loss(x, pos); gb, = torch.autograd.grad(b, x)
    assert torch.equal(a, b) and torch.allclose(ga, gb, rtol=1e-5, atol=1e-7)       # the gradient rows are float atomics: same sums, any order
    ref = x[pos].var(dim=1).sum(-1).mean()
    assert abs(float(a) - float(ref)) <= 1e-5 * abs(float(ref))
    for bad_value in (50, -51, 10 ** 12):
        bad = idx.clone(); bad[5, 1] = bad_value
        with pytest.raises(IndexError):                        # the wrapper validates every new index tensor once, like the reference's indexing
            loss.reg_loss(x, bad)
        # ... and below the wrapper (stream capture, other callers of the C ABI) the kernels guard themselves: NaN loss, nothing out of
        # bounds, and NO gradient for the rows of the affected group -- a NaN there would poison the parameters and both Adam moments
        import ctypes
        from adgs import _lib
        guard = torch.full((64,), 7.0, device="cuda")          # memory next to the gradient: must stay untouched
        xs = x.detach().contiguous()
        work = torch.zeros(loss.AUX_WORK_DOUBLES, dtype=torch.float64, device="cuda")
        out, gl, dx = torch.empty(1, device="cuda"), torch.ones(1, device="cuda"), torch.zeros_like(xs)
        st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
        _lib.check(_lib.lib().adgs_group_var_forward(50, 20, 8, 12, 4, xs.data_ptr(), bad.data_ptr(), work.data_ptr(), out.data_ptr(), st), "fwd")
        _lib.check(_lib.lib().adgs_group_var_backward(50, 20, 8, 12, 4, xs.data_ptr(), bad.data_ptr(), gl.data_ptr(), dx.data_ptr(), st), "bwd")
        torch.cuda.synchronize()
        assert torch.isnan(out[0]) and bool((guard == 7.0).all()) and bool(torch.isfinite(dx).all())
        ok = idx.clone(); ok[5] = ok[4]                        # the same index with group 5 replaced: every other group's gradient is unchanged
        dx_ok = torch.zeros_like(xs)
        _lib.check(_lib.lib().adgs_group_var_backward(50, 20, 8, 12, 4, xs.data_ptr(), ok.data_ptr(), gl.data_ptr(), dx_ok.data_ptr(), st), "bwd")
        only4 = idx[4:5].repeat(20, 1); dx4 = torch.zeros_like(xs)
        _lib.check(_lib.lib().adgs_group_var_backward(50, 20, 8, 12, 4, xs.data_ptr(), only4.data_ptr(), gl.data_ptr(), dx4.data_ptr(), st), "bwd")
        torch.cuda.synchronize()
        assert torch.allclose(dx, dx_ok - dx4 / 20.0, rtol=1e-4, atol=1e-6)
    loss.reg_loss(x, idx)                                      # a valid tensor is validated once ...
    idx[0, 0] = 77                                             # ... and again after an in-place change (version counter)
    with pytest.raises(IndexError):
        loss.reg_loss(x, idx)


def test_work_arena_survives_many_outstanding_terms_and_failed_forwards():
    """The zero-on-return work buffers of the loss kernels (adgs.loss._WorkArena): (1) more loss terms between a forward and its backward
    than the ring holds (gradient accumulation) fall back to fresh buffers instead of raising, and every backward still reads its own
    totals; (2) a slice whose forward died between the sum kernel and the finish kernel (token dropped before done()) is zero-filled
    before it is handed out again -- otherwise every later term that got the slice would be silently biased."""
    from adgs import loss
    g = torch.Generator().manual_seed(7)
    H, W = 24, 40
    preds = [torch.rand(H, W, generator=g).cuda().requires_grad_(True) for _ in range(loss._WorkArena.N + 12)]
    gts = [torch.rand(H, W, generator=g).cuda() for _ in preds]
    terms = [loss.get_depth_loss(p, t) for p, t in zip(preds, gts)]          # 76 forwards, no backward yet: 76 live tokens
    refs = []
    for p, t in zip(preds, gts):
        q = p.detach().clone().requires_grad_(True)
        l = loss.get_depth_loss(q, t); l.backward()
        refs.append((l.detach(), q.grad))
    for (l, p), (rl, rg) in zip(zip(terms, preds), refs):
        l.backward()
        assert torch.equal(l.detach(), rl) and torch.equal(p.grad, rg)
    # (2) spoil a slice by hand
    dev = preds[0].device
    buf, tok = loss._work(dev, loss.AUX_WORK_DOUBLES)
    arena, i = tok.arena, tok.i
    buf.fill_(123.0)                      # what a sum kernel without its finish kernel leaves behind
    del tok                               # dropped without done(): the forward "raised"
    assert i in arena.dirty and not arena.busy[i]
    arena.next = i                        # the very next term gets this slice
    p = torch.rand(H, W, generator=g).cuda()
    t = (torch.rand(H, W, generator=g) > 0.5).float().cuda()
    a = loss.sky_loss(p, t)
    arena.next = (i + 5) % arena.N
    b = loss.sky_loss(p, t)
    assert torch.equal(a, b) and i not in arena.dirty
    assert float(arena.buf[i][:2 * 256].abs().max()) == 0.0          # the slot rows (the two scalars behind them hold the last term's totals)


@pytest.mark.parametrize("H,W,D_S", [(97, 131, 1), (64, 80, 3)])
def test_image_losses_node_equals_the_six_functions(H, W, D_S):
    """adgs.loss.image_losses (one autograd node for train.py:78-99) against l1_ssim / get_depth_loss / get_flow_loss / obj_loss / sky_loss
    on the same inputs: the same kernels, so the same values bit for bit; the gradients of every input equal (img_opacity receives the
    sum of the flow and the sky term either way)."""
    from adgs import loss
    g = torch.Generator().manual_seed(H * 1000 + W)
    dev = "cuda"
    r = lambda *s: torch.rand(*s, generator=g)
    gt_img, gt_depth, gt_sem, gt_sky = r(3, H, W).to(dev), (r(H, W) * 0.5 + 0.01).to(dev), (r(H, W) > 0.8).float().to(dev), (r(H, W) > 0.7).float().to(dev)
    K = torch.tensor([[90.0, 0.0, W / 2.0], [0.0, 90.0, H / 2.0], [0.0, 0.0, 1.0]])
    R, T = torch.eye(3), torch.tensor([0.05, -0.02, 0.1])
    flow_pkg = (0.4, K, R, T, torch.stack([r(H, W) * (W - 1), r(H, W) * (H - 1)]).to(dev), (r(H, W) > 0.3).float().to(dev))
    base = dict(image=r(3, H, W), depth=r(H, W) * 0.4 + 0.05, img_flow=torch.cat([r(2, H, W) * 4 - 2, r(1, H, W) * 5 + 1]), img_opacity=r(H, W) * 0.98 + 0.01,
                img_semantic=r(D_S, H, W))
    w = torch.tensor([0.8, 0.2, 0.1, 0.1, 0.1, 0.05], device=dev)

    def run(fused):
        x = {k: v.clone().to(dev).requires_grad_(True) for k, v in base.items()}
        if fused:
            terms = loss.image_losses(x["image"], gt_img, x["depth"], gt_depth, x["img_flow"], flow_pkg, x["img_opacity"], x["img_semantic"], gt_sem, gt_sky, dist=0.02)
        else:
            l1, s = loss.l1_ssim(x["image"], gt_img)
            terms = (l1, s, loss.get_depth_loss(x["depth"], gt_depth), loss.get_flow_loss(x["img_flow"], flow_pkg, x["img_opacity"], dist=0.02),
                     loss.obj_loss(x["img_semantic"], gt_sem), loss.sky_loss(x["img_opacity"], gt_sky))
        total = (torch.stack([t.reshape(()) for t in terms]) * w).sum()
        total.backward()
        return [t.detach().clone() for t in terms], {k: v.grad.detach().clone() for k, v in x.items()}

    ta, ga = run(False)
    tb, gb = run(True)
    for a, b in zip(ta, tb):
        assert torch.equal(a, b), (a, b)
    for k in ga:
        assert gb[k].shape == ga[k].shape, k
        assert torch.allclose(gb[k], ga[k], rtol=1e-6, atol=1e-9 + 1e-6 * float(ga[k].abs().max())), k
