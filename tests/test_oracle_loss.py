"""The photometric-loss oracle (oracle/loss_oracle.py) against golden vectors produced by the reference's own
utils/loss_utils.py (tests/golden/make_loss_golden.py): values and autograd gradients."""
import os

import numpy as np
import pytest

from oracle import loss_oracle

GOLD = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "loss_golden.npz"))
CASES = sorted({k.split("/")[0] for k in GOLD.files if not k.startswith("depth")})
DEPTH_CASES = sorted({k.split("/")[0] for k in GOLD.files if k.startswith("depth")})


def test_window_is_the_references():
    g = loss_oracle.gaussian_window()
    assert g.dtype == np.float32 and g.shape == (11,) and abs(float(g.sum()) - 1.0) < 1e-6
    assert np.allclose(g, g[::-1]) and int(np.argmax(g)) == 5


@pytest.mark.parametrize("case", CASES)
@pytest.mark.parametrize("dtype", [np.float64, np.float32])
def test_oracle_matches_reference_values_and_gradients(case, dtype):
    img, gt = GOLD[case + "/img"], GOLD[case + "/gt"]
    l1, s, g_l1, g_s = loss_oracle.l1_ssim(img, gt, dtype=dtype)
    tol = 2e-6 if dtype is np.float64 else 1e-5
    assert abs(l1 - float(GOLD[case + "/l1"])) <= tol and abs(s - float(GOLD[case + "/ssim"])) <= tol
    np.testing.assert_allclose(g_l1, GOLD[case + "/g_l1"], rtol=1e-6, atol=1e-9)
    ref = GOLD[case + "/g_ssim"]
    np.testing.assert_allclose(g_s, ref, rtol=0, atol=(3e-5 if dtype is np.float64 else 1e-4) * np.abs(ref).max())


def test_identical_images_and_finite_difference():
    rng = np.random.default_rng(0)
    x = rng.random((3, 20, 23))
    l1, s, _, g = loss_oracle.l1_ssim(x, x)
    assert l1 == 0.0 and abs(s - 1.0) < 1e-12
    y = np.clip(x + 0.1 * rng.standard_normal(x.shape), 0, 1)
    _, s0, _, g = loss_oracle.l1_ssim(y, x)
    for idx in [(0, 0, 0), (1, 10, 11), (2, 19, 22), (0, 3, 21)]:
        d = np.zeros_like(y); d[idx] = 1e-6
        fd = (loss_oracle.l1_ssim(y + d, x)[1] - loss_oracle.l1_ssim(y - d, x)[1]) / 2e-6
        assert abs(fd - g[idx]) <= 1e-6 * max(1.0, abs(g[idx]) * 1e3), (idx, fd, g[idx])


@pytest.mark.parametrize("case", DEPTH_CASES)
def test_depth_loss_oracle_matches_reference(case):
    loss, grad = loss_oracle.depth_loss(GOLD[case + "/pred"], GOLD[case + "/gt"], GOLD[case + "/mask"])
    ref = float(GOLD[case + "/loss"])
    assert abs(loss - ref) <= 2e-5 * max(1.0, abs(ref)), (loss, ref)
    gref = GOLD[case + "/g_pred"]
    np.testing.assert_allclose(grad, gref, rtol=0, atol=2e-4 * max(np.abs(gref).max(), 1e-12))


lo = loss_oracle
GOLD2 = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "loss2_golden.npz"))


@pytest.mark.parametrize("name", ["flow_24x40", "flow_noopacity_17x23", "flow_none_selected_8x8"])
def test_flow_loss_oracle_vs_reference_golden(name):
    """get_flow_loss of the reference (values and autograd gradients) vs the NumPy restatement."""
    g = lambda k: GOLD2[name + "/" + k]
    op = g("opacity") if g("opacity").size else None
    loss, g_f, g_o = lo.flow_loss(g("img_flow"), g("flow"), g("vis"), op, g("K"), g("R"), g("T"), 0.02)
    np.testing.assert_allclose(loss, float(g("loss")), rtol=2e-5, atol=1e-8)
    np.testing.assert_allclose(g_f, g("g_img_flow"), rtol=2e-4, atol=2e-7)
    if op is not None:
        np.testing.assert_allclose(g_o, g("g_opacity"), rtol=2e-4, atol=2e-8)
    if name.startswith("flow_none"):
        assert loss == 0.0
    else:
        assert np.abs(g("g_img_flow")).max() > 0 and (g("img_flow")[2] < 0.5).any()


def test_bce_clip_oracle_vs_train_py_expression():
    g = lambda k: GOLD2["bce_19x31/" + k]
    loss, grad = lo.bce_clip_loss(g("pred"), (g("gt_sem") > 0).astype(np.float64))
    np.testing.assert_allclose(loss, float(g("obj")), rtol=1e-6); np.testing.assert_allclose(grad, g("g_obj"), rtol=2e-5, atol=1e-9)
    loss, grad = lo.bce_clip_loss(g("pred"), g("gt_sky"), invert=True)
    np.testing.assert_allclose(loss, float(g("sky")), rtol=1e-6); np.testing.assert_allclose(grad, g("g_sky"), rtol=2e-5, atol=1e-9)


GOLD3 = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "loss3_golden.npz"))
REG_CASES = sorted({k.split("/")[0] for k in GOLD3.files})


@pytest.mark.parametrize("case", REG_CASES)
def test_regulariser_oracles_vs_train_py_expressions(case):
    """reg_loss / sigma_loss / reg_sigma_loss (train.py:104-113) and their reference-autograd gradients."""
    x, s, idx = GOLD3[case + "/xyz_deform_param"], GOLD3[case + "/gs_time_sigma"], GOLD3[case + "/obj_near_idx"]
    loss, g = loss_oracle.group_var_loss(x, idx)
    assert abs(loss - float(GOLD3[case + "/reg_loss"])) <= 2e-6 * max(1.0, abs(loss))
    np.testing.assert_allclose(g, GOLD3[case + "/g_reg"], rtol=2e-5, atol=2e-6 * np.abs(g).max())      # the golden gradients are float32 autograd
    loss, g = loss_oracle.group_var_loss(s, idx)
    assert abs(loss - float(GOLD3[case + "/reg_sigma_loss"])) <= 2e-6 * max(1.0, abs(loss))
    np.testing.assert_allclose(g, GOLD3[case + "/g_reg_sigma"], rtol=2e-5, atol=2e-6 * np.abs(g).max())
    loss, g = loss_oracle.sigma_loss(s, float(GOLD3[case + "/frame_gap"]))
    assert abs(loss - float(GOLD3[case + "/sigma_loss"])) <= 2e-6 * max(1.0, abs(loss))
    np.testing.assert_allclose(g, GOLD3[case + "/g_sigma"], rtol=2e-5, atol=2e-6 * np.abs(g).max())


def test_group_variance_finite_differences_and_repeated_rows():
    rng = np.random.default_rng(5)
    x = rng.standard_normal((12, 3, 4))
    idx = np.array([[0, 1, 2], [2, 3, 0], [5, 5, 7]])             # a row twice in one group, rows shared between groups, rows in none
    loss, g = loss_oracle.group_var_loss(x, idx)
    for i in [(0, 0, 0), (2, 1, 3), (5, 2, 1), (7, 0, 2), (9, 1, 1)]:
        d = np.zeros_like(x); d[i] = 1e-6
        fd = (loss_oracle.group_var_loss(x + d, idx)[0] - loss_oracle.group_var_loss(x - d, idx)[0]) / 2e-6
        assert abs(fd - g[i]) <= 1e-7, (i, fd, g[i])
    assert np.all(g[[4, 6, 8, 9, 10, 11]] == 0)
