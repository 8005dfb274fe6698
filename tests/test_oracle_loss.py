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
