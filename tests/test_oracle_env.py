"""The environment-map oracle against golden vectors from the reference's own scene/env.py (tests/golden/make_env_golden.py)."""
import os

import numpy as np
import pytest

from oracle import env_oracle

GOLD = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "env_golden.npz"))
CASES = sorted({k.split("/")[0] for k in GOLD.files})


@pytest.mark.parametrize("case", CASES)
def test_background_and_gradient_match_reference(case):
    gm, R, focal = GOLD[case + "/grid_map"], GOLD[case + "/R"], float(GOLD[case + "/focal"])
    H, W = [int(v) for v in GOLD[case + "/HW"]]
    bg = env_oracle.background(gm, H, W, focal, R)
    ref = GOLD[case + "/bg"]
    # the sample position carries float32 rounding of the reference's ray arithmetic: ~1e-5 texel on these small maps
    assert np.abs(bg - ref).max() <= 2e-5, np.abs(bg - ref).max()
    g = env_oracle.background_grad(gm, H, W, focal, R, GOLD[case + "/w"])
    gref = GOLD[case + "/g_grid"]
    assert np.abs(g - gref).max() <= 5e-5 * max(np.abs(gref).max(), 1.0)


def test_zero_padding_and_finite_difference():
    rng = np.random.default_rng(1)
    gm = rng.standard_normal((3, 16, 16))
    R = np.eye(3)
    bg = env_oracle.background(gm, 9, 11, 20.0, R)
    assert bg.shape == (3, 9, 11) and np.all((bg > 0) & (bg < 1))
    w = rng.standard_normal(bg.shape)
    g = env_oracle.background_grad(gm, 9, 11, 20.0, R, w)
    idx = np.unravel_index(np.argmax(np.abs(g)), g.shape)
    d = np.zeros_like(gm); d[idx] = 1e-5
    fd = ((env_oracle.background(gm + d, 9, 11, 20.0, R) - env_oracle.background(gm - d, 9, 11, 20.0, R)) * w).sum() / 2e-5
    assert abs(fd - g[idx]) <= 1e-6 * max(1.0, abs(g[idx]))
