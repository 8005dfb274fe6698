"""Pins the NumPy deformation oracle (oracle/deform_oracle.py) against golden vectors produced by
the REFERENCE's own Python (tests/golden/make_deform_golden.py -> deform_golden.npz), and the
restated roma quaternion helpers against scipy."""
import os

import numpy as np
import pytest

from oracle import deform_oracle as do

GOLD = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "deform_golden.npz"))
FUNC_CASES = sorted({k[len("func_"):-len("_order")] for k in GOLD.files if k.startswith("func_") and k.endswith("_order")})
PKG_CASES = sorted({k[len("pkg_"):-len("_ts")] for k in GOLD.files if k.startswith("pkg_") and k.endswith("_ts")})


def test_golden_file_is_complete():
    assert len(FUNC_CASES) == 11 and len(PKG_CASES) == 3
    assert set(GOLD["vs"].tolist()) == {0.0, 1e-3, 0.37, 0.5, 0.999, 1.0}


@pytest.mark.parametrize("k", range(6))
def test_deboor_cox_matrix(k):
    np.testing.assert_array_equal(do.get_deboor_cox_mat(k), GOLD["deboor_%d" % k])
    if k == 3:      # the M_3 the reference documents (utils/func_utils.py:16-21)
        np.testing.assert_allclose(do.get_deboor_cox_mat(3) * 6, [[1, 4, 1, 0], [-3, 0, 3, 0], [3, -6, 3, 0], [-1, 3, -3, 1]], atol=1e-6)


@pytest.mark.parametrize("name", FUNC_CASES)
def test_get_func_result_matches_reference(name):
    oa = GOLD["func_%s_order" % name].tolist()
    param = GOLD["func_%s_param" % name]
    for vi, v in enumerate(GOLD["vs"].tolist()):
        ref = GOLD["func_%s_out_%d" % (name, vi)]
        got = do.get_func_result(v, param, oa)
        tol = 2e-5 if name.startswith("quat_") else 2e-6
        np.testing.assert_allclose(got, ref, rtol=tol, atol=tol, err_msg="%s v=%g" % (name, v))
        # float64 evaluation agrees too (no float32-specific trick is load-bearing)
        np.testing.assert_allclose(do.get_func_result(v, param, oa, np.float64), ref, rtol=5e-5, atol=5e-5)


def test_all_zero_orders_return_python_zero():
    assert do.get_func_result(0.3, np.zeros((2, 3, 0), np.float32), [0] * 6) == 0.0


def _model(tag):
    pre = "pkg_%s_" % tag
    m = {k[len(pre) + 3:]: GOLD[k] for k in GOLD.files if k.startswith(pre + "in_")}
    m["order_args"] = {k: GOLD[pre + "order_" + k].tolist() for k in ("xyz", "rotation", "shs", "background")}
    m["use_time_mask"] = bool(GOLD[pre + "use_time_mask"])
    return m, pre


@pytest.mark.parametrize("tag", PKG_CASES)
def test_get_deformed_pkg_matches_reference(tag):
    m, pre = _model(tag)
    for ti, t in enumerate(GOLD[pre + "ts"].tolist()):
        got = do.get_deformed_pkg(m, t)
        for key in ("xyz", "rotation", "shs", "opacity", "scales"):
            ref = GOLD[pre + "t%d_%s" % (ti, key)]
            np.testing.assert_allclose(got[key], ref, rtol=3e-6, atol=3e-6, err_msg="%s t=%g %s" % (tag, t, key))


def test_quaternion_helpers_match_scipy():
    from scipy.spatial.transform import Rotation as R
    rng = np.random.RandomState(0)
    q = rng.randn(200, 4)
    q /= np.linalg.norm(q, axis=1, keepdims=True)
    q[:20, :3] *= 1e-5; q[:20] /= np.linalg.norm(q[:20], axis=1, keepdims=True)      # tiny angles: series branch
    rv = do.unitquat_to_rotvec(q)
    np.testing.assert_allclose(rv, R.from_quat(q).as_rotvec(), atol=1e-12)
    back = do.rotvec_to_unitquat(rv)
    sign = np.sign(np.sum(back * q, axis=1, keepdims=True))
    np.testing.assert_allclose(back * sign, q, atol=1e-12)
    np.testing.assert_allclose(do.rotvec_to_unitquat(rv), R.from_rotvec(rv).as_quat(), atol=1e-12)
    p = rng.randn(200, 4); p /= np.linalg.norm(p, axis=1, keepdims=True)
    prod = do.quat_product(p, q)
    ref = (R.from_quat(p) * R.from_quat(q)).as_quat()
    sign = np.sign(np.sum(prod * ref, axis=1, keepdims=True))
    np.testing.assert_allclose(prod * sign, ref, atol=1e-12)
    np.testing.assert_allclose(do.quat_product(do.quat_conjugation(q), q), np.tile([0, 0, 0, 1.0], (200, 1)), atol=1e-12)


def test_quat_spline_interpolates_control_rotations_at_knots():
    """Analytic check of the cumulative form: with order k=1 the curve passes through control
    quaternion j at v = j/(n-1)."""
    rng = np.random.RandomState(3)
    n = 5
    param = (rng.randn(3, 4, n) * 0.4).astype(np.float32)
    ctrl = param + np.array([1, 0, 0, 0], np.float32).reshape(1, 4, 1)
    ctrl = ctrl / np.linalg.norm(ctrl, axis=1, keepdims=True)
    for j in range(n):
        v = j / (n - 1)
        out = do.get_func_result(v, param, [0, 0, 0, 0, n, 1], np.float64)
        tgt = ctrl[:, :, j]
        sign = np.sign(np.sum(out * tgt, axis=1, keepdims=True))
        np.testing.assert_allclose(out * sign, tgt, atol=2e-6)


@pytest.mark.parametrize("name", FUNC_CASES)
def test_torch_deform_ref_matches_reference_values_and_autograd(name):
    """The float64 torch restatement used as GRADIENT oracle for the HIP kernels reproduces the
    reference's outputs and the gradients of the reference's own autograd."""
    import torch
    from tests import torch_deform_ref as tr
    oa = GOLD["func_%s_order" % name].tolist()
    for vi, v in enumerate(GOLD["vs"].tolist()):
        p = torch.tensor(GOLD["func_%s_param" % name], dtype=torch.float64, requires_grad=True)
        r = tr.get_func_result(v, p, oa)
        w = torch.linspace(0.5, 1.5, r.numel(), dtype=torch.float64).reshape(r.shape)
        (r * w).sum().backward()
        tol = 3e-5 if name.startswith("quat_") else 3e-6
        np.testing.assert_allclose(r.detach().numpy(), GOLD["func_%s_out_%d" % (name, vi)], rtol=tol, atol=tol)
        np.testing.assert_allclose(p.grad.numpy(), GOLD["func_%s_grad_%d" % (name, vi)], rtol=10 * tol, atol=10 * tol)
