"""The densification oracle (oracle/densify_oracle.py) against golden vectors produced by the REFERENCE's own Python
(tests/golden/make_densify_golden.py: densify_and_prune / reset_opacity / add_densification_stats on the CPU).
Row selection, ordering and optimizer-state surgery are bit-exact; the two computed quantities (split positions and
scales) are within float32 rounding of a different libm."""
import os

import numpy as np
import pytest

from oracle import densify_oracle as do

GOLD = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "densify_golden.npz"))
GROUPS = do.SCENE_GROUPS + do.OBJ_GROUPS + ["deform_background"]


def load_state(pre):
    st = dict(p={}, m={}, v={})
    for g in GROUPS:
        st["p"][g] = GOLD[pre + "p_" + g].copy()
        if pre + "m_" + g in GOLD.files:
            st["m"][g] = GOLD[pre + "m_" + g].copy(); st["v"][g] = GOLD[pre + "v_" + g].copy()
    for k in ("gs_time", "xyz_gradient_accum", "denom", "max_radii2D"):
        st[k] = GOLD[pre + k].copy()
    return st


def args_of(tag):
    a = GOLD["dp_%s_args" % tag]
    return dict(max_scene_grad=a[0], max_obj_grad=a[1], min_opacity=a[2], prune_big_points=bool(a[3]), percent_dense=a[4], scene_extent=a[5],
                object_extent=a[6])


def compare_state(got, pre, computed_tol=2e-6):
    for g in GROUPS:
        want = GOLD[pre + "p_" + g]
        assert got["p"][g].shape == want.shape, g
        if g.endswith("_xyz") and not g.startswith("deform") or g.endswith("_scaling"):
            np.testing.assert_allclose(got["p"][g], want, rtol=computed_tol, atol=computed_tol, err_msg=g)
        else:
            assert np.array_equal(got["p"][g], want), g
        if pre + "m_" + g in GOLD.files:
            assert np.array_equal(got["m"][g], GOLD[pre + "m_" + g]) and np.array_equal(got["v"][g], GOLD[pre + "v_" + g]), g
    for k in ("gs_time", "xyz_gradient_accum", "denom", "max_radii2D"):
        assert np.array_equal(got[k], GOLD[pre + k]), k


@pytest.mark.parametrize("tag", ["small", "big", "none_selected"])
def test_densify_and_prune_oracle_vs_reference_golden(tag):
    st = load_state("dp_%s_in_" % tag)
    n0 = st["p"]["scene_xyz"].shape[0] + st["p"]["obj_xyz"].shape[0]
    do.densify_and_prune(st, args_of(tag), GOLD["dp_%s_samples_scene" % tag], GOLD["dp_%s_samples_obj" % tag])
    compare_state(st, "dp_%s_out_" % tag)
    n1 = st["p"]["scene_xyz"].shape[0] + st["p"]["obj_xyz"].shape[0]
    assert (n1 != n0) == (tag != "none_selected")


def test_reset_opacity_and_stats_oracle_vs_reference_golden():
    st = load_state("ro_in_")
    do.reset_opacity(st)
    for g in ("scene_opacity", "obj_opacity"):
        np.testing.assert_allclose(st["p"][g], GOLD["ro_out_p_" + g], rtol=2e-6, atol=2e-6)
        assert not st["m"][g].any() and not st["v"][g].any() and not GOLD["ro_out_m_" + g].any()
    for g in GROUPS:
        if g not in ("scene_opacity", "obj_opacity"):
            assert np.array_equal(st["p"][g], GOLD["ro_out_p_" + g])
    do.add_densification_stats(st, GOLD["st_grad"], GOLD["st_filter"])
    np.testing.assert_allclose(st["xyz_gradient_accum"], GOLD["st_out_accum"], rtol=1e-6, atol=1e-9)
    assert np.array_equal(st["denom"], GOLD["st_out_denom"])
