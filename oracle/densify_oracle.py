"""NumPy oracle of the reference's densification (scene/gaussian_model.py:465-469, 545-867).

TEST INFRASTRUCTURE ONLY: only tests/ may import this module.  The product path (ad-gs_amd/) never does.

A step-by-step restatement -- clone, split (cat + prune of the parents), final prune, with the optimizer-state surgery of
cat_tensors_to_optimizer / _prune_optimizer -- on a plain dict of float32 arrays:
    st["p"][group], st["m"][group], st["v"][group]   parameter, exp_avg, exp_avg_sq per optimizer group name
    st["gs_time"], st["xyz_gradient_accum"], st["denom"], st["max_radii2D"]
The normal samples of densify_and_split are an INPUT (the reference draws them with torch.normal).
Pinned by tests/test_oracle_densify.py on golden vectors produced by the reference's own Python
(tests/golden/make_densify_golden.py).
"""
import numpy as np

SCENE_GROUPS = ["scene_xyz", "scene_shs_dc", "scene_shs_rest", "scene_opacity", "scene_scaling", "scene_rotation", "deform_shs_scene"]
OBJ_GROUPS = ["obj_xyz", "obj_shs_dc", "obj_shs_rest", "obj_opacity", "obj_scaling", "obj_rotation", "deform_xyz", "deform_rotation",
              "deform_shs_obj", "time_sigma"]
f32 = np.float32


def _is_obj(name):                      # _prune_optimizer's rule (:566-569)
    return "obj" in name or name in ("deform_xyz", "deform_rotation", "deform_shs_obj", "time_sigma")


def build_rotation(r):                  # utils/general_utils.py:79-95
    r = r.astype(f32)
    norm = np.sqrt(r[:, 0] * r[:, 0] + r[:, 1] * r[:, 1] + r[:, 2] * r[:, 2] + r[:, 3] * r[:, 3], dtype=f32)
    q = r / norm[:, None]
    w, x, y, z = q[:, 0], q[:, 1], q[:, 2], q[:, 3]
    one, two = f32(1), f32(2)
    R = np.stack([one - two * (y * y + z * z), two * (x * y - w * z), two * (x * z + w * y),
                  two * (x * y + w * z), one - two * (x * x + z * z), two * (y * z - w * x),
                  two * (x * z - w * y), two * (y * z + w * x), one - two * (x * x + y * y)], axis=-1).reshape(-1, 3, 3)
    return R.astype(f32)


def sigmoid(x):
    return (f32(1) / (f32(1) + np.exp(-x.astype(f32), dtype=f32))).astype(f32)


def _cat(st, new):                      # cat_tensors_to_optimizer (:613-635) + densification_postfix (:637-703)
    for name, ext in new.items():
        ext = ext.astype(f32)
        st["p"][name] = np.concatenate([st["p"][name], ext], 0)
        if name in st["m"]:
            st["m"][name] = np.concatenate([st["m"][name], np.zeros_like(ext)], 0)
            st["v"][name] = np.concatenate([st["v"][name], np.zeros_like(ext)], 0)
    N = st["p"]["scene_xyz"].shape[0] + st["p"]["obj_xyz"].shape[0]
    st["xyz_gradient_accum"] = np.zeros((N, 1), f32)
    st["denom"] = np.zeros((N, 1), f32)
    st["max_radii2D"] = np.zeros((N,), f32)


def prune_points(st, scene_mask, obj_mask):      # :581-611 (masks = what to REMOVE)
    vs, vo = ~scene_mask, ~obj_mask
    for name in list(st["p"].keys()):
        if name == "deform_background":
            continue
        keep = vo if _is_obj(name) else vs
        st["p"][name] = st["p"][name][keep]
        if name in st["m"]:
            st["m"][name] = st["m"][name][keep]
            st["v"][name] = st["v"][name][keep]
    st["gs_time"] = st["gs_time"][vo]
    valid = np.concatenate([vs, vo], 0)
    st["xyz_gradient_accum"] = st["xyz_gradient_accum"][valid]
    st["denom"] = st["denom"][valid]
    st["max_radii2D"] = st["max_radii2D"][valid]


def densify_and_clone(st, sm, om, a):            # :769-823
    p = st["p"]
    sm = sm & (np.exp(p["scene_scaling"], dtype=f32).max(1) <= f32(a["scene_extent"] * a["percent_dense"]))
    om = om & (np.exp(p["obj_scaling"], dtype=f32).max(1) <= f32(a["object_extent"] * a["percent_dense"]))
    new = {n: p[n][sm] for n in SCENE_GROUPS}
    new.update({n: p[n][om] for n in OBJ_GROUPS})
    new_time = st["gs_time"][om]
    _cat(st, new)
    st["gs_time"] = np.concatenate([st["gs_time"], new_time], 0)


def densify_and_split(st, sm, om, a, samples_scene, samples_obj, N=2):   # :715-767
    p = st["p"]
    sm = sm & (np.exp(p["scene_scaling"], dtype=f32).max(1) > f32(a["scene_extent"] * a["percent_dense"]))
    om = om & (np.exp(p["obj_scaling"], dtype=f32).max(1) > f32(a["object_extent"] * a["percent_dense"]))
    new = {}
    for side, mask, samples in (("scene", sm, samples_scene), ("obj", om, samples_obj)):
        rep = lambda x: np.tile(x, (N,) + (1,) * (x.ndim - 1))
        samples = np.asarray(samples, f32).reshape(-1, 3)
        assert samples.shape[0] == N * int(mask.sum()), "sample count must match the number of split parents"
        rots = rep(build_rotation(p[side + "_rotation"][mask]))
        xyz = np.einsum("nij,nj->ni", rots, samples).astype(f32) + rep(p[side + "_xyz"][mask])
        new[side + "_xyz"] = xyz
        new[side + "_scaling"] = np.log(rep(np.exp(p[side + "_scaling"][mask], dtype=f32)) / f32(0.8 * N), dtype=f32)
        for n in (side + "_rotation", side + "_shs_dc", side + "_shs_rest", side + "_opacity", "deform_shs_" + side):
            new[n] = rep(p[n][mask])
    for n in ("deform_xyz", "deform_rotation", "time_sigma"):
        new[n] = np.tile(p[n][om], (N,) + (1,) * (p[n].ndim - 1))
    new_time = np.tile(st["gs_time"][om], (N, 1))
    _cat(st, new)
    st["gs_time"] = np.concatenate([st["gs_time"], new_time], 0)
    prune_points(st, np.concatenate([sm, np.zeros(N * int(sm.sum()), bool)]), np.concatenate([om, np.zeros(N * int(om.sum()), bool)]))


def densify_and_prune(st, a, samples_scene, samples_obj):      # :835-861
    """a: dict(max_scene_grad, max_obj_grad, min_opacity, prune_big_points, percent_dense, scene_extent, object_extent)."""
    Ns = st["p"]["scene_xyz"].shape[0]
    with np.errstate(divide="ignore", invalid="ignore"):
        grads = st["xyz_gradient_accum"].astype(f32) / st["denom"].astype(f32)
    grads[np.isnan(grads)] = 0.0
    grads = np.abs(grads[:, 0])                              # torch.norm over a length-1 last dimension
    sm, om = grads[:Ns] >= f32(a["max_scene_grad"]), grads[Ns:] >= f32(a["max_obj_grad"])
    densify_and_clone(st, sm, om, a)
    p = st["p"]
    sm = np.concatenate([sm, np.zeros(p["scene_xyz"].shape[0] - sm.shape[0], bool)])
    om = np.concatenate([om, np.zeros(p["obj_xyz"].shape[0] - om.shape[0], bool)])
    densify_and_split(st, sm, om, a, samples_scene, samples_obj)
    p = st["p"]
    sp = sigmoid(p["scene_opacity"])[:, 0] < f32(a["min_opacity"])
    op = sigmoid(p["obj_opacity"])[:, 0] < f32(a["min_opacity"])
    if a["prune_big_points"]:
        sp |= np.exp(p["scene_scaling"], dtype=f32).max(1) > f32(a["scene_extent"] * 0.05)
        op |= np.exp(p["obj_scaling"], dtype=f32).max(1) > f32(a["object_extent"] * 0.1)
    prune_points(st, sp, op)


def reset_opacity(st):                  # :465-469 (replace_tensor_to_optimizer zeroes the Adam moments, :546-559)
    for name in ("scene_opacity", "obj_opacity"):
        x = np.minimum(sigmoid(st["p"][name]), f32(0.01)).astype(f32)
        st["p"][name] = np.log(x / (f32(1) - x), dtype=f32)
        if name in st["m"]:
            st["m"][name] = np.zeros_like(st["p"][name]); st["v"][name] = np.zeros_like(st["p"][name])


def add_densification_stats(st, viewspace_grad, visibility_filter):      # :863-867
    g = np.asarray(viewspace_grad, f32)
    n = np.sqrt(g[:, 0] * g[:, 0] + g[:, 1] * g[:, 1], dtype=f32)[:, None]
    f = np.asarray(visibility_filter, bool)
    st["xyz_gradient_accum"][f] += n[f]
    st["denom"][f] += 1
