#!/usr/bin/env python3
"""Generates tests/golden/densify_golden.npz by IMPORTING THE REFERENCE's own Python (scene/gaussian_model.py:
densify_and_prune, densify_and_clone, densify_and_split, prune_points, cat_tensors_to_optimizer, _prune_optimizer,
reset_opacity, add_densification_stats) from /root/reference in the authoring container and running it on the CPU.
Only inputs and outputs (data) are stored -- no reference source.

Run:  python tests/golden/make_densify_golden.py        (needs /root/reference; CPU only)

Shims: the stub modules / device rewrite of make_deform_golden.py.  torch.normal is wrapped to RECORD the samples the
reference draws in densify_and_split (scene first, then object; scene/gaussian_model.py:719-720,730-731), so that the
oracle and the HIP path can be fed the same noise.
"""
import os
import sys
import types

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from make_deform_golden import REF, CudaToCpu, install_stubs  # noqa: E402

OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "densify_golden.npz")

GROUPS = [("scene_xyz", "_scene_xyz"), ("scene_shs_dc", "_scene_shs_dc"), ("scene_shs_rest", "_scene_shs_rest"), ("scene_opacity", "_scene_opacity"),
          ("scene_scaling", "_scene_scaling"), ("scene_rotation", "_scene_rotation"), ("obj_xyz", "_obj_xyz"), ("obj_shs_dc", "_obj_shs_dc"),
          ("obj_shs_rest", "_obj_shs_rest"), ("obj_opacity", "_obj_opacity"), ("obj_scaling", "_obj_scaling"), ("obj_rotation", "_obj_rotation"),
          ("deform_rotation", "rotation_deform_param"), ("deform_shs_scene", "shs_deform_param_scene"), ("deform_shs_obj", "shs_deform_param_obj"),
          ("deform_xyz", "xyz_deform_param"), ("deform_background", "background_deform_param"), ("time_sigma", "gs_time_sigma")]


def build_model(GaussianModel, func_utils, Ns, No, seed):
    oargs = dict(xyz=[6, 5, 0, 6, 0, 0], rotation=[0, 0, 0, 0, 6, 5], shs=[0, 0, 0, 6, 0, 0], background=[0] * 6)
    torch.manual_seed(seed)
    gm = GaussianModel(3, oargs)
    P = lambda t: torch.nn.Parameter(t.requires_grad_(True))
    r = lambda *s: torch.randn(*s)
    gm._scene_xyz, gm._obj_xyz = P(r(Ns, 3) * 5), P(r(No, 3))
    gm._scene_shs_dc, gm._obj_shs_dc = P(r(Ns, 1, 3)), P(r(No, 1, 3))
    gm._scene_shs_rest, gm._obj_shs_rest = P(r(Ns, 15, 3) * 0.1), P(r(No, 15, 3) * 0.1)
    gm._scene_scaling, gm._obj_scaling = P(r(Ns, 3) * 0.8 - 1.5), P(r(No, 3) * 0.8 - 2.0)
    gm._scene_rotation, gm._obj_rotation = P(r(Ns, 4)), P(r(No, 4))
    gm._scene_opacity, gm._obj_opacity = P(r(Ns, 1) * 2.5 - 1.0), P(r(No, 1) * 2.5 - 1.0)
    gm.xyz_deform_param = P(r(No, 3, func_utils.get_param_num(oargs["xyz"])) * 0.1)
    gm.rotation_deform_param = P(r(No, 4, func_utils.get_param_num(oargs["rotation"])) * 0.2)
    gm.shs_deform_param_scene = P(r(Ns, 3, func_utils.get_param_num(oargs["shs"])) * 0.1)
    gm.shs_deform_param_obj = P(r(No, 3, func_utils.get_param_num(oargs["shs"])) * 0.1)
    gm.background_deform_param = P(torch.zeros(1, 3, 0))
    gm.gs_time = torch.rand(No, 1)
    gm.gs_time_sigma = P(torch.randn(No, 2) * 0.3 - 1.5)
    gm.use_time_mask = True
    gm.percent_dense, gm.scene_extent, gm.object_extent = 0.01, 20.0, 8.0
    groups = [{"params": [getattr(gm, attr)], "lr": 1e-3, "name": name} for name, attr in GROUPS]
    gm.optimizer = torch.optim.Adam(groups, lr=0.0, eps=1e-15)
    # one real optimizer step so that exp_avg / exp_avg_sq are populated (the background group has no elements)
    for name, attr in GROUPS:
        p = getattr(gm, attr)
        p.grad = torch.randn_like(p) * 0.01
    gm.optimizer.step()
    gm.optimizer.zero_grad(set_to_none=True)
    N = Ns + No
    gm.xyz_gradient_accum = torch.rand(N, 1) * 3e-3
    gm.denom = torch.randint(0, 4, (N, 1)).float()          # zeros exercise the NaN -> 0 rule
    gm.max_radii2D = torch.rand(N) * 30
    return gm


def snapshot(gm, out, pre):
    for name, attr in GROUPS:
        p = getattr(gm, attr)
        out[pre + "p_" + name] = p.detach().numpy().copy()
        st = gm.optimizer.state.get(p, None)
        if st is not None and len(st):
            out[pre + "m_" + name] = st["exp_avg"].numpy().copy()
            out[pre + "v_" + name] = st["exp_avg_sq"].numpy().copy()
            out[pre + "step_" + name] = np.asarray(float(st["step"]))
    out[pre + "gs_time"] = gm.gs_time.numpy().copy()
    out[pre + "xyz_gradient_accum"] = gm.xyz_gradient_accum.numpy().copy()
    out[pre + "denom"] = gm.denom.numpy().copy()
    out[pre + "max_radii2D"] = gm.max_radii2D.numpy().copy()


def main():
    install_stubs()
    sys.path.insert(0, REF)
    out = {}
    with CudaToCpu():
        from utils import func_utils
        from scene.gaussian_model import GaussianModel
        real_normal = torch.normal
        cases = dict(small=(24, 12, 0, 8e-4, 6e-4, 0.005, False), big=(70, 30, 1, 8e-4, 6e-4, 0.005, True), none_selected=(16, 8, 2, 1.0, 1.0, -1e9, False))
        for tag, (Ns, No, seed, thr_s, thr_o, min_op, big) in cases.items():
            gm = build_model(GaussianModel, func_utils, Ns, No, seed)
            if tag == "none_selected":           # consistent statistics (no x/0 = inf rows): nothing passes the threshold
                gm.xyz_gradient_accum = gm.xyz_gradient_accum * (gm.denom > 0)
            pre = "dp_%s_" % tag
            snapshot(gm, out, pre + "in_")
            out[pre + "args"] = np.array([thr_s, thr_o, min_op, float(big), gm.percent_dense, gm.scene_extent, gm.object_extent], np.float64)
            drawn = []

            def rec_normal(*a, **k):
                s = real_normal(*a, **k)
                drawn.append(s.detach().clone())
                return s
            torch.normal = rec_normal
            try:
                torch.manual_seed(100 + seed)
                gm.densify_and_prune(thr_s, thr_o, min_op, big)
            finally:
                torch.normal = real_normal
            assert len(drawn) == 2
            out[pre + "samples_scene"], out[pre + "samples_obj"] = drawn[0].numpy(), drawn[1].numpy()
            snapshot(gm, out, pre + "out_")
            print(tag, "N", Ns + No, "->", gm.get_pts_num, "split parents", drawn[0].shape[0] // 2, drawn[1].shape[0] // 2)
        # reset_opacity (scene/gaussian_model.py:465-469) and add_densification_stats (:863-867)
        gm = build_model(GaussianModel, func_utils, 20, 10, 5)
        snapshot(gm, out, "ro_in_")
        gm.reset_opacity()
        snapshot(gm, out, "ro_out_")
        N = gm.get_pts_num
        vs = torch.randn(N, 3).requires_grad_(True)
        vs.grad = torch.randn(N, 3) * 1e-3
        filt = torch.rand(N) < 0.6
        out["st_grad"], out["st_filter"] = vs.grad.numpy().copy(), filt.numpy().copy()
        gm.add_densification_stats(dict(viewspace_points=vs, visibility_filter=filt))
        out["st_out_accum"], out["st_out_denom"] = gm.xyz_gradient_accum.numpy().copy(), gm.denom.numpy().copy()
    np.savez_compressed(OUT, **out)
    print("wrote", OUT, "with", len(out), "arrays,", os.path.getsize(OUT), "bytes")


if __name__ == "__main__":
    main()
