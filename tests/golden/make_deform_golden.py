#!/usr/bin/env python3
"""Generates tests/golden/deform_golden.npz by IMPORTING THE REFERENCE's own Python
(utils/func_utils.py, scene/gaussian_model.py) from /root/reference in the authoring
container.  Only inputs and outputs (data) are stored -- no reference source.

Run:  python tests/golden/make_deform_golden.py        (needs /root/reference; CPU only)

Shims:
  * stub modules for imports that are absent here (roma, plyfile, simple_knn._C,
    pytorch3d.ops, cv2, open3d);
  * a TorchFunctionMode that rewrites device='cuda' -> 'cpu' (hard-coded in the reference).
`roma` is NOT available, so the four quaternion helpers are supplied by a torch
restatement of their published algorithm: cases that go through them are stored under
keys prefixed `quat_` and are "unpinned at the roma boundary" (they still pin the
reference's own slicing / normalisation / cumulative-basis code around those calls).
"""
import os
import sys
import types

import numpy as np
import torch

REF = "/root/reference"
OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "deform_golden.npz")


# ---------------- roma restatement in torch (XYZW) ----------------
def quat_conjugation(q):
    out = q.clone()
    out[..., :3] = -out[..., :3]
    return out


def quat_product(p, q):
    vec = p[..., 3:4] * q[..., :3] + q[..., 3:4] * p[..., :3] + torch.cross(p[..., :3], q[..., :3], dim=-1)
    last = p[..., 3] * q[..., 3] - torch.sum(p[..., :3] * q[..., :3], dim=-1)
    return torch.cat([vec, last[..., None]], dim=-1)


def unitquat_to_rotvec(quat, shortest_arc=True):
    shape = quat.shape[:-1]
    q = quat.reshape(-1, 4)
    if shortest_arc:
        q = torch.where(q[:, 3:4] < 0, -q, q)
    angle = 2 * torch.atan2(torch.norm(q[:, :3], dim=1), q[:, 3])
    small = angle <= 1e-3
    scale = torch.where(small, 2 + angle ** 2 / 12 + 7 * angle ** 4 / 2880, angle / torch.sin(torch.where(small, torch.ones_like(angle), angle) / 2))
    return (scale[:, None] * q[:, :3]).reshape(*shape, 3)


def rotvec_to_unitquat(rotvec):
    shape = rotvec.shape[:-1]
    rv = rotvec.reshape(-1, 3)
    n = torch.norm(rv, dim=-1)
    small = n <= 1e-3
    safe = torch.where(small, torch.ones_like(n), n)
    scale = torch.where(small, 0.5 - n ** 2 / 48 + n ** 4 / 3840, torch.sin(safe / 2) / safe)
    return torch.cat([scale[:, None] * rv, torch.cos(n / 2)[:, None]], dim=-1).reshape(*shape, 4)


def install_stubs():
    roma = types.ModuleType("roma")
    roma.unitquat_slerp = lambda *a, **k: (_ for _ in ()).throw(NotImplementedError())
    roma.unitquat_to_rotvec = unitquat_to_rotvec
    roma.rotvec_to_unitquat = rotvec_to_unitquat
    roma.quat_conjugation = quat_conjugation
    roma.quat_product = quat_product
    sys.modules["roma"] = roma
    for name in ("plyfile", "simple_knn", "simple_knn._C", "pytorch3d", "pytorch3d.ops", "cv2", "open3d"):
        sys.modules.setdefault(name, types.ModuleType(name))
    sys.modules["plyfile"].PlyData = object
    sys.modules["plyfile"].PlyElement = object
    sys.modules["simple_knn._C"].distCUDA2 = None
    sys.modules["pytorch3d.ops"].knn_points = None


class CudaToCpu(torch.overrides.TorchFunctionMode):
    def __torch_function__(self, func, types_, args=(), kwargs=None):
        kwargs = dict(kwargs or {})
        if kwargs.get("device") is not None and "cuda" in str(kwargs["device"]):
            kwargs["device"] = "cpu"
        return func(*args, **kwargs)


def main():
    install_stubs()
    sys.path.insert(0, REF)
    out = {}
    with CudaToCpu():
        from utils import func_utils
        from scene.gaussian_model import GaussianModel

        # ---- get_func_result on the order_args / v grid of SURVEY.md 8(c) ----
        cases = {
            "bs6k5_fft6": [6, 5, 0, 6, 0, 0], "bs9k5_fft6": [9, 5, 0, 6, 0, 0], "bs50k5_fft6": [50, 5, 0, 6, 0, 0],
            "bs7k1": [7, 1, 0, 0, 0, 0], "bs7k2": [7, 2, 0, 0, 0, 0], "poly3": [0, 0, 3, 0, 0, 0], "fft6": [0, 0, 0, 6, 0, 0],
            "bs8k3_poly2_fft4": [8, 3, 2, 4, 0, 0],
            "quat_q6k5": [0, 0, 0, 0, 6, 5], "quat_q9k2": [0, 0, 0, 0, 9, 2], "quat_q8k1": [0, 0, 0, 0, 8, 1],
        }
        vs = [0.0, 1e-3, 0.37, 0.5, 0.999, 1.0]
        out["vs"] = np.array(vs, np.float64)
        for name, oa in cases.items():
            D = 4 if oa[4] != 0 else 3
            torch.manual_seed(0)
            n_par = func_utils.get_param_num(oa)
            param0 = (torch.rand(8, D, n_par) * 2 - 1) * (0.3 if oa[4] != 0 else 1.0)
            out["func_%s_order" % name] = np.array(oa, np.int64)
            out["func_%s_param" % name] = param0.numpy()
            for vi, v in enumerate(vs):
                p = param0.clone().requires_grad_(True)
                r = func_utils.get_func_result(v, p, oa)
                w = torch.linspace(0.5, 1.5, r.numel()).reshape(r.shape)
                (r * w).sum().backward()
                out["func_%s_out_%d" % (name, vi)] = r.detach().numpy()
                out["func_%s_grad_%d" % (name, vi)] = p.grad.numpy()
        for k in range(0, 6):
            out["deboor_%d" % k] = func_utils.get_deboor_cox_mat(k)

        # ---- GaussianModel.get_deformed_pkg (scene/gaussian_model.py:216-231) ----
        for tag, oargs, mask in (
            ("waymo_like", dict(xyz=[6, 5, 0, 6, 0, 0], rotation=[0, 0, 0, 0, 0, 0], shs=[0, 0, 0, 6, 0, 0], background=[0] * 6), False),
            ("kitti_like_mask", dict(xyz=[9, 2, 0, 6, 0, 0], rotation=[0, 0, 0, 4, 0, 0], shs=[0, 0, 0, 6, 0, 0], background=[9, 2, 0, 6, 0, 0]), True),
            ("quat_rot", dict(xyz=[6, 5, 0, 6, 0, 0], rotation=[0, 0, 0, 0, 6, 5], shs=[0, 0, 0, 6, 0, 0], background=[0] * 6), True),
        ):
            torch.manual_seed(1)
            Ns, No = 5, 4
            gm = GaussianModel(3, oargs)
            r = lambda *s: torch.randn(*s)
            gm._scene_xyz, gm._obj_xyz = r(Ns, 3), r(No, 3)
            gm._scene_shs_dc, gm._obj_shs_dc = r(Ns, 1, 3), r(No, 1, 3)
            gm._scene_shs_rest, gm._obj_shs_rest = r(Ns, 15, 3) * 0.1, r(No, 15, 3) * 0.1
            gm._scene_scaling, gm._obj_scaling = r(Ns, 3) * 0.3 - 2, r(No, 3) * 0.3 - 2
            gm._scene_rotation, gm._obj_rotation = r(Ns, 4), r(No, 4)
            gm._scene_opacity, gm._obj_opacity = r(Ns, 1), r(No, 1)
            gm.xyz_deform_param = r(No, 3, func_utils.get_param_num(oargs["xyz"])) * 0.1
            gm.rotation_deform_param = r(No, 4, max(func_utils.get_param_num(oargs["rotation"]), 0)) * 0.2
            gm.shs_deform_param_scene = r(Ns, 3, func_utils.get_param_num(oargs["shs"])) * 0.1
            gm.shs_deform_param_obj = r(No, 3, func_utils.get_param_num(oargs["shs"])) * 0.1
            gm.background_deform_param = r(1, 3, func_utils.get_param_num(oargs["background"])) * 0.1
            gm.gs_time = torch.rand(No, 1)
            gm.gs_time_sigma = torch.randn(No, 2) * 0.3 - 1.5
            gm.use_time_mask = mask
            names = ["_scene_xyz", "_obj_xyz", "_scene_shs_dc", "_obj_shs_dc", "_scene_shs_rest", "_obj_shs_rest", "_scene_scaling",
                     "_obj_scaling", "_scene_rotation", "_obj_rotation", "_scene_opacity", "_obj_opacity", "xyz_deform_param",
                     "rotation_deform_param", "shs_deform_param_scene", "shs_deform_param_obj", "background_deform_param", "gs_time",
                     "gs_time_sigma"]
            pre = "pkg_%s_" % tag
            for n in names:
                out[pre + "in_" + n.lstrip("_")] = getattr(gm, n).numpy()
            out[pre + "use_time_mask"] = np.array(int(mask))
            for kk, vv in oargs.items():
                out[pre + "order_" + kk] = np.array(vv, np.int64)
            for ti, t in enumerate((0.0, 0.4, 1.0)):
                pkg = gm.get_deformed_pkg(t)
                for kk, vv in pkg.items():
                    out[pre + "t%d_%s" % (ti, kk)] = vv.detach().numpy()
                out[pre + "t%d_scales" % ti] = gm.get_scaling.detach().numpy()
            out[pre + "ts"] = np.array([0.0, 0.4, 1.0])
    np.savez_compressed(OUT, **out)
    print("wrote", OUT, "with", len(out), "arrays,", os.path.getsize(OUT), "bytes")


if __name__ == "__main__":
    main()
