"""Differentiable torch (CPU, float64) restatement of get_func_result / get_deformed_pkg.
Test infrastructure: validated against the golden vectors AND golden gradients produced by the
reference's own autograd (tests/test_oracle_deform.py), then used as the gradient oracle for the
HIP deformation kernels."""
import numpy as np
import torch

from oracle import deform_oracle as do


def quat_conjugation(q):
    return torch.cat([-q[..., :3], q[..., 3:]], dim=-1)


def quat_product(p, q):
    vec = p[..., 3:4] * q[..., :3] + q[..., 3:4] * p[..., :3] + torch.cross(p[..., :3], q[..., :3], dim=-1)
    last = p[..., 3] * q[..., 3] - torch.sum(p[..., :3] * q[..., :3], dim=-1)
    return torch.cat([vec, last[..., None]], dim=-1)


def unitquat_to_rotvec(q):
    q = torch.where(q[..., 3:4] < 0, -q, q)
    angle = 2 * torch.atan2(torch.norm(q[..., :3], dim=-1), q[..., 3])
    small = angle <= 1e-3
    safe = torch.where(small, torch.ones_like(angle), angle)
    scale = torch.where(small, 2 + angle ** 2 / 12 + 7 * angle ** 4 / 2880, safe / torch.sin(safe / 2))
    return scale[..., None] * q[..., :3]


def rotvec_to_unitquat(rv):
    n = torch.norm(rv, dim=-1)
    small = n <= 1e-3
    safe = torch.where(small, torch.ones_like(n), n)
    scale = torch.where(small, 0.5 - n ** 2 / 48 + n ** 4 / 3840, torch.sin(safe / 2) / safe)
    return torch.cat([scale[..., None] * rv, torch.cos(n / 2)[..., None]], dim=-1)


def get_func_result(v, param, oa, dtype=torch.float64):
    result = 0.0
    offset = 0
    tt = lambda a: torch.tensor(np.asarray(a, dtype=np.float64), dtype=dtype)
    if oa[0] != 0:
        start, u = do.segment(v, oa[0], oa[1])
        func = tt(do.bspline_basis(u, oa[1]))
        result = result + torch.sum(param[..., start + offset: start + oa[1] + offset + 1] * func, dim=-1)
        offset += oa[0]
    if oa[2] != 0:
        result = result + torch.sum(param[..., offset: offset + oa[2]] * tt(do.poly_basis(v, oa[2])), dim=-1)
        offset += oa[2]
    if oa[3] != 0:
        result = result + torch.sum(param[..., offset: offset + 2 * oa[3]] * tt(do.fft_basis(v, oa[3])), dim=-1)
        offset += 2 * oa[3]
    if oa[4] != 0:
        start, u = do.segment(v, oa[4], oa[5])
        k = oa[5]
        ctrl = param[..., start + offset: start + k + offset + 1] + torch.tensor([1.0, 0, 0, 0], dtype=dtype).reshape(-1, 1)
        ctrl = torch.nn.functional.normalize(torch.permute(ctrl, (0, 2, 1)), dim=-1)[..., [1, 2, 3, 0]]
        func = do.bspline_basis(u, k)
        cum = tt(np.flip(np.cumsum(np.flip(func), dtype=np.float32))[1:].copy())
        vec = unitquat_to_rotvec(quat_product(quat_conjugation(ctrl[:, :-1, :]), ctrl[:, 1:, :]))
        quat = rotvec_to_unitquat(vec * cum[None, :, None])
        vector = ctrl[:, 0]
        for i in range(quat.shape[1]):
            vector = quat_product(vector, quat[:, i])
        result = result + vector[..., [3, 0, 1, 2]]
        offset += oa[4]
    return result


def get_deformed_pkg(m, t, oa, use_time_mask):
    """m: dict of torch float64 tensors named like oracle/deform_oracle.get_deformed_pkg's input."""
    obj_xyz = m["obj_xyz"] + get_func_result(t, m["xyz_deform_param"], oa["xyz"])
    xyz = torch.cat([m["scene_xyz"], obj_xyz], 0) + get_func_result(t, m["background_deform_param"], oa["background"])
    obj_rot = get_func_result(t, m["rotation_deform_param"], oa["rotation"])
    if oa["rotation"][4] == 0:
        obj_rot = m["obj_rotation"] + obj_rot
    rotation = torch.nn.functional.normalize(torch.cat([m["scene_rotation"], obj_rot], 0))
    shs_param = torch.cat([m["shs_deform_param_scene"], m["shs_deform_param_obj"]], 0)
    dc = torch.cat([m["scene_shs_dc"], m["obj_shs_dc"]], 0)[:, 0] + get_func_result(t, shs_param, oa["shs"])
    shs = torch.cat([dc[:, None], torch.cat([m["scene_shs_rest"], m["obj_shs_rest"]], 0)], 1)
    if use_time_mask:
        dt = t - m["gs_time"]
        sig = torch.exp(m["gs_time_sigma"])
        sig = torch.where(dt < 0.0, sig[:, :1], sig[:, 1:])
        mask = torch.exp(-0.5 * (dt / sig) ** 2)
        opacity = torch.cat([torch.sigmoid(m["scene_opacity"]), torch.sigmoid(m["obj_opacity"]) * mask], 0)
    else:
        opacity = torch.sigmoid(torch.cat([m["scene_opacity"], m["obj_opacity"]], 0))
    scales = torch.exp(torch.cat([m["scene_scaling"], m["obj_scaling"]], 0))
    return dict(xyz=xyz, rotation=rotation, shs=shs, opacity=opacity, scales=scales)
