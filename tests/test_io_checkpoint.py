"""Checkpoint interchange (adgs.io): the PLY + deform.pth pair the reference's GaussianModel.save_ply / load_ply exchange
(scene/gaussian_model.py:413-541).  Header and column layout are checked against the reference's attribute list and its
[N, 3, coeffs] -> flattened column convention; save -> load is bit-exact."""
import os

import numpy as np
import torch

from adgs import io as aio


class _M:
    max_sh_degree = 3


def _model(Ns=7, No=5, seed=0):
    g = torch.Generator().manual_seed(seed)
    r = lambda *s: torch.randn(*s, generator=g)
    m = _M()
    m._scene_xyz, m._obj_xyz = r(Ns, 3), r(No, 3)
    m._scene_shs_dc, m._obj_shs_dc = r(Ns, 1, 3), r(No, 1, 3)
    m._scene_shs_rest, m._obj_shs_rest = r(Ns, 15, 3), r(No, 15, 3)
    m._scene_opacity, m._obj_opacity = r(Ns, 1), r(No, 1)
    m._scene_scaling, m._obj_scaling = r(Ns, 3), r(No, 3)
    m._scene_rotation, m._obj_rotation = r(Ns, 4), r(No, 4)
    m.order_args = dict(xyz=[6, 5, 0, 6, 0, 0], rotation=[0, 0, 0, 0, 6, 5], shs=[0, 0, 0, 6, 0, 0], background=[0] * 6)
    m.xyz_deform_param, m.rotation_deform_param = r(No, 3, 18), r(No, 4, 6)
    m.shs_deform_param_scene, m.shs_deform_param_obj, m.background_deform_param = r(Ns, 3, 12), r(No, 3, 12), torch.zeros(1, 3, 0)
    m.gs_time, m.gs_time_sigma, m.use_time_mask, m.scene_extent = torch.rand(No, 1, generator=g), r(No, 2), True, 17.5
    return m


def test_ply_layout_is_the_references(tmp_path):
    m = _model()
    path = os.path.join(str(tmp_path), "point_cloud", "iteration_7", "point_cloud.ply")
    aio.save_ply(m, path)
    raw = open(path, "rb").read()
    head, body = raw.split(b"end_header\n", 1)
    lines = head.decode().splitlines()
    assert lines[:3] == ["ply", "format binary_little_endian 1.0", "element vertex 12"]
    names = [l.split()[2] for l in lines[3:]]
    assert all(l.startswith("property float ") for l in lines[3:])
    want = ['x', 'y', 'z', 'nx', 'ny', 'nz'] + ["shs_dc_%d" % i for i in range(3)] + ["shs_rest_%d" % i for i in range(45)] + ["opacity"] + \
        ["scale_%d" % i for i in range(3)] + ["rot_%d" % i for i in range(4)] + ["obj"]            # construct_list_of_attributes (:413-427)
    assert names == want and len(body) == 12 * len(want) * 4
    t = np.frombuffer(body, "<f4").reshape(12, len(want))
    assert np.array_equal(t[:7, :3], m._scene_xyz.numpy()) and np.array_equal(t[7:, :3], m._obj_xyz.numpy()) and not t[:, 3:6].any()
    # transpose(1, 2).flatten: column shs_rest_{c * 15 + k} = coefficient k of colour channel c
    assert np.array_equal(t[:7, 9 + 1 * 15 + 4], m._scene_shs_rest[:, 4, 1].numpy())
    assert np.array_equal(t[:, -1], np.r_[np.zeros(7), np.ones(5)].astype(np.float32))
    assert os.path.exists(os.path.join(os.path.dirname(path), "deform.pth"))


def test_save_load_round_trip_is_bit_exact(tmp_path):
    m = _model(9, 4, 3)
    path = os.path.join(str(tmp_path), "point_cloud.ply")
    aio.save_ply(m, path)
    n = aio.load_ply(_M(), path, device="cpu")
    for k in ("_scene_xyz", "_obj_xyz", "_scene_shs_dc", "_obj_shs_dc", "_scene_shs_rest", "_obj_shs_rest", "_scene_opacity", "_obj_opacity",
              "_scene_scaling", "_obj_scaling", "_scene_rotation", "_obj_rotation", "xyz_deform_param", "rotation_deform_param",
              "shs_deform_param_scene", "shs_deform_param_obj", "background_deform_param", "gs_time", "gs_time_sigma"):
        a, b = getattr(m, k), getattr(n, k)
        assert a.shape == b.shape and torch.equal(a, b.detach()), k
        assert k == "gs_time" or (isinstance(b, torch.nn.Parameter) and b.requires_grad), k
    assert n.use_time_mask is True and n.order_args == m.order_args and n.scene_extent == 17.5 and n.active_sh_degree == 3


def test_reader_accepts_ascii_and_other_elements(tmp_path):
    p = os.path.join(str(tmp_path), "a.ply")
    open(p, "w").write("ply\nformat ascii 1.0\ncomment made by hand\nelement vertex 2\nproperty float x\nproperty double y\nproperty uchar obj\n"
                       "element face 0\nproperty list uchar int vertex_indices\nend_header\n1.5 2.5 1\n-3 4 0\n")
    names, c = aio.read_ply(p)
    assert names == ["x", "y", "obj"] and c["x"].tolist() == [1.5, -3.0] and c["y"].tolist() == [2.5, 4.0] and c["obj"].tolist() == [1, 0]
