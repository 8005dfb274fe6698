"""Checkpoint interchange with the reference: `point_cloud.ply` + `deform.pth` exactly as GaussianModel.save_ply / load_ply
write and read them (scene/gaussian_model.py:413-459, 465-541), so that models trained by either implementation load in the
other.  The reference goes through the `plyfile` package (not installed here); the binary little-endian PLY it produces --
one `vertex` element of float32 properties in construct_list_of_attributes order -- is written and parsed directly.
File I/O on the host; nothing here is on the per-frame path.
"""
import os

import numpy as np
import torch


def construct_list_of_attributes(n_dc, n_rest, n_scale, n_rot):
    """scene/gaussian_model.py:413-427."""
    l = ['x', 'y', 'z', 'nx', 'ny', 'nz']
    l += ['shs_dc_{}'.format(i) for i in range(n_dc)]
    l += ['shs_rest_{}'.format(i) for i in range(n_rest)]
    l.append('opacity')
    l += ['scale_{}'.format(i) for i in range(n_scale)]
    l += ['rot_{}'.format(i) for i in range(n_rot)]
    l.append('obj')
    return l


def write_ply(path, names, table):
    """One `vertex` element, every property float32, binary little endian (what plyfile's PlyData([el]).write emits)."""
    table = np.ascontiguousarray(table, dtype='<f4')
    assert table.ndim == 2 and table.shape[1] == len(names)
    header = "ply\nformat binary_little_endian 1.0\nelement vertex %d\n" % table.shape[0]
    header += "".join("property float %s\n" % n for n in names) + "end_header\n"
    with open(path, "wb") as f:
        f.write(header.encode("ascii"))
        f.write(table.tobytes())


_PLY_TYPES = {"char": "i1", "int8": "i1", "uchar": "u1", "uint8": "u1", "short": "i2", "int16": "i2", "ushort": "u2", "uint16": "u2",
              "int": "i4", "int32": "i4", "uint": "u4", "uint32": "u4", "float": "f4", "float32": "f4", "double": "f8", "float64": "f8"}


def read_ply(path):
    """First element of a PLY file (binary little/big endian or ascii, scalar properties) -> (names, {name: array})."""
    with open(path, "rb") as f:
        if f.readline().strip() != b"ply":
            raise ValueError("%s is not a PLY file" % path)
        fmt, count, props, in_first, seen = None, 0, [], False, 0
        while True:
            line = f.readline()
            if not line:
                raise ValueError("%s: unterminated PLY header" % path)
            tok = line.decode("ascii").split()
            if not tok or tok[0] == "comment" or tok[0] == "obj_info":
                continue
            if tok[0] == "format":
                fmt = tok[1]
            elif tok[0] == "element":
                seen += 1
                in_first = seen == 1
                if in_first:
                    count = int(tok[2])
            elif tok[0] == "property" and in_first:
                if tok[1] == "list":
                    raise ValueError("list properties are not supported")
                props.append((tok[2], _PLY_TYPES[tok[1]]))
            elif tok[0] == "end_header":
                break
        if fmt == "ascii":
            data = np.loadtxt(f, max_rows=count, ndmin=2) if count else np.zeros((0, len(props)))
            cols = {n: data[:, i].astype(t) for i, (n, t) in enumerate(props)}
        else:
            end = "<" if fmt == "binary_little_endian" else ">"
            rec = np.dtype([(n, end + t) for n, t in props])
            data = np.frombuffer(f.read(count * rec.itemsize), dtype=rec, count=count)
            cols = {n: np.asarray(data[n]) for n, _ in props}
    return [n for n, _ in props], cols


def save_ply(model, path):
    """GaussianModel.save_ply (:429-463): point_cloud.ply + deform.pth next to it."""
    os.makedirs(os.path.dirname(os.path.abspath(path)), exist_ok=True)
    cat = lambda a, b: torch.cat([a.detach(), b.detach()], dim=0)
    xyz = cat(model._scene_xyz, model._obj_xyz).cpu().numpy()
    shs_dc = cat(model._scene_shs_dc, model._obj_shs_dc).transpose(1, 2).flatten(start_dim=1).contiguous().cpu().numpy()
    shs_rest = cat(model._scene_shs_rest, model._obj_shs_rest).transpose(1, 2).flatten(start_dim=1).contiguous().cpu().numpy()
    opacities = cat(model._scene_opacity, model._obj_opacity).cpu().numpy()
    scale = cat(model._scene_scaling, model._obj_scaling).cpu().numpy()
    rotation = cat(model._scene_rotation, model._obj_rotation).cpu().numpy()
    Ns, No = model._scene_xyz.shape[0], model._obj_xyz.shape[0]
    obj = np.concatenate([np.zeros((Ns, 1), np.float32), np.ones((No, 1), np.float32)], 0)
    names = construct_list_of_attributes(shs_dc.shape[1], shs_rest.shape[1], scale.shape[1], rotation.shape[1])
    write_ply(path, names, np.concatenate((xyz, np.zeros_like(xyz), shs_dc, shs_rest, opacities, scale, rotation, obj), axis=1))
    torch.save((model.xyz_deform_param, model.rotation_deform_param, model.shs_deform_param_scene, model.shs_deform_param_obj,
                model.background_deform_param, model.gs_time, model.gs_time_sigma, model.use_time_mask, model.order_args,
                getattr(model, "scene_extent", 0.0)), os.path.join(os.path.dirname(os.path.abspath(path)), "deform.pth"))


def load_ply(model, path, device="cuda"):
    """GaussianModel.load_ply (:465-541): fills the raw parameters of `model` from point_cloud.ply + deform.pth."""
    names, c = read_ply(path)
    col = lambda n: np.asarray(c[n], np.float64)
    xyz = np.stack((col("x"), col("y"), col("z")), axis=1)
    opacities = col("opacity")[..., np.newaxis]
    obj_mask = col("obj") > 0.5
    scene_mask = np.logical_not(obj_mask)
    shs_dc = np.zeros((xyz.shape[0], 3, 1))
    for i in range(3):
        shs_dc[:, i, 0] = col("shs_dc_%d" % i)
    by_index = lambda prefix: sorted([n for n in names if n.startswith(prefix)], key=lambda x: int(x.split('_')[-1]))
    extra = by_index("shs_rest_")
    M = (model.max_sh_degree + 1) ** 2
    if len(extra) != 3 * M - 3:
        raise ValueError("the PLY holds %d shs_rest columns, max_sh_degree %d needs %d" % (len(extra), model.max_sh_degree, 3 * M - 3))
    shs_extra = np.stack([col(n) for n in extra], axis=1).reshape((xyz.shape[0], 3, M - 1)) if extra else np.zeros((xyz.shape[0], 3, 0))
    scales = np.stack([col(n) for n in by_index("scale_")], axis=1)
    rots = np.stack([col(n) for n in by_index("rot_")], axis=1)
    P = lambda a: torch.nn.Parameter(torch.tensor(a, dtype=torch.float32, device=device).requires_grad_(True))
    PT = lambda a: torch.nn.Parameter(torch.tensor(a, dtype=torch.float32, device=device).transpose(1, 2).contiguous().requires_grad_(True))
    for side, mask in (("scene", scene_mask), ("obj", obj_mask)):
        setattr(model, "_%s_xyz" % side, P(xyz[mask]))
        setattr(model, "_%s_shs_dc" % side, PT(shs_dc[mask]))
        setattr(model, "_%s_shs_rest" % side, PT(shs_extra[mask]))
        setattr(model, "_%s_opacity" % side, P(opacities[mask]))
        setattr(model, "_%s_scaling" % side, P(scales[mask]))
        setattr(model, "_%s_rotation" % side, P(rots[mask]))
    (xyz_dp, rot_dp, shs_s, shs_o, bg_dp, gs_time, gs_time_sigma, model.use_time_mask, model.order_args, model.scene_extent) = torch.load(
        os.path.join(os.path.dirname(os.path.abspath(path)), "deform.pth"), map_location=device, weights_only=False)
    n_par = lambda a: a[0] + a[2] + 2 * a[3] + a[4]
    assert xyz_dp.shape[0] == model._obj_xyz.shape[0]
    assert xyz_dp.shape[-1] == n_par(model.order_args['xyz']) and rot_dp.shape[-1] == n_par(model.order_args['rotation'])
    assert shs_o.shape[-1] == n_par(model.order_args['shs']) and shs_s.shape[-1] == n_par(model.order_args['shs'])
    assert bg_dp.shape[-1] == n_par(model.order_args['background'])
    G = lambda t: torch.nn.Parameter(t.detach().to(device).requires_grad_(True))
    model.xyz_deform_param, model.rotation_deform_param = G(xyz_dp), G(rot_dp)
    model.shs_deform_param_scene, model.shs_deform_param_obj, model.background_deform_param = G(shs_s), G(shs_o), G(bg_dp)
    model.gs_time = gs_time.to(device)
    model.gs_time_sigma = G(gs_time_sigma)
    model.active_sh_degree = model.max_sh_degree
    return model
