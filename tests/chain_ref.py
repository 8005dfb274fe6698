"""Oracle CHAIN for the whole hot path (test infrastructure):

    raw GaussianModel parameters --deform_oracle (NumPy f32)--> activated tensors
        --raster_oracle (C++ f32 / f64)--> images + dL/d(activated tensors)
        --float64 torch restatement of the deformation (tests/torch_deform_ref.py) carries the chain rule-->
    gradients of every RAW parameter

which is what `gaussian_renderer.render()` + `loss.backward()` compute in the reference
(gaussian_renderer/__init__.py:57-94, scene/gaussian_model.py:173-231).  With an environment map the composite
`render = fg + (1 - O) * bg` (gaussian_renderer/__init__.py:93-94) and oracle/env_oracle.py are part of the chain.
"""
import numpy as np
import torch

from oracle import deform_oracle as do
from oracle import oracle
from tests import torch_deform_ref as tr

RAW_NAMES = ["scene_xyz", "obj_xyz", "scene_shs_dc", "obj_shs_dc", "scene_shs_rest", "obj_shs_rest", "scene_scaling", "obj_scaling",
             "scene_rotation", "obj_rotation", "scene_opacity", "obj_opacity", "xyz_deform_param", "rotation_deform_param",
             "shs_deform_param_scene", "shs_deform_param_obj", "background_deform_param", "gs_time_sigma", "gs_time"]


def attr_of(name):
    return name if name.endswith("deform_param") or name.startswith("shs_deform") or name.startswith("gs_") else "_" + name


def raw_numpy(model):
    """{reference attribute name without the leading underscore: float32 array} of a SyntheticGaussianModel-like object."""
    return {n: getattr(model, attr_of(n)).detach().cpu().numpy().astype(np.float32) for n in RAW_NAMES}


def deformed_xyz64(m64, t, oa):
    """scene/gaussian_model.py:173-185 in float64 torch (differentiable)."""
    obj = m64["obj_xyz"] + tr.get_func_result(t, m64["xyz_deform_param"], oa["xyz"])
    return torch.cat([m64["scene_xyz"], obj], 0) + tr.get_func_result(t, m64["background_deform_param"], oa["background"])


def deformed_xyz32(raw, t, oa):
    obj = raw["obj_xyz"] + do.get_func_result(t, raw["xyz_deform_param"], oa["xyz"])
    xyz = np.concatenate([raw["scene_xyz"], np.asarray(obj, np.float32)], 0) + do.get_func_result(t, raw["background_deform_param"], oa["background"])
    return np.asarray(xyz, np.float32)


def run_chain(raw, oa, use_time_mask, t, t_flow, cam, H, W, degree, ups, semantic=None, precision="f32", env=None, inv_depth=True, strict=False, strict_mask=None):
    """raw: raw_numpy(model).  ups: dict of upstream image gradients (numpy): 'color' (or 'render' with env), 'depth', 'img_opacity',
    and with t_flow / semantic 'flow' / 'semantic' -- or a callable(images dict) -> such a dict (a loss evaluated on the chain's own images).
    env: None or dict(grid_map [C,Hm,Wm], focal, R [3,3]).
    Returns dict(images..., radii, act (activated f32 tensors), act_grads, raw_grads {name: float64 array}, env_grad, explained (the
    oracle's gate-flip masks, tests/parity.py)); with strict=True also raw_grads_strict / act_grads_strict / env_grad_strict: the same
    backward with every upstream gradient zeroed at the gate-flip pixels (the strict gradient pass of tests/parity.py); strict_mask: use
    this [H, W] pixel mask instead of the run's own (the conditioning draws of a failed strict comparison: float64 run, perturbed run)."""
    from tests import parity
    npm = dict(raw)
    npm["order_args"], npm["use_time_mask"] = oa, use_time_mask
    act = do.get_deformed_pkg(npm, t)
    flow = deformed_xyz32(raw, t_flow, oa) if t_flow is not None else None
    o = oracle.RasterOracle(precision)
    fwd = o.forward(np.zeros(3, np.float32), act["xyz"], None, act["opacity"], act["scales"], act["rotation"], 1.0, None, cam["viewmatrix"],
                    cam["projmatrix"], cam["tanfovx"], cam["tanfovy"], H, W, act["shs"], flow, semantic, degree, cam["campos"], False, inv_depth)
    z = lambda c: np.zeros((c, H, W), np.float32)
    out = dict(fwd)
    strict = strict or strict_mask is not None
    out["explained"] = parity.explained_masks(o.gate_margins()) if strict else None
    bg = None
    O = np.asarray(fwd["img_opacity"], np.float64)
    if env is not None:
        from oracle import env_oracle
        bg = env_oracle.background(env["grid_map"], H, W, env["focal"], env["R"])
        out["background"] = bg
        out["render"] = np.asarray(fwd["color"], np.float64) + (1.0 - O) * bg
    if callable(ups):
        ups = ups(out)
    out["ups"] = ups
    # chain rule through the deformation in float64
    m64 = {k: torch.tensor(np.asarray(v, np.float64), dtype=torch.float64, requires_grad=(k != "gs_time")) for k, v in raw.items()}
    pkg = tr.get_deformed_pkg(m64, t, oa, use_time_mask)
    P = act["xyz"].shape[0]
    outs = [pkg["xyz"], pkg["rotation"], pkg["shs"], pkg["opacity"], pkg["scales"]]
    if t_flow is not None:
        outs.append(deformed_xyz64(m64, t_flow, oa))
    T64 = lambda a, shape: torch.tensor(np.asarray(a, np.float64).reshape(shape), dtype=torch.float64)

    def backward(ups, retain):
        g_color = np.asarray(ups["render"] if env is not None else ups["color"], np.float32)
        g_op = np.asarray(ups["img_opacity"], np.float32).reshape(1, H, W).copy()
        env_grad = None
        if env is not None:
            from oracle import env_oracle
            g_op = (g_op.astype(np.float64) - (g_color.astype(np.float64) * bg).sum(0, keepdims=True)).astype(np.float32)
            env_grad = env_oracle.background_grad(env["grid_map"], H, W, env["focal"], env["R"], (1.0 - O) * g_color.astype(np.float64))
        bw = o.backward(g_color, ups["depth"], ups["flow"] if t_flow is not None else z(3), ups["semantic"] if semantic is not None else None, g_op)
        gr = [T64(bw["dL_dmeans3D"], (P, 3)), T64(bw["dL_drotations"], (P, 4)), T64(bw["dL_dsh"], tuple(pkg["shs"].shape)),
              T64(bw["dL_dopacity"], (P, 1)), T64(bw["dL_dscales"], (P, 3))]
        if t_flow is not None:
            gr.append(T64(bw["dL_dflow_points"], (P, 3)))
        for v in m64.values():
            v.grad = None
        torch.autograd.backward(outs, gr, retain_graph=retain)
        raw_grads = {k: (None if v.grad is None else v.grad.numpy().copy()) for k, v in m64.items() if k != "gs_time"}
        return bw, raw_grads, env_grad

    bw, raw_grads, env_grad = backward(ups, strict)
    out.update(act=act, flow_points=flow, act_grads=bw, raw_grads=raw_grads, env_grad=env_grad)
    if strict:
        bw_s, raw_s, env_s = backward(parity.mask_upstream(ups, out["explained"]["pixel"] if strict_mask is None else strict_mask), False)
        out.update(act_grads_strict=bw_s, raw_grads_strict=raw_s, env_grad_strict=env_s)
    return out
