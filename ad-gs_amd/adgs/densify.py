"""Densification on the HIP path: drop-ins for the reference GaussianModel's
densify_and_prune / reset_opacity (scene/gaussian_model.py:465-469, 545-861).

`model` is any object with the reference GaussianModel's attributes: the raw parameter tensors (_scene_xyz ...
gs_time_sigma), `optimizer` (torch.optim.Adam or adgs.optim.FusedAdam with the reference's group names,
:346-372), `gs_time`, the statistics `xyz_gradient_accum` [N,1], `denom` [N,1], `max_radii2D` [N], and
`percent_dense`, `scene_extent`, `object_extent`.  The result -- row order, parameter values, Adam moments -- is the
reference's; the work is one row map per side and one gather per tensor (include/adgs_densify.h) instead of three rounds
of cat / boolean-mask surgery over 17 tensors x 3.  Two host read-backs remain (the counts that size the normal samples
and the new tensors); the reference has a dozen.  There is no CPU path.
"""
import ctypes

import torch

from . import _lib

# optimizer group name -> GaussianModel attribute (scene/gaussian_model.py:346-372)
GROUP_ATTR = {
    "scene_xyz": "_scene_xyz", "scene_shs_dc": "_scene_shs_dc", "scene_shs_rest": "_scene_shs_rest", "scene_opacity": "_scene_opacity",
    "scene_scaling": "_scene_scaling", "scene_rotation": "_scene_rotation", "deform_shs_scene": "shs_deform_param_scene",
    "obj_xyz": "_obj_xyz", "obj_shs_dc": "_obj_shs_dc", "obj_shs_rest": "_obj_shs_rest", "obj_opacity": "_obj_opacity",
    "obj_scaling": "_obj_scaling", "obj_rotation": "_obj_rotation", "deform_xyz": "xyz_deform_param", "deform_rotation": "rotation_deform_param",
    "deform_shs_obj": "shs_deform_param_obj", "time_sigma": "gs_time_sigma", "deform_background": "background_deform_param"}
SCENE_GROUPS = ["scene_xyz", "scene_shs_dc", "scene_shs_rest", "scene_opacity", "scene_scaling", "scene_rotation", "deform_shs_scene"]
OBJ_GROUPS = ["obj_xyz", "obj_shs_dc", "obj_shs_rest", "obj_opacity", "obj_scaling", "obj_rotation", "deform_xyz", "deform_rotation",
              "deform_shs_obj", "time_sigma"]


class DensifySide(ctypes.Structure):
    """adgs_densify_side (include/adgs_densify.h)."""
    _fields_ = [("N", ctypes.c_int32), ("grad_accum", ctypes.c_void_p), ("denom", ctypes.c_void_p), ("scaling", ctypes.c_void_p),
                ("opacity", ctypes.c_void_p), ("grad_threshold", ctypes.c_float), ("dense_extent", ctypes.c_float), ("min_opacity", ctypes.c_float),
                ("big_extent", ctypes.c_float), ("prune_big", ctypes.c_int32)]


def _stream(dev):
    return _lib.stream_ptr(dev)


def _ptr(t):
    return None if (t is None or t.numel() == 0) else t.data_ptr()


def _f32(x):
    """A python float rounded like torch rounds a scalar operand of a float32 tensor."""
    return float(torch.tensor(float(x), dtype=torch.float32))


class _Plan:
    """Row map of one side: result row i comes from source row row_src[i]; row_aux tells what kind of row it is."""

    def __init__(self, n_out, row_src, row_aux, samples):
        self.n_out, self.row_src, self.row_aux, self.samples = n_out, row_src, row_aux, samples

    def gather(self, src, is_state=False):
        src = src.detach()
        if not src.is_contiguous():
            src = src.contiguous()
        dst = torch.empty((self.n_out,) + tuple(src.shape[1:]), dtype=torch.float32, device=src.device)
        L = int(src[0].numel()) if src.shape[0] else int(torch.Size(src.shape[1:]).numel())
        if self.n_out and L:
            with torch.cuda.device(src.device):
                _lib.check(_lib.lib().adgs_densify_gather_rows(_ptr(src), _ptr(dst), L, self.n_out, _ptr(self.row_src), _ptr(self.row_aux), int(is_state),
                                                               _stream(src.device)), "adgs_densify_gather_rows")
        return dst


def _groups_by_name(optimizer):
    return {g["name"]: g for g in optimizer.param_groups}


def densify_and_prune(model, max_scene_grad, max_obj_grad, min_opacity, prune_big_points):
    """GaussianModel.densify_and_prune (scene/gaussian_model.py:835-861) without its trailing set_obj_near_idx()."""
    lib = _lib.lib()
    dev = model._scene_xyz.device if model._scene_xyz.numel() else model._obj_xyz.device
    if dev.type != "cuda":
        raise RuntimeError("densify_and_prune: the model must live on a HIP device; there is no CPU path")
    Ns, No = model._scene_xyz.shape[0], model._obj_xyz.shape[0]
    accum, denom = model.xyz_gradient_accum.contiguous().view(-1), model.denom.contiguous().view(-1)
    if accum.numel() != Ns + No or denom.numel() != Ns + No:
        raise ValueError("densify_and_prune: statistics do not match the number of Gaussians")
    i32 = dict(dtype=torch.int32, device=dev)
    sides = {}
    counts = torch.zeros(4, **i32)
    with torch.cuda.device(dev):
        for k, (name, N, off, thr, extent, big) in enumerate((("scene", Ns, 0, max_scene_grad, model.scene_extent, 0.05),
                                                              ("obj", No, Ns, max_obj_grad, model.object_extent, 0.1))):
            scaling = getattr(model, "_%s_scaling" % name).detach().contiguous()
            opacity = getattr(model, "_%s_opacity" % name).detach().contiguous()
            s = DensifySide()
            s.N = N
            s.grad_accum, s.denom = _ptr(accum[off:off + N]), _ptr(denom[off:off + N])
            s.scaling, s.opacity = _ptr(scaling), _ptr(opacity)
            s.grad_threshold, s.dense_extent = _f32(thr), _f32(extent * model.percent_dense)
            s.min_opacity, s.big_extent, s.prune_big = _f32(min_opacity), _f32(extent * big), int(bool(prune_big_points))
            ci, si = torch.empty(max(N, 1), **i32), torch.empty(max(N, 1), **i32)
            ws = torch.empty(lib.adgs_densify_workspace_bytes(N), dtype=torch.uint8, device=dev)
            _lib.check(lib.adgs_densify_select(ctypes.byref(s), _ptr(ci), _ptr(si), counts[2 * k:].data_ptr(), _ptr(ws), _stream(dev)), "adgs_densify_select")
            sides[name] = dict(struct=s, N=N, clone_index=ci, split_index=si, scaling=scaling, opacity=opacity, keep=(accum, denom))
        n = counts.tolist()                                   # read-back 1: sizes of the normal samples
        out_counts = torch.zeros(2, **i32)
        for k, name in enumerate(("scene", "obj")):
            sd = sides[name]
            sd["n_clone"], sd["n_split"] = int(n[2 * k]), int(n[2 * k + 1])
            # the reference's own draw (:719-720, :730-731), scene first: same shapes, same generator consumption
            stds = torch.exp(sd["scaling"])[sd["split_index"][:sd["n_split"]].long()].repeat(2, 1)
            sd["samples"] = torch.normal(mean=0.0, std=stds).contiguous()
        for k, name in enumerate(("scene", "obj")):
            sd = sides[name]
            total = sd["N"] + sd["n_clone"] + 2 * sd["n_split"]
            sd["row_src"], sd["row_aux"] = torch.empty(max(total, 1), **i32), torch.empty(max(total, 1), **i32)
            ws = torch.empty(lib.adgs_densify_workspace_bytes(total), dtype=torch.uint8, device=dev)
            _lib.check(lib.adgs_densify_plan(ctypes.byref(sd["struct"]), _ptr(sd["clone_index"]), sd["n_clone"], _ptr(sd["split_index"]), sd["n_split"],
                                             _ptr(sd["row_src"]), _ptr(sd["row_aux"]), out_counts[k:].data_ptr(), _ptr(ws), _stream(dev)), "adgs_densify_plan")
        m = out_counts.tolist()                               # read-back 2: sizes of the new tensors
        plans = {name: _Plan(int(m[k]), sides[name]["row_src"], sides[name]["row_aux"], sides[name]["samples"]) for k, name in enumerate(("scene", "obj"))}
        _apply_plans(model, plans, dev)
    return dict(scene=(sides["scene"]["n_clone"], sides["scene"]["n_split"], plans["scene"].n_out),
                obj=(sides["obj"]["n_clone"], sides["obj"]["n_split"], plans["obj"].n_out))


def _apply_plans(model, plans, dev):
    lib = _lib.lib()
    groups = _groups_by_name(model.optimizer)
    new_tensors = {}
    for side, names in (("scene", SCENE_GROUPS), ("obj", OBJ_GROUPS)):
        plan = plans[side]
        for name in names:
            attr = GROUP_ATTR[name]
            old = getattr(model, attr)
            new_tensors[name] = plan.gather(old)
        # split children: sampled position, shrunk scale (:721-723)
        if plan.n_out:
            xyz, sc, rot = (getattr(model, "_%s_%s" % (side, k)).detach().contiguous() for k in ("xyz", "scaling", "rotation"))
            _lib.check(lib.adgs_densify_split_rows(_ptr(xyz), _ptr(sc), _ptr(rot), _ptr(plan.samples), plan.n_out, _ptr(plan.row_src), _ptr(plan.row_aux),
                                                   _ptr(new_tensors[side + "_xyz"]), _ptr(new_tensors[side + "_scaling"]), _stream(dev)),
                       "adgs_densify_split_rows")
        for name in names:
            group = groups.get(name)
            old = getattr(model, GROUP_ATTR[name])
            param = torch.nn.Parameter(new_tensors[name].requires_grad_(True))
            if group is not None:
                assert len(group["params"]) == 1
                p_old = group["params"][0]
                state = model.optimizer.state.get(p_old, None)
                if state is not None and "exp_avg" in state:             # cat_tensors_to_optimizer / _prune_optimizer (:561-635)
                    state["exp_avg"] = plan.gather(state["exp_avg"], is_state=True)
                    state["exp_avg_sq"] = plan.gather(state["exp_avg_sq"], is_state=True)
                    del model.optimizer.state[p_old]
                    model.optimizer.state[param] = state
                elif state is not None and p_old in model.optimizer.state:
                    del model.optimizer.state[p_old]
                group["params"][0] = param
            setattr(model, GROUP_ATTR[name], param)
    if getattr(model, "gs_time", None) is not None and torch.is_tensor(model.gs_time):
        model.gs_time = plans["obj"].gather(model.gs_time)
    N = plans["scene"].n_out + plans["obj"].n_out
    f32 = dict(dtype=torch.float32, device=dev)
    model.xyz_gradient_accum = torch.zeros((N, 1), **f32)         # densification_postfix (:699-702), then masked: still zeros
    model.denom = torch.zeros((N, 1), **f32)
    model.max_radii2D = torch.zeros((N,), **f32)


def reset_opacity(model):
    """GaussianModel.reset_opacity (:465-469): opacity = inverse_sigmoid(min(sigmoid(opacity), 0.01)) as one in-place kernel per
    side; the Adam moments of the two opacity groups are zeroed (replace_tensor_to_optimizer, :546-559)."""
    lib = _lib.lib()
    groups = _groups_by_name(model.optimizer) if getattr(model, "optimizer", None) is not None else {}
    for name in ("scene_opacity", "obj_opacity"):
        attr = GROUP_ATTR[name]
        old = getattr(model, attr)
        if not old.is_cuda:
            raise RuntimeError("reset_opacity: the model must live on a HIP device; there is no CPU path")
        new = old.detach().clone().contiguous()
        with torch.cuda.device(new.device):
            _lib.check(lib.adgs_reset_opacity(new.numel(), _ptr(new), _stream(new.device)), "adgs_reset_opacity")
        param = torch.nn.Parameter(new.requires_grad_(True))
        group = groups.get(name)
        if group is not None:
            p_old = group["params"][0]
            state = model.optimizer.state.get(p_old, None)
            if state is not None and "exp_avg" in state:
                state["exp_avg"] = torch.zeros_like(new); state["exp_avg_sq"] = torch.zeros_like(new)
                del model.optimizer.state[p_old]
                model.optimizer.state[param] = state
            group["params"][0] = param
        setattr(model, attr, param)
