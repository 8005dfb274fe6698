"""Per-frame deformation on the HIP path: drop-in `get_func_result` and a fused
`get_deformed_pkg` for objects that carry the reference GaussianModel's raw parameters.

Mirrors utils/func_utils.py:33-173 and scene/gaussian_model.py:88-231 of the reference.
The time-dependent basis values are identical for all Gaussians; they are evaluated here on
the host once per frame with the same float32 torch ops the reference uses, and handed to
the HIP kernels (adgs_func_eval_* / adgs_deform_* in include/adgs_deform.h) that do the
per-Gaussian work.  Both directions are hand-written HIP; there is no PyTorch fallback.
"""
import ctypes
import functools
import math

import numpy as np
import torch

from . import _lib

MAX_TERMS = 48
MAX_QUAT = 8


class FuncEval(ctypes.Structure):
    _fields_ = [("n_params", ctypes.c_int32), ("n_terms", ctypes.c_int32 * 3), ("index", ctypes.c_int32 * MAX_TERMS),
                ("weight", ctypes.c_float * MAX_TERMS), ("quat_start", ctypes.c_int32), ("quat_k", ctypes.c_int32),
                ("quat_cum", ctypes.c_float * MAX_QUAT)]


_PTRS = ["scene_xyz", "obj_xyz", "scene_rotation", "obj_rotation", "scene_shs_dc", "obj_shs_dc", "scene_shs_rest", "obj_shs_rest",
         "scene_opacity", "obj_opacity", "scene_scaling", "obj_scaling", "xyz_deform_param", "rotation_deform_param",
         "shs_deform_param_scene", "shs_deform_param_obj", "background_deform_param", "gs_time", "gs_time_sigma"]


class DeformParams(ctypes.Structure):
    _fields_ = [("Ns", ctypes.c_int32), ("No", ctypes.c_int32), ("sh_coeffs", ctypes.c_int32), ("use_time_mask", ctypes.c_int32),
                ("t", ctypes.c_float)] + [(n, ctypes.c_void_p) for n in _PTRS]


class DeformOutputs(ctypes.Structure):
    _fields_ = [(n, ctypes.c_void_p) for n in ("xyz", "rotation", "shs", "opacity", "scales")]


_GRADS = ["scene_xyz", "obj_xyz", "scene_rotation", "obj_rotation", "scene_shs_dc", "obj_shs_dc", "scene_shs_rest", "obj_shs_rest",
          "scene_opacity", "obj_opacity", "scene_scaling", "obj_scaling", "xyz_deform_param", "rotation_deform_param",
          "shs_deform_param_scene", "shs_deform_param_obj", "background_deform_param", "gs_time_sigma"]


class DeformGrads(ctypes.Structure):
    _fields_ = [(n, ctypes.c_void_p) for n in _GRADS]


_M_CACHE = {}


def get_deboor_cox_mat(order):
    """De Boor-Cox basis matrix of order k, (k+1)x(k+1) float32 (reference utils/func_utils.py:33-50)."""
    if order in _M_CACHE:
        return _M_CACHE[order]
    if order == 0:
        m = np.array([[1.0]], dtype=np.float32)
    else:
        prior = get_deboor_cox_mat(order - 1)
        zrow = np.zeros((1, prior.shape[1]), dtype=np.float32)
        lo, hi = np.concatenate([prior, zrow], 0), np.concatenate([zrow, prior], 0)
        i = np.arange(order, dtype=np.int32)
        a = np.zeros((order, order + 1), dtype=np.float32)
        a[i, i] = i + 1
        a[i, i + 1] = order - i - 1
        b = np.zeros((order, order + 1), dtype=np.float32)
        b[i, i] = -1
        b[i, i + 1] = 1
        m = (lo @ a + hi @ b) / order
    _M_CACHE[order] = m
    return m


def get_param_num(args):
    return args[0] + args[2] + 2 * args[3] + args[4]


def _bspline_basis(u, order):
    freq = torch.arange(0.0, order + 1.0, 1.0, dtype=torch.float32)
    return (u ** freq) @ torch.tensor(get_deboor_cox_mat(order), dtype=torch.float32)


def make_func_eval(v, order_args, n_params=None):
    """Host-side evaluation of every basis value of get_func_result(v, ., order_args).  Memoised: a training run revisits
    the same camera time stamps, and the evaluation is a handful of tiny CPU torch ops that have no place in the per-frame
    launch path (the returned struct is shared and must not be modified)."""
    return _make_func_eval_cached(float(v), tuple(int(a) for a in order_args), None if n_params is None else int(n_params))


@functools.lru_cache(maxsize=8192)
def _make_func_eval_cached(v, order_args, n_params):
    oa = list(order_args)
    f = FuncEval()
    f.n_params = int(get_param_num(oa) if n_params is None else n_params)
    f.quat_start, f.quat_k = -1, 0
    idx, wts, counts, offset = [], [], [0, 0, 0], 0
    if oa[0] != 0:
        interval = oa[0] - oa[1]
        start = min(int(v * interval), interval - 1)
        u = v * interval - start
        b = _bspline_basis(u, oa[1])
        idx += [start + offset + j for j in range(oa[1] + 1)]
        wts += b.tolist()
        counts[0] = oa[1] + 1
        offset += oa[0]
    if oa[2] != 0:
        freq = torch.linspace(1.0, oa[2], oa[2], dtype=torch.float32)
        idx += [offset + j for j in range(oa[2])]
        wts += (v ** freq).tolist()
        counts[1] = oa[2]
        offset += oa[2]
    if oa[3] != 0:
        freq = torch.linspace(1.0, oa[3], oa[3], dtype=torch.float32) * np.pi
        idx += [offset + j for j in range(2 * oa[3])]
        wts += torch.cat([torch.sin(v * freq), torch.cos(v * freq)]).tolist()
        counts[2] = 2 * oa[3]
        offset += 2 * oa[3]
    if oa[4] != 0:
        interval = oa[4] - oa[5]
        start = min(int(v * interval), interval - 1)
        u = v * interval - start
        b = _bspline_basis(u, oa[5])
        cum = torch.flip(torch.cumsum(torch.flip(b, dims=(-1,)), dim=-1), dims=(-1,))[1:]
        if oa[5] + 1 > MAX_QUAT:
            raise ValueError("quaternion spline order %d not supported (max %d)" % (oa[5], MAX_QUAT - 1))
        f.quat_start, f.quat_k = start + offset, oa[5]
        for j, c in enumerate(cum.tolist()):
            f.quat_cum[j] = c
    if len(idx) > MAX_TERMS:
        raise ValueError("too many basis terms (%d > %d)" % (len(idx), MAX_TERMS))
    for j, (i, w) in enumerate(zip(idx, wts)):
        f.index[j] = int(i)
        f.weight[j] = float(w)
    for j in range(3):
        f.n_terms[j] = counts[j]
    return f


def _stream(dev):
    return _lib.stream_ptr(dev)


def _dp(t):
    return None if (t is None or t.numel() == 0) else t.data_ptr()


class _FuncEvalFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, param, feval):
        if not param.is_cuda:
            raise RuntimeError("get_func_result: param must be on a HIP device; there is no CPU path")
        p = param.contiguous().float()
        lead, D = p.shape[:-2], p.shape[-2]
        N = int(np.prod(lead)) if len(lead) else 1
        out = torch.empty(*lead, D, dtype=torch.float32, device=p.device)
        with _lib.on_device(p.device):
            _lib.check(_lib.lib().adgs_func_eval_forward(N, D, p.data_ptr(), ctypes.byref(feval), out.data_ptr(), _stream(p.device)),
                       "adgs_func_eval_forward")
        ctx.save_for_backward(p)
        ctx.feval, ctx.N, ctx.D = feval, N, D
        return out

    @staticmethod
    def backward(ctx, g):
        (p,) = ctx.saved_tensors
        gp = torch.zeros_like(p)
        g = g.contiguous().float()
        with _lib.on_device(p.device):
            _lib.check(_lib.lib().adgs_func_eval_backward(ctx.N, ctx.D, p.data_ptr(), ctypes.byref(ctx.feval), g.data_ptr(), gp.data_ptr(),
                                                          _stream(p.device)), "adgs_func_eval_backward")
        return gp, None


def get_func_result(v, param, order_args):
    """Drop-in for utils.func_utils.get_func_result (reference :121-173): returns the python float
    0.0 when every order is zero, else a tensor [..., D] differentiable w.r.t. `param`."""
    if all(int(a) == 0 for a in order_args):
        return 0.0
    return _FuncEvalFn.apply(param, make_func_eval(float(v), order_args, param.shape[-1]))


# ------------------------------------------------------------------ fused get_deformed_pkg
_MODEL_ATTRS = {  # ctypes field -> reference GaussianModel attribute (scene/gaussian_model.py:47-84)
    "scene_xyz": "_scene_xyz", "obj_xyz": "_obj_xyz", "scene_rotation": "_scene_rotation", "obj_rotation": "_obj_rotation",
    "scene_shs_dc": "_scene_shs_dc", "obj_shs_dc": "_obj_shs_dc", "scene_shs_rest": "_scene_shs_rest", "obj_shs_rest": "_obj_shs_rest",
    "scene_opacity": "_scene_opacity", "obj_opacity": "_obj_opacity", "scene_scaling": "_scene_scaling", "obj_scaling": "_obj_scaling",
    "xyz_deform_param": "xyz_deform_param", "rotation_deform_param": "rotation_deform_param",
    "shs_deform_param_scene": "shs_deform_param_scene", "shs_deform_param_obj": "shs_deform_param_obj",
    "background_deform_param": "background_deform_param", "gs_time": "gs_time", "gs_time_sigma": "gs_time_sigma"}


class _DeformPkgFn(torch.autograd.Function):
    """19 raw parameter tensors -> (xyz, rotation, shs, opacity, scales, flow_xyz)."""

    @staticmethod
    def forward(ctx, meta, *tensors):
        t, order_args, use_time_mask, want, flow_t = meta[:5]
        skip_scene = bool(meta[6]) if len(meta) > 6 else False
        ts = [None if x is None else x.contiguous() for x in tensors]
        named = dict(zip(_PTRS, ts))
        dev = named["scene_xyz"].device
        if dev.type != "cuda":
            raise RuntimeError("get_deformed_pkg: parameters must be on a HIP device; there is no CPU path")
        Ns, No = named["scene_xyz"].shape[0], named["obj_xyz"].shape[0]
        N = Ns + No
        M = 1 + named["scene_shs_rest"].shape[1]
        p = DeformParams()
        p.Ns, p.No, p.sh_coeffs, p.use_time_mask, p.t = Ns, No, M, int(bool(use_time_mask)) | (2 if skip_scene else 0), float(t)      # bit 1: ADGS_DEFORM_SKIP_SCENE
        for n in _PTRS:
            setattr(p, n, _dp(named[n]))
        # n_params is the parameter tensor's last dimension even when it has no rows (a model without object Gaussians)
        fe = {k: make_func_eval(float(t), order_args[k], named[pn].shape[-1] if named[pn] is not None and named[pn].dim() >= 1 else 0)
              for k, pn in (("xyz", "xyz_deform_param"), ("rotation", "rotation_deform_param"), ("shs", "shs_deform_param_scene"),
                            ("background", "background_deform_param"))}
        if flow_t is not None:
            for k, pn in (("xyz", "xyz_deform_param"), ("background", "background_deform_param")):
                fe[k + "_flow"] = make_func_eval(float(flow_t), order_args[k], fe[k].n_params)
        f32 = dict(dtype=torch.float32, device=dev)
        flow_xyz = torch.empty(N, 3, **f32) if flow_t is not None else None
        outs = dict(xyz=torch.empty(N, 3, **f32) if "xyz" in want else None,
                    rotation=torch.empty(N, 4, **f32) if "rotation" in want else None,
                    shs=torch.empty(N, M, 3, **f32) if "shs" in want else None,
                    opacity=torch.empty(N, 1, **f32) if "opacity" in want else None,
                    scales=torch.empty(N, 3, **f32) if "scales" in want else None)
        o = DeformOutputs()
        for k, v in outs.items():
            setattr(o, k, _dp(v))
        with _lib.on_device(dev):
            _lib.check(_lib.lib().adgs_deform_forward_flow(
                ctypes.byref(p), ctypes.byref(fe["xyz"]), ctypes.byref(fe["rotation"]), ctypes.byref(fe["shs"]), ctypes.byref(fe["background"]),
                ctypes.byref(fe["xyz_flow"]) if flow_t is not None else None, ctypes.byref(fe["background_flow"]) if flow_t is not None else None,
                ctypes.byref(o), _dp(flow_xyz), _stream(dev)), "adgs_deform_forward_flow")
        ctx.save_for_backward(*[x for x in ts if x is not None])
        ctx.present = [x is not None for x in ts]
        ctx.meta, ctx.fe, ctx.dims = meta, fe, (Ns, No, M)
        order = ("xyz", "rotation", "shs", "opacity", "scales")
        ctx.want = want
        return tuple(outs[k] if outs[k] is not None else torch.empty(0, **f32) for k in order) + (
            flow_xyz if flow_xyz is not None else torch.empty(0, **f32),)

    @staticmethod
    def backward(ctx, g_xyz, g_rot, g_shs, g_op, g_sc, g_flow):
        t, order_args, use_time_mask, want, flow_t = ctx.meta[:5]
        arena = ctx.meta[5] if len(ctx.meta) > 5 else None
        skip_scene = bool(ctx.meta[6]) if len(ctx.meta) > 6 else False
        saved = list(ctx.saved_tensors)
        ts = [saved.pop(0) if pr else None for pr in ctx.present]
        named = dict(zip(_PTRS, ts))
        dev = named["scene_xyz"].device
        Ns, No, M = ctx.dims
        p = DeformParams()
        p.Ns, p.No, p.sh_coeffs, p.use_time_mask, p.t = Ns, No, M, int(bool(use_time_mask)) | (2 if skip_scene else 0), float(t)
        for n in _PTRS:
            setattr(p, n, _dp(named[n]))
        up = {}
        for k, g in (("xyz", g_xyz), ("rotation", g_rot), ("shs", g_shs), ("opacity", g_op), ("scales", g_sc)):
            up[k] = g.contiguous().float() if (k in want and g is not None and g.numel() > 0) else None
        up["flow"] = g_flow.contiguous().float() if (flow_t is not None and g_flow is not None and g_flow.numel() > 0) else None
        # which output each raw parameter feeds; a parameter whose output has no upstream gradient gets
        # no gradient tensor at all, the others are fully written by the kernels (no zero fill) except
        # the atomically accumulated background row
        dep = {"scene_xyz": "xyz", "obj_xyz": "xyz", "xyz_deform_param": "xyz", "background_deform_param": "xyz",
               "scene_rotation": "rotation", "obj_rotation": "rotation", "rotation_deform_param": "rotation",
               "scene_shs_dc": "shs", "obj_shs_dc": "shs", "scene_shs_rest": "shs", "obj_shs_rest": "shs",
               "shs_deform_param_scene": "shs", "shs_deform_param_obj": "shs",
               "scene_opacity": "opacity", "obj_opacity": "opacity", "gs_time_sigma": "opacity",
               "scene_scaling": "scales", "obj_scaling": "scales"}
        # factored exchange (adgs.dp.FactoredSHExchange): xyz_deform_param's gradient is w(t) (x) g_xyz + w(t_flow) (x) g_flow of the
        # object range -- the two upstream tensors go to the exchange and the [No,3,Cx] rows are not materialised here
        xyz_factored = False
        sink = getattr(arena, "xyz_sink", None) if arena is not None else None
        if sink is not None and named["xyz_deform_param"] is not None and named["xyz_deform_param"].numel() > 0 and \
                ctx.needs_input_grad[1 + _PTRS.index("xyz_deform_param")] and (up["xyz"] is not None or up["flow"] is not None):
            xyz_factored = bool(sink(None if up["xyz"] is None else up["xyz"][Ns:], None if up["flow"] is None else up["flow"][Ns:]))
        grads, gs = {}, DeformGrads()
        for n in _GRADS:
            src = named[n]
            has_up = up[dep[n]] is not None or (dep[n] == "xyz" and up["flow"] is not None)
            need = (src is not None and src.numel() > 0 and ctx.needs_input_grad[1 + _PTRS.index(n)] and has_up)
            if skip_scene and n in ("scene_xyz", "scene_rotation", "scene_opacity", "scene_scaling"):
                need = False              # these gradients come from the rasterizer's backward (raw scene geometry)
            if not need or (xyz_factored and n == "xyz_deform_param"):
                grads[n] = None
            elif n == "background_deform_param":          # accumulated with atomics
                grads[n] = torch.zeros_like(src)
            else:
                # a gradient arena (adgs.dp.GradArena) hands out slices of ONE persistent flat buffer, so that the data-parallel
                # all-reduce runs on that buffer directly instead of on a concatenated copy of these tensors
                g = arena.take(n, src) if arena is not None else None
                grads[n] = g if g is not None else torch.empty_like(src)
            setattr(gs, n, _dp(grads[n]))
        fe = ctx.fe
        with _lib.on_device(dev):
            _lib.check(_lib.lib().adgs_deform_backward_flow(
                ctypes.byref(p), ctypes.byref(fe["xyz"]), ctypes.byref(fe["rotation"]), ctypes.byref(fe["shs"]), ctypes.byref(fe["background"]),
                ctypes.byref(fe["xyz_flow"]) if flow_t is not None else None, ctypes.byref(fe["background_flow"]) if flow_t is not None else None,
                _dp(up["xyz"]), _dp(up["rotation"]), _dp(up["shs"]), _dp(up["opacity"]), _dp(up["scales"]), _dp(up["flow"]),
                ctypes.byref(gs), _stream(dev)), "adgs_deform_backward_flow")
        return (None,) + tuple(grads.get(n) for n in _PTRS)


def raw_scene_ok(model):
    """The raw-scene path (the rasterizer's preprocess applies exp / normalize / sigmoid to the raw scene tensors itself) needs a
    scene range that the deformation leaves alone: no background deformation (scene/gaussian_model.py:183), at least one scene
    Gaussian."""
    bg = model.order_args.get("background", [0] * 6)
    return all(int(a) == 0 for a in bg) and model._scene_xyz.shape[0] > 0 and model._scene_xyz.is_cuda


def get_deformed_pkg(model, t, want=("xyz", "rotation", "shs", "opacity", "scales"), raw_sh=False, flow_time=None, raw_scene=False):
    """Fused scene/gaussian_model.py:216-231 (+ get_scaling :89-91) on the raw parameters of `model`
    (any object with the reference GaussianModel's attributes).  Returns the reference's dict
    {'xyz','rotation','shs','opacity'} plus 'scales'.  With raw_sh=True the [N,M,3] SH tensor is not
    materialised: 'shs' is a diff_gaussian_rasterization.RawSH for GaussianRasterizer.forward_rawsh.
    With flow_time the dict also holds 'flow_xyz' = get_deformed_xyz(flow_time) (reference
    gaussian_renderer/__init__.py:57), evaluated in the same pass over the deformation rows.
    raw_scene=True (with raw_sh, when raw_scene_ok(model)): the deformation pass covers the OBJECT range only -- rows [0, Ns) of
    the returned tensors are uninitialised and must not be read -- and the RawSH carries the raw scene tensors, whose activations
    the rasterizer's preprocess applies itself (gradients come back from the rasterizer's backward)."""
    if raw_sh and "shs" in want:
        from diff_gaussian_rasterization import RawSH
        geo = bool(raw_scene) and raw_scene_ok(model)
        out = get_deformed_pkg(model, t, want=tuple(w for w in want if w != "shs"), flow_time=flow_time, raw_scene="objects_only" if geo else False)
        sp = model.shs_deform_param_scene
        out["shs"] = RawSH(model._scene_shs_dc, model._obj_shs_dc, model._scene_shs_rest, model._obj_shs_rest, sp, model.shs_deform_param_obj,
                           make_func_eval(float(t), model.order_args["shs"], sp.shape[-1]),
                           *((model._scene_xyz, model._scene_scaling, model._scene_rotation, model._scene_opacity, getattr(model, "grad_arena", None))
                             if geo else ()),
                           adam=getattr(getattr(model, "optimizer", None), "backward_epilogue", None))
        return out
    tensors = [getattr(model, _MODEL_ATTRS[n], None) for n in _PTRS]
    meta = (float(t), dict(model.order_args), bool(getattr(model, "use_time_mask", False)), tuple(want),
            None if flow_time is None else float(flow_time), getattr(model, "grad_arena", None), raw_scene == "objects_only")
    xyz, rot, shs, op, sc, flow = _DeformPkgFn.apply(meta, *tensors)
    out = {}
    if flow_time is not None:
        out["flow_xyz"] = flow
    for k, v in (("xyz", xyz), ("rotation", rot), ("shs", shs), ("opacity", op), ("scales", sc)):
        if k in want:
            out[k] = v
    return out


def get_deformed_xyz(model, t):
    """scene/gaussian_model.py:173-185 (used for the flow points at flow_time)."""
    return get_deformed_pkg(model, t, want=("xyz",))["xyz"]
