"""`diff_gaussian_rasterization._C` drop-in: same three entry points as the reference's
pybind module (submodules/depth-diff-gaussian-rasterization/ext.cpp:15-19,
rasterize_points.cu:35-275), implemented over the C ABI of libadgs_hip.so.

Tensors must live on a HIP ("cuda") device; there is no CPU path.
"""
import ctypes
import threading

import torch

from adgs import _lib

NUM_CHANNELS = 3
FLOW_CHANNELS = 3
SEMANTIC_CHANNELS = 32


def _ptr(t):
    """Data pointer of a contiguous fp32/int32 tensor; NULL for an empty tensor
    (the reference passes `torch.Tensor([])` for absent inputs)."""
    if t is None or t.numel() == 0:
        return None
    return t.data_ptr()


def _prep(t, device, name, dtype=torch.float32):
    if t is None or t.numel() == 0:
        return None
    if t.device != device:
        raise RuntimeError("%s must be on %s (got %s); there is no CPU rasterizer" % (name, device, t.device))
    if t.dtype != dtype:
        raise RuntimeError("%s must be %s (got %s)" % (name, dtype, t.dtype))
    return t.contiguous()


class _Buffer:
    """Byte buffer grown through the C allocator callback (the reference's resizeFunctional, rasterize_points.cu:27-33).

    ONE ctypes callback serves every buffer of the process: creating a CFUNCTYPE thunk costs ~10 us, and a forward needs three buffers.
    The library hands the callback the `user` pointer it was given next to it (include/adgs_rasterizer.h: adgs_alloc_fn) -- here a slot
    number of a per-thread table of the tensors being grown (the callback runs synchronously inside the forward call, on its thread)."""
    _tls = threading.local()

    def __init__(self, device):
        self.t = torch.empty(0, dtype=torch.uint8, device=device)
        slots = getattr(_Buffer._tls, "slots", None)
        if slots is None:
            slots = _Buffer._tls.slots = {}
        self.user = (max(slots) + 1) if slots else 1
        slots[self.user] = self.t             # the table holds the tensor, not self: no reference cycle (0.4 GB of frame state at C3)
        self.cb = _ALLOC_CB

    def release(self):
        """The forward call is over: the slot is free (the tensor lives on in the caller's hands)."""
        _Buffer._tls.slots.pop(self.user, None)


def _alloc(user, nbytes):
    t = _Buffer._tls.slots[user]
    t.resize_(int(nbytes))
    return t.data_ptr()


_ALLOC_CB = _lib.ALLOC_FN(_alloc)
_on = _lib.on_device


def _stream_ptr(device):
    return _lib.stream_ptr(device)


def rasterize_gaussians(background, means3D, colors, opacity, scales, rotations, scale_modifier, cov3D_precomp,
                        viewmatrix, projmatrix, tan_fovx, tan_fovy, image_height, image_width, sh, flow_points, semantic,
                        degree, campos, prefiltered, inv_depth, debug, training=True, plan=None):
    """training=False (extension): the forward-only render, adgs_raster_render -- the same images and radii bit for bit, the three
    state buffers come back as scratch no backward may be run over.  plan (a dict, extension): filled with the validated inputs of this call
    and their pointers, for rasterize_gaussians_backward(plan=...) of the same call pair (see rasterize_gaussians_rawsh)."""
    if means3D.dim() != 2 or means3D.size(1) != 3:
        raise RuntimeError("means3D must have dimensions (num_points, 3)")
    if not means3D.is_cuda:
        raise RuntimeError("means3D must be on a HIP device; there is no CPU rasterizer")
    lib = _lib.lib()
    dev = means3D.device
    P, H, W = means3D.size(0), int(image_height), int(image_width)
    D_S = 0
    if semantic.size(0) != 0:
        D_S = semantic.size(1)
        assert D_S <= SEMANTIC_CHANNELS
    if flow_points.size(0) != 0:
        assert flow_points.size(1) == FLOW_CHANNELS
    f32 = dict(dtype=torch.float32, device=dev)
    # outputs the kernels fully overwrite need no zero fill (adgs_raster_needs_zero_init)
    lazy = P != 0 and lib.adgs_raster_needs_zero_init(D_S) == 0
    alloc = lambda written, *shape: (torch.empty if (lazy and written) else torch.zeros)(shape, **f32)
    out_color = alloc(sh.size(0) != 0 or colors.size(0) != 0, NUM_CHANNELS, H, W)
    out_depth = alloc(True, 1, H, W)
    img_opacity = alloc(True, 1, H, W)
    img_flow = alloc(flow_points.size(0) != 0, FLOW_CHANNELS, H, W)
    img_semantic = alloc(D_S > 0, D_S, H, W)
    radii = (torch.empty if lazy else torch.zeros)((P,), dtype=torch.int32, device=dev)
    geom, binning, img = _Buffer(dev), _Buffer(dev), _Buffer(dev)
    rendered = 0
    if P != 0:
        M = sh.size(1) if sh.size(0) != 0 else 0
        keep = [_prep(t, dev, n) for t, n in (
            (background, "bg"), (means3D, "means3D"), (sh, "sh"), (colors, "colors_precomp"), (flow_points, "flow_points"),
            (semantic, "semantic"), (opacity, "opacities"), (scales, "scales"), (rotations, "rotations"),
            (cov3D_precomp, "cov3D_precomp"), (viewmatrix, "viewmatrix"), (projmatrix, "projmatrix"), (campos, "campos"))]
        bg_, m3_, sh_, col_, fl_, sem_, op_, sc_, rot_, cov_, view_, proj_, cam_ = keep
        if plan is not None:
            plan["keep"] = keep
            plan["ptrs"] = tuple(_ptr(t) for t in (bg_, m3_, sh_, col_, fl_, sem_, sc_, rot_, cov_, view_, proj_, cam_))
        try:
            with _on(dev):
                rendered = _lib.check((lib.adgs_raster_forward if training else lib.adgs_raster_render)(
                    geom.cb, geom.user, binning.cb, binning.user, img.cb, img.user, P, int(degree), M, D_S, _ptr(bg_), W, H,
                    _ptr(m3_), _ptr(sh_), _ptr(col_), _ptr(fl_), _ptr(sem_), _ptr(op_), _ptr(sc_), float(scale_modifier), _ptr(rot_),
                    _ptr(cov_), _ptr(view_), _ptr(proj_), _ptr(cam_), float(tan_fovx), float(tan_fovy), int(bool(prefiltered)),
                    _ptr(out_color), _ptr(out_depth), _ptr(img_opacity), _ptr(img_flow), _ptr(img_semantic), int(bool(inv_depth)),
                    _ptr(radii), int(bool(debug)), _stream_ptr(dev)), "adgs_raster_forward" if training else "adgs_raster_render")
        finally:
            for b in (geom, binning, img):
                b.release()
    else:
        for b in (geom, binning, img):
            b.release()
    return rendered, out_color, out_depth, img_opacity, radii, geom.t, binning.t, img.t, img_flow, img_semantic


def rasterize_gaussians_backward(background, means3D, radii, colors, scales, rotations, scale_modifier, cov3D_precomp,
                                 viewmatrix, projmatrix, tan_fovx, tan_fovy, dL_dout_color, dL_dout_depth, dL_dout_flow,
                                 dL_dout_semantic, semantic, flow_points, sh, degree, campos, geomBuffer, R, binningBuffer,
                                 imageBuffer, img_opacity, grad_img_opacity, inv_depth, debug, plan=None):
    lib = _lib.lib()
    dev = means3D.device
    P = means3D.size(0)
    H, W = dL_dout_color.size(1), dL_dout_color.size(2)
    M = sh.size(1) if sh.size(0) != 0 else 0
    D_S = 0
    if semantic.size(0) != 0:
        D_S = semantic.size(1)
        assert D_S <= SEMANTIC_CHANNELS
    if flow_points.size(0) != 0:
        assert flow_points.size(1) == FLOW_CHANNELS
    # the FORWARD of this state decided the pipeline (frame table of the library): its backward never looks at the environment
    lazy = P != 0 and lib.adgs_raster_backward_needs_zero_init(_ptr(geomBuffer), _ptr(imageBuffer), W, H, P) == 0
    z = lambda written, *shape: (torch.empty if (lazy and written) else torch.zeros)(shape, dtype=torch.float32, device=dev)
    has_sr = scales.size(0) != 0
    has_flow = flow_points.size(0) != 0 and dL_dout_flow is not None and dL_dout_flow.numel() != 0
    has_sem = D_S > 0 and dL_dout_semantic is not None and dL_dout_semantic.numel() != 0
    dL_dmeans3D, dL_dmeans2D, dL_dcolors = z(True, P, 3), z(True, P, 3), z(True, P, NUM_CHANNELS)
    dL_dopacity, dL_dcov3D, dL_dsh = z(True, P, 1), z(True, P, 6), z(True, P, M, 3)
    # dL_ddepths / dL_dconic never leave this function (the reference allocates them as scratch, rasterize_points.cu:195-206): the default
    # pipeline keeps them inside its fused preprocess backward (NULL = not wanted, include/adgs_rasterizer.h)
    dL_ddepths, dL_dconic = (None, None) if lazy else (z(True, P, 1), z(True, P, 2, 2))
    dL_dscales, dL_drotations = z(has_sr, P, 3), z(has_sr, P, 4)
    dL_dflow_points, dL_dsemantic = z(has_flow, P, FLOW_CHANNELS), z(has_sem, P, D_S)
    if P != 0:
        keep = [_prep(t, dev, n) for t, n in (
            (dL_dout_color, "dL_dout_color"), (dL_dout_depth, "dL_dout_depth"), (dL_dout_flow, "dL_dout_flow"), (dL_dout_semantic, "dL_dout_semantic"),
            (grad_img_opacity, "grad_img_opacity"), (img_opacity, "img_opacity"))]
        gc_, gd_, gf_, gs_, go_, io_ = keep
        if plan is not None and "ptrs" in plan:      # validated and marshalled by the forward of this call pair
            p_bg, p_m3, p_sh, p_col, p_fl, p_sem, p_sc, p_rot, p_cov, p_view, p_proj, p_cam = plan["ptrs"]
        else:
            keep2 = [_prep(t, dev, n) for t, n in (
                (background, "bg"), (means3D, "means3D"), (sh, "sh"), (colors, "colors_precomp"), (flow_points, "flow_points"),
                (semantic, "semantic"), (scales, "scales"), (rotations, "rotations"), (cov3D_precomp, "cov3D_precomp"),
                (viewmatrix, "viewmatrix"), (projmatrix, "projmatrix"), (campos, "campos"))]
            p_bg, p_m3, p_sh, p_col, p_fl, p_sem, p_sc, p_rot, p_cov, p_view, p_proj, p_cam = (_ptr(t) for t in keep2)
        radii_ = _prep(radii, dev, "radii", torch.int32)
        with _on(dev):
            _lib.check(lib.adgs_raster_backward(
                P, int(degree), M, int(R), D_S, p_bg, W, H, p_m3, p_sh, p_col, p_fl, p_sem,
                p_sc, float(scale_modifier), p_rot, p_cov, p_view, p_proj, p_cam,
                float(tan_fovx), float(tan_fovy), _ptr(radii_), _ptr(geomBuffer), _ptr(binningBuffer), _ptr(imageBuffer),
                _ptr(gc_), _ptr(gd_), _ptr(gf_), _ptr(gs_),
                _ptr(dL_dmeans2D), _ptr(dL_dconic), _ptr(dL_dopacity), _ptr(dL_dcolors), _ptr(dL_ddepths), _ptr(dL_dmeans3D),
                _ptr(dL_dcov3D), _ptr(dL_dsh), _ptr(dL_dscales), _ptr(dL_drotations), _ptr(dL_dflow_points), _ptr(dL_dsemantic),
                _ptr(go_), _ptr(io_), int(bool(inv_depth)), int(bool(debug)), _stream_ptr(dev)), "adgs_raster_backward")
    return (dL_dmeans2D, dL_dcolors, dL_dopacity, dL_dmeans3D, dL_dcov3D, dL_dsh, dL_dscales, dL_drotations,
            dL_dflow_points, dL_dsemantic)


def mark_visible(means3D, viewmatrix, projmatrix):
    if not means3D.is_cuda:
        raise RuntimeError("means3D must be on a HIP device; there is no CPU rasterizer")
    lib = _lib.lib()
    dev = means3D.device
    P = means3D.size(0)
    present = torch.zeros((P,), dtype=torch.bool, device=dev)
    if P != 0:
        m3, view, proj = (_prep(t, dev, n) for t, n in ((means3D, "means3D"), (viewmatrix, "viewmatrix"), (projmatrix, "projmatrix")))
        with _on(dev):
            _lib.check(lib.adgs_mark_visible(P, _ptr(m3), _ptr(view), _ptr(proj), _ptr(present), _stream_ptr(dev)), "adgs_mark_visible")
    return present


# ------------------------------------------------------------------ raw-SH fast path (no reference counterpart)
class ShSource(ctypes.Structure):
    """adgs_sh_source (include/adgs_rasterizer.h)."""
    from adgs.deform import FuncEval as _FE
    _fields_ = [("Ns", ctypes.c_int32)] + [(n, ctypes.c_void_p) for n in ("scene_dc", "obj_dc", "scene_rest", "obj_rest", "scene_deform",
                                                                              "obj_deform")] + [("f", _FE)] + \
               [(n, ctypes.c_void_p) for n in ("scene_xyz", "scene_scaling", "scene_rotation", "scene_opacity", "bg_image")]


class ShGrads(ctypes.Structure):
    """adgs_sh_grads (include/adgs_rasterizer.h)."""
    _fields_ = [("struct_bytes", ctypes.c_uint64)] + [(n, ctypes.c_void_p) for n in (
        "scene_dc", "obj_dc", "scene_rest", "obj_rest", "scene_deform", "obj_deform", "rgb_factor",
        "scene_xyz", "scene_scaling", "scene_rotation", "scene_opacity", "bg_image", "adam")]


def _sh_source(raw, dev):
    """raw: (scene_dc, obj_dc, scene_rest, obj_rest, scene_deform, obj_deform, func_eval[, (scene_xyz, scene_scaling, scene_rotation,
    scene_opacity) -- the RAW scene geometry, activations applied by the preprocess])."""
    src = ShSource()
    ts = [_prep(t, dev, n) for t, n in zip(raw[:6], ("scene_shs_dc", "obj_shs_dc", "scene_shs_rest", "obj_shs_rest",
                                                      "shs_deform_param_scene", "shs_deform_param_obj"))]
    src.Ns = raw[0].size(0)
    for name, t in zip(("scene_dc", "obj_dc", "scene_rest", "obj_rest", "scene_deform", "obj_deform"), ts):
        setattr(src, name, _ptr(t))
    src.f = raw[6]
    geo = raw[7] if len(raw) > 7 else None
    if geo is not None:
        if src.Ns == 0:
            geo = None
        else:
            gs = [_prep(t, dev, n) for t, n in zip(geo, ("_scene_xyz", "_scene_scaling", "_scene_rotation", "_scene_opacity"))]
            if any(g is None for g in gs) or [g.shape[0] for g in gs] != [src.Ns] * 4:
                raise RuntimeError("raw scene geometry: four tensors of Ns rows (xyz, scaling, rotation, opacity) are required")
            for name, t in zip(("scene_xyz", "scene_scaling", "scene_rotation", "scene_opacity"), gs):
                setattr(src, name, _ptr(t))
            ts = ts + gs
    bg_image = raw[8] if len(raw) > 8 else None
    if bg_image is not None:
        b = _prep(bg_image, dev, "bg_image")
        src.bg_image = _ptr(b)
        ts = ts + [b]
    return src, ts


def rasterize_gaussians_rawsh(background, means3D, opacity, scales, rotations, scale_modifier, viewmatrix, projmatrix, tan_fovx, tan_fovy,
                              image_height, image_width, sh_raw, flow_points, semantic, degree, campos, inv_depth, debug, training=True, plan=None):
    """plan (a dict, extension): filled with what the backward of this very call needs again -- the adgs_sh_source struct, the validated
    (contiguous) inputs and their pointers -- so that rasterize_gaussians_backward_rawsh(plan=...) does not validate and marshal them a
    second time (the Python between the loss kernels and the backward's first launch is GPU idle time on a slow host)."""
    if not means3D.is_cuda:
        raise RuntimeError("means3D must be on a HIP device; there is no CPU rasterizer")
    lib = _lib.lib()
    dev = means3D.device
    P, H, W = means3D.size(0), int(image_height), int(image_width)
    D_S = semantic.size(1) if semantic.size(0) != 0 else 0
    if lib.adgs_raster_needs_zero_init(D_S) != 0:
        raise RuntimeError("the raw-SH path needs the default pipeline (not ADGS_RASTER_MODE=classic, D_S <= ADGS_V2_MAX_SEMANTIC)")
    M = 1 + sh_raw[2].size(1)
    if sh_raw[0].size(0) + sh_raw[1].size(0) != P:
        raise RuntimeError("raw SH tensors do not match the number of Gaussians")
    f32 = dict(dtype=torch.float32, device=dev)
    # an output no kernel writes (absent input) is a zero image: materialised for the training forward (autograd hands it on), a broadcast
    # view of one zero for the forward-only render (29 MB of fill per 1920x1280 frame for an image nobody asked for)
    zero_img = (lambda *shape: torch.zeros((), **f32).expand(shape)) if not training else (lambda *shape: torch.zeros(shape, **f32))
    alloc = lambda written, *shape: torch.empty(shape, **f32) if (written and P != 0) else zero_img(*shape)
    out_color, out_depth, img_opacity = alloc(True, NUM_CHANNELS, H, W), alloc(True, 1, H, W), alloc(True, 1, H, W)
    img_flow, img_semantic = alloc(flow_points.size(0) != 0, FLOW_CHANNELS, H, W), alloc(D_S > 0, D_S, H, W)
    radii = (torch.empty if P != 0 else torch.zeros)((P,), dtype=torch.int32, device=dev)
    geom, binning, img = _Buffer(dev), _Buffer(dev), _Buffer(dev)
    rendered = 0
    bg_image = sh_raw[8] if len(sh_raw) > 8 else None
    if bg_image is not None and tuple(bg_image.shape) != (NUM_CHANNELS, H, W):
        raise RuntimeError("bg_image must be [3, H, W]")
    if P == 0 and bg_image is not None:
        out_color = bg_image.detach().clone()          # nothing to blend: T = 1 everywhere
    if P != 0:
        src, keep_sh = _sh_source(sh_raw, dev)
        keep = [_prep(t, dev, n) for t, n in ((background, "bg"), (means3D, "means3D"), (flow_points, "flow_points"), (semantic, "semantic"),
                                              (opacity, "opacities"), (scales, "scales"), (rotations, "rotations"), (viewmatrix, "viewmatrix"),
                                              (projmatrix, "projmatrix"), (campos, "campos"))]
        bg_, m3_, fl_, sem_, op_, sc_, rot_, view_, proj_, cam_ = keep
        if plan is not None:
            plan["src"], plan["keep"] = src, (keep_sh, keep)
            plan["ptrs"] = (_ptr(bg_), _ptr(m3_), _ptr(fl_), _ptr(sem_), _ptr(sc_), _ptr(rot_), _ptr(view_), _ptr(proj_), _ptr(cam_))
        try:
            with _on(dev):
                rendered = _lib.check((lib.adgs_raster_forward_rawsh if training else lib.adgs_raster_render_rawsh)(
                    geom.cb, geom.user, binning.cb, binning.user, img.cb, img.user, P, int(degree), M, D_S, _ptr(bg_), W, H, _ptr(m3_), ctypes.byref(src),
                    _ptr(fl_), _ptr(sem_), _ptr(op_), _ptr(sc_), float(scale_modifier), _ptr(rot_), _ptr(view_), _ptr(proj_), _ptr(cam_),
                    float(tan_fovx), float(tan_fovy), _ptr(out_color), _ptr(out_depth), _ptr(img_opacity), _ptr(img_flow), _ptr(img_semantic),
                    int(bool(inv_depth)), _ptr(radii), int(bool(debug)), _stream_ptr(dev)), "adgs_raster_forward_rawsh" if training else "adgs_raster_render_rawsh")
        finally:
            for b in (geom, binning, img):
                b.release()
    else:
        for b in (geom, binning, img):
            b.release()
    return rendered, out_color, out_depth, img_opacity, radii, geom.t, binning.t, img.t, img_flow, img_semantic


def rasterize_gaussians_backward_rawsh(background, means3D, radii, scales, rotations, scale_modifier, viewmatrix, projmatrix, tan_fovx, tan_fovy,
                                       dL_dout_color, dL_dout_depth, dL_dout_flow, dL_dout_semantic, semantic, flow_points, sh_raw,
                                       sh_needs_grad, degree, campos, geomBuffer, R, binningBuffer, imageBuffer, img_opacity, grad_img_opacity,
                                       inv_depth, debug, want_rgb_factor=False, geo_grad_alloc=None, adam=None, plan=None, want_sem_grad=True):
    """With want_rgb_factor the result carries one more entry: the [P,3] clamp-masked colour gradient every SH gradient row is a
    multiple of (include/adgs_exchange.h); combined with sh_needs_grad all False the SH rows are not materialised at all.
    adam: an adgs.optim.BackwardClaim (FusedAdam(in_backward=True)) -- the tensors it names take the Adam step inside the backward
    kernels and get no gradient tensor (include/adgs_optim.h: adgs_sh_adam)."""
    lib = _lib.lib()
    dev = means3D.device
    P = means3D.size(0)
    H, W = dL_dout_color.size(1), dL_dout_color.size(2)
    M = 1 + sh_raw[2].size(1)
    D_S = semantic.size(1) if semantic.size(0) != 0 else 0
    e = lambda *shape: torch.empty(shape, dtype=torch.float32, device=dev)
    has_flow = flow_points.size(0) != 0 and dL_dout_flow is not None and dL_dout_flow.numel() != 0
    has_sem = D_S > 0 and dL_dout_semantic is not None and dL_dout_semantic.numel() != 0
    dL_dmeans3D, dL_dmeans2D, dL_dopacity, dL_dscales, dL_drotations = e(P, 3), e(P, 3), e(P, 1), e(P, 3), e(P, 4)
    # dL_dconic / dL_dcolors / dL_ddepths / dL_dcov3D are intermediates of the reference's backward (consumed inside the fused
    # preprocess backward here): not materialised on this path (NULL = not wanted, include/adgs_rasterizer.h)
    dL_dcolors = dL_ddepths = dL_dconic = dL_dcov3D = None
    dL_dflow_points = e(P, FLOW_CHANNELS) if has_flow else torch.zeros((P, FLOW_CHANNELS), dtype=torch.float32, device=dev)
    if not want_sem_grad and D_S == 1:
        dL_dsemantic = None        # a constant semantic input (the object mask of gaussian_renderer.render()): NULL = not wanted (channel 0 is a row of the preprocess backward)
    else:
        dL_dsemantic = e(P, D_S) if has_sem else torch.zeros((P, D_S), dtype=torch.float32, device=dev)
    sh_needs_grad = list(sh_needs_grad)
    fused = adam.fused if (adam is not None and P != 0) else {}
    for i, name in enumerate(("scene_rest", "obj_rest", "scene_deform", "obj_deform")):
        if fused.get(name):
            sh_needs_grad[2 + i] = False                           # updated in place of being stored
    need = list(sh_needs_grad)
    # the deform-param gradients (stored or applied) are derived from the dc gradients
    need[0], need[1] = need[0] or need[4] or bool(fused.get("scene_deform")), need[1] or need[5] or bool(fused.get("obj_deform"))
    sh_grads = [torch.empty_like(t) if (nd and t is not None and t.numel() != 0) else None for t, nd in zip(sh_raw[:6], need)]
    if torch.is_tensor(want_rgb_factor):           # the caller's own [P,3] destination (adgs.dp.FactoredSHExchange's send buffer)
        rgb_factor = want_rgb_factor
        if rgb_factor.shape != (P, 3) or rgb_factor.dtype != torch.float32 or not rgb_factor.is_contiguous() or rgb_factor.device != dev:
            raise RuntimeError("rgb_factor destination must be a contiguous float32 [P,3] tensor on the rasterizer's device")
    else:
        rgb_factor = (e(P, 3) if P != 0 else torch.zeros((0, 3), dtype=torch.float32, device=dev)) if want_rgb_factor else None
    geo_grads = None
    bg_grad = None
    if P == 0 and len(sh_raw) > 8 and sh_raw[8] is not None:
        bg_grad = dL_dout_color.clone()                # T = 1 everywhere
    if P != 0:
        if plan is not None and "src" in plan:       # the forward of this call pair validated and marshalled these already
            src = plan["src"]
        else:
            src, keep_sh = _sh_source(sh_raw, dev)
        gs = ShGrads()
        gs.struct_bytes = ctypes.sizeof(ShGrads)
        for name, t in zip(("scene_dc", "obj_dc", "scene_rest", "obj_rest", "scene_deform", "obj_deform"), sh_grads):
            setattr(gs, name, _ptr(t))
        gs.rgb_factor = _ptr(rgb_factor)
        if fused:
            gs.adam = ctypes.addressof(adam.struct)
        bg_image = sh_raw[8] if len(sh_raw) > 8 else None
        if bg_image is not None:
            bg_grad = torch.empty_like(bg_image, memory_format=torch.contiguous_format)
            gs.bg_image = _ptr(bg_grad)
        geo = sh_raw[7] if len(sh_raw) > 7 and src.scene_xyz else None
        if geo is not None:          # raw scene geometry: the gradients of the four raw tensors (every row written by the kernel)
            names = ("scene_xyz", "scene_scaling", "scene_rotation", "scene_opacity")
            geo_grads = [(geo_grad_alloc(n, t) if geo_grad_alloc is not None else None) for n, t in zip(names, geo)]
            geo_grads = [g if g is not None else torch.empty_like(t) for g, t in zip(geo_grads, geo)]
            for n, g in zip(names, geo_grads):
                setattr(gs, n, _ptr(g))
        keep = [_prep(t, dev, n) for t, n in (
            (dL_dout_color, "dL_dout_color"), (dL_dout_depth, "dL_dout_depth"), (dL_dout_flow, "dL_dout_flow"),
            (dL_dout_semantic, "dL_dout_semantic"), (grad_img_opacity, "grad_img_opacity"), (img_opacity, "img_opacity"))]
        gc_, gd_, gf_, gs_, go_, io_ = keep
        if plan is not None and "ptrs" in plan:
            p_bg, p_m3, p_fl, p_sem, p_sc, p_rot, p_view, p_proj, p_cam = plan["ptrs"]
        else:
            keep2 = [_prep(t, dev, n) for t, n in (
                (background, "bg"), (means3D, "means3D"), (flow_points, "flow_points"), (semantic, "semantic"), (scales, "scales"),
                (rotations, "rotations"), (viewmatrix, "viewmatrix"), (projmatrix, "projmatrix"), (campos, "campos"))]
            p_bg, p_m3, p_fl, p_sem, p_sc, p_rot, p_view, p_proj, p_cam = (_ptr(t) for t in keep2)
        radii_ = _prep(radii, dev, "radii", torch.int32)
        with _on(dev):
            _lib.check(lib.adgs_raster_backward_rawsh(
                P, int(degree), M, int(R), D_S, p_bg, W, H, p_m3, ctypes.byref(src), p_fl, p_sem, p_sc,
                float(scale_modifier), p_rot, p_view, p_proj, p_cam, float(tan_fovx), float(tan_fovy), _ptr(radii_),
                _ptr(geomBuffer), _ptr(binningBuffer), _ptr(imageBuffer), _ptr(gc_), _ptr(gd_), _ptr(gf_), _ptr(gs_),
                _ptr(dL_dmeans2D), _ptr(dL_dconic), _ptr(dL_dopacity), _ptr(dL_dcolors), _ptr(dL_ddepths), _ptr(dL_dmeans3D), _ptr(dL_dcov3D),
                ctypes.byref(gs), _ptr(dL_dscales), _ptr(dL_drotations), _ptr(dL_dflow_points), _ptr(dL_dsemantic), _ptr(go_), _ptr(io_),
                int(bool(inv_depth)), int(bool(debug)), _stream_ptr(dev)), "adgs_raster_backward_rawsh")
    sh_grads = [g if nd else None for g, nd in zip(sh_grads, sh_needs_grad)]
    if P == 0 or not (len(sh_raw) > 7 and sh_raw[7] is not None and sh_raw[0].size(0) > 0):
        geo_grads = None
    res = (dL_dmeans2D, dL_dopacity, dL_dmeans3D, sh_grads, dL_dscales, dL_drotations, dL_dflow_points, dL_dsemantic)
    return res + ((rgb_factor,) if rgb_factor is not None else (None,)) + (geo_grads, bg_grad)
