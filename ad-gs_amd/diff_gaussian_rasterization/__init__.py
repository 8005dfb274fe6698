"""Drop-in `diff_gaussian_rasterization` for AD-GS on MI355X (gfx950).

Same public surface and autograd contract as the reference package
(submodules/depth-diff-gaussian-rasterization/diff_gaussian_rasterization/__init__.py:21-251):

    GaussianRasterizationSettings   13-field NamedTuple            (ref :176-189)
    GaussianRasterizer(nn.Module)   .forward(...) -> 6-tuple, .markVisible(...)   (ref :191-251)
    rasterize_gaussians(...)        functional form                 (ref :21-46)

so `gaussian_renderer.render()`, `scene/gaussian_model.py` and `train.py` of the
reference run unchanged.  Underneath, `_C` is a ctypes binding of the hand-written
HIP library (libadgs_hip.so); nothing here falls back to PyTorch or the CPU.
"""
from typing import NamedTuple

import torch
import torch.nn as nn

from . import _C


class GaussianRasterizationSettings(NamedTuple):
    image_height: int
    image_width: int
    tanfovx: float
    tanfovy: float
    bg: torch.Tensor
    scale_modifier: float
    viewmatrix: torch.Tensor
    projmatrix: torch.Tensor
    sh_degree: int
    campos: torch.Tensor
    prefiltered: bool
    inv_depth: bool
    debug: bool


def _snapshot(args):
    """CPU copies of every tensor argument, for the debug dumps (ref :17-19)."""
    return tuple(a.detach().cpu().clone() if isinstance(a, torch.Tensor) else a for a in args)


def _call_with_dump(fn, args, debug, dump_name, phase):
    """With settings.debug the reference snapshots the arguments before the call and
    writes them with torch.save if the native call raises (ref :92-99, :149-156)."""
    if not debug:
        return fn(*args)
    saved = _snapshot(args)
    try:
        return fn(*args)
    except Exception:
        torch.save(saved, dump_name)
        print("\nAn error occured in %s. Writing %s for debugging.\n" % (phase, dump_name))
        raise


class _RasterizeGaussians(torch.autograd.Function):
    """11 inputs -> 6 outputs; backward returns 11 grads in input order (ref :48-174)."""

    @staticmethod
    def forward(ctx, means3D, means2D, sh, colors_precomp, opacities, scales, rotations, cov3Ds_precomp,
                flow_points, semantic, raster_settings):
        s = raster_settings
        native_args = (s.bg, means3D, colors_precomp, opacities, scales, rotations, s.scale_modifier, cov3Ds_precomp,
                       s.viewmatrix, s.projmatrix, s.tanfovx, s.tanfovy, s.image_height, s.image_width, sh, flow_points,
                       semantic, s.sh_degree, s.campos, s.prefiltered, s.inv_depth, s.debug)
        plan = {}      # the validated inputs / pointers of this call, for its backward (_C.rasterize_gaussians)
        (num_rendered, color, depth, img_opacity, radii, geom_buf, binning_buf, img_buf, img_flow,
         img_semantic) = _call_with_dump(lambda *a: _C.rasterize_gaussians(*a, plan=plan), native_args, s.debug, "snapshot_fw.dump", "forward")
        ctx.raster_settings, ctx.plan = s, plan
        ctx.num_rendered = num_rendered
        ctx.set_materialize_grads(False)     # outputs the loss does not use arrive as None (NULL for the kernels), not as zero fills
        ctx.save_for_backward(colors_precomp, means3D, scales, rotations, cov3Ds_precomp, radii, sh, geom_buf, binning_buf,
                              img_buf, img_opacity, flow_points, semantic)
        return color, radii, depth, img_opacity, img_flow, img_semantic

    @staticmethod
    def backward(ctx, grad_out_color, grad_radii, grad_depth, grad_img_opacity, grad_img_flow, grad_img_semantic):
        s = ctx.raster_settings
        (colors_precomp, means3D, scales, rotations, cov3Ds_precomp, radii, sh, geom_buf, binning_buf, img_buf,
         img_opacity, flow_points, semantic) = ctx.saved_tensors
        if grad_out_color is None:           # the C binding takes H, W from this tensor
            grad_out_color = torch.zeros((3, s.image_height, s.image_width), dtype=torch.float32, device=means3D.device)
        native_args = (s.bg, means3D, radii, colors_precomp, scales, rotations, s.scale_modifier, cov3Ds_precomp,
                       s.viewmatrix, s.projmatrix, s.tanfovx, s.tanfovy, grad_out_color, grad_depth, grad_img_flow,
                       grad_img_semantic, semantic, flow_points, sh, s.sh_degree, s.campos, geom_buf, ctx.num_rendered,
                       binning_buf, img_buf, img_opacity, grad_img_opacity, s.inv_depth, s.debug)
        (grad_means2D, grad_colors_precomp, grad_opacities, grad_means3D, grad_cov3Ds_precomp, grad_sh, grad_scales,
         grad_rotations, grad_flow_points, grad_semantic) = _call_with_dump(
            lambda *a: _C.rasterize_gaussians_backward(*a, plan=ctx.plan), native_args, s.debug, "snapshot_bw.dump", "backward")
        return (grad_means3D, grad_means2D, grad_sh, grad_colors_precomp, grad_opacities, grad_scales, grad_rotations,
                grad_cov3Ds_precomp, grad_flow_points, grad_semantic, None)


class RawSH(NamedTuple):
    """SH coefficients left in the reference GaussianModel's raw layout (scene || object,
    dc + f_shs(t) || rest) for `GaussianRasterizer.forward_rawsh` (adgs.deform.get_deformed_pkg(raw_sh=True))."""
    scene_dc: torch.Tensor
    obj_dc: torch.Tensor
    scene_rest: torch.Tensor
    obj_rest: torch.Tensor
    scene_deform: torch.Tensor
    obj_deform: torch.Tensor
    func_eval: object          # adgs.deform.FuncEval of f_shs at the camera time
    # RAW scene geometry (all four or None): _scene_xyz, _scene_scaling (log), _scene_rotation (unnormalised), _scene_opacity (logit).
    # The preprocess applies exp / normalize / sigmoid itself for the Gaussians idx < Ns and never reads rows idx < Ns of means3D /
    # scales / rotations / opacities / flow_points; the backward writes the gradients of these raw tensors directly.
    scene_xyz: object = None
    scene_scaling: object = None
    scene_rotation: object = None
    scene_opacity: object = None
    grad_arena: object = None  # adgs.dp.GradArena: where the raw scene geometry gradients are written (one flat all-reduce buffer)
    adam: object = None        # adgs.optim.BackwardEpilogue (FusedAdam(in_backward=True)): when armed, the backward applies the Adam step to the
    #                            rest / deformation tensors in place of storing their gradients (their .grad stays None)


class _RasterizeGaussiansRawSH(torch.autograd.Function):
    """Same operator as _RasterizeGaussians with the SH input/gradients in the raw tensors' layout.

    With a `factor_sink` (a list) the backward does not materialise the six SH parameter gradients: it appends the
    [P,3] clamp-masked colour gradient they are all multiples of (include/adgs_exchange.h) to the list and returns None
    for them; adgs.dp.FactoredSHExchange turns the factors of all cameras of an iteration into the summed gradients."""

    @staticmethod
    def forward(ctx, means3D, means2D, opacities, scales, rotations, flow_points, semantic, scene_dc, obj_dc, scene_rest, obj_rest,
                scene_deform, obj_deform, func_eval, raster_settings, factor_sink=None, scene_xyz=None, scene_scaling=None, scene_rotation=None,
                scene_opacity=None, grad_arena=None, bg_image=None, adam=None):
        s = raster_settings
        geo = (scene_xyz, scene_scaling, scene_rotation, scene_opacity) if scene_xyz is not None else None
        raw = (scene_dc, obj_dc, scene_rest, obj_rest, scene_deform, obj_deform, func_eval, geo, bg_image)
        plan = {}      # the validated inputs / pointers / adgs_sh_source of this call, for its backward (_C.rasterize_gaussians_rawsh)
        (num_rendered, color, depth, img_opacity, radii, geom_buf, binning_buf, img_buf, img_flow, img_semantic) = _C.rasterize_gaussians_rawsh(
            s.bg, means3D, opacities, scales, rotations, s.scale_modifier, s.viewmatrix, s.projmatrix, s.tanfovx, s.tanfovy, s.image_height,
            s.image_width, raw, flow_points, semantic, s.sh_degree, s.campos, s.inv_depth, s.debug, plan=plan)
        ctx.raster_settings, ctx.num_rendered, ctx.func_eval, ctx.factor_sink, ctx.plan = s, num_rendered, func_eval, factor_sink, plan
        ctx.has_geo, ctx.grad_arena, ctx.has_bg, ctx.adam = geo is not None, grad_arena, bg_image is not None, adam
        ctx.set_materialize_grads(False)
        ctx.save_for_backward(means3D, scales, rotations, radii, geom_buf, binning_buf, img_buf, img_opacity, flow_points, semantic,
                              scene_dc, obj_dc, scene_rest, obj_rest, scene_deform, obj_deform, *(geo or ()), *((bg_image,) if bg_image is not None else ()))
        return color, radii, depth, img_opacity, img_flow, img_semantic

    @staticmethod
    def backward(ctx, grad_out_color, grad_radii, grad_depth, grad_img_opacity, grad_img_flow, grad_img_semantic):
        s = ctx.raster_settings
        saved = ctx.saved_tensors
        (means3D, scales, rotations, radii, geom_buf, binning_buf, img_buf, img_opacity, flow_points, semantic,
         scene_dc, obj_dc, scene_rest, obj_rest, scene_deform, obj_deform) = saved[:16]
        geo = tuple(saved[16:20]) if ctx.has_geo else None
        bg_image = saved[-1] if ctx.has_bg else None
        raw = (scene_dc, obj_dc, scene_rest, obj_rest, scene_deform, obj_deform, ctx.func_eval, geo, bg_image)
        if grad_out_color is None:
            grad_out_color = torch.zeros((3, s.image_height, s.image_width), dtype=torch.float32, device=means3D.device)
        factored = ctx.factor_sink is not None
        need = (False,) * 6 if factored else ctx.needs_input_grad[7:13]
        arena = ctx.grad_arena
        # FusedAdam(in_backward=True), armed for this backward: the step is applied where the gradient rows are produced
        claim = ctx.adam.claim(dict(scene_rest=scene_rest, obj_rest=obj_rest, scene_deform=scene_deform, obj_deform=obj_deform),
                               dict(zip(("scene_rest", "obj_rest", "scene_deform", "obj_deform"), need[2:6])), factored) if ctx.adam is not None else None
        try:
            res = _C.rasterize_gaussians_backward_rawsh(
                s.bg, means3D, radii, scales, rotations, s.scale_modifier, s.viewmatrix, s.projmatrix, s.tanfovx, s.tanfovy, grad_out_color,
                grad_depth, grad_img_flow, grad_img_semantic, semantic, flow_points, raw, need, s.sh_degree, s.campos,
                geom_buf, ctx.num_rendered, binning_buf, img_buf, img_opacity, grad_img_opacity, s.inv_depth, s.debug,
                want_rgb_factor=(ctx.factor_sink.next_target(means3D.size(0)) if hasattr(ctx.factor_sink, "next_target") else True) if factored else False,
                geo_grad_alloc=(arena.take if arena is not None else None), adam=claim, plan=ctx.plan, want_sem_grad=ctx.needs_input_grad[6])
        except Exception:
            if claim is not None:          # the native call validates everything before its first launch: nothing was stepped
                claim.rollback()
            raise
        (g_means2D, g_opac, g_means3D, g_sh, g_scales, g_rot, g_flow, g_sem, g_factor, g_geo, g_bg) = res
        if factored:
            ctx.factor_sink.append(g_factor)
        g_geo = tuple(g_geo) if g_geo is not None else (None,) * 4
        return (g_means3D, g_means2D, g_opac, g_scales, g_rot, g_flow, g_sem) + tuple(g_sh) + (None, None, None) + g_geo + (None, g_bg, None)


def _no_backward_can_follow(*tensors):
    """True when autograd will never ask for a backward of this call: gradients are switched off (the reference's evaluation path,
    render.py:156 `with torch.no_grad()`), or no input requires one."""
    return not torch.is_grad_enabled() or not any(torch.is_tensor(t) and t.requires_grad for t in tensors)


def rasterize_gaussians(means3D, means2D, sh, colors_precomp, opacities, scales, rotations, cov3Ds_precomp, flow_points,
                        semantic, raster_settings):
    if _no_backward_can_follow(means3D, means2D, sh, colors_precomp, opacities, scales, rotations, cov3Ds_precomp, flow_points, semantic):
        # the forward-only render (adgs_raster_render): the same six outputs bit for bit, nothing published for a backward
        s = raster_settings
        args = (s.bg, means3D, colors_precomp, opacities, scales, rotations, s.scale_modifier, cov3Ds_precomp, s.viewmatrix, s.projmatrix, s.tanfovx,
                s.tanfovy, s.image_height, s.image_width, sh, flow_points, semantic, s.sh_degree, s.campos, s.prefiltered, s.inv_depth, s.debug)
        with torch.no_grad():
            r = _call_with_dump(lambda *a: _C.rasterize_gaussians(*a, training=False), args, s.debug, "snapshot_fw.dump", "forward")
        return r[1], r[4], r[2], r[3], r[8], r[9]
    return _RasterizeGaussians.apply(means3D, means2D, sh, colors_precomp, opacities, scales, rotations, cov3Ds_precomp,
                                     flow_points, semantic, raster_settings)


class GaussianRasterizer(nn.Module):
    def __init__(self, raster_settings):
        super().__init__()
        self.raster_settings = raster_settings

    def markVisible(self, positions):
        """Boolean mask of points passing the near-plane test for this camera (ref :196-205)."""
        with torch.no_grad():
            s = self.raster_settings
            return _C.mark_visible(positions, s.viewmatrix, s.projmatrix)

    def forward_rawsh(self, means3D, means2D, opacities, sh_raw, scales, rotations, flow_points=None, semantic=None, factor_sink=None, bg_image=None):
        """Extension (no reference counterpart): like forward(), with the SH coefficients given as a RawSH.
        factor_sink: see _RasterizeGaussiansRawSH (factored SH gradients for data-parallel training).
        bg_image [3,H,W]: per-pixel background composited in the blend epilogue -- the first output is then
        `foreground + (1 - img_opacity) * bg_image` (gaussian_renderer/__init__.py:93-94) and bg_image receives a gradient."""
        empty = lambda t: torch.Tensor([]) if t is None else t
        geo4 = [getattr(sh_raw, n, None) for n in ("scene_xyz", "scene_scaling", "scene_rotation", "scene_opacity")]
        if _no_backward_can_follow(means3D, means2D, opacities, scales, rotations, flow_points, semantic, sh_raw.scene_dc, sh_raw.obj_dc, sh_raw.scene_rest,
                                   sh_raw.obj_rest, sh_raw.scene_deform, sh_raw.obj_deform, bg_image, *geo4):
            s = self.raster_settings
            raw = (sh_raw.scene_dc, sh_raw.obj_dc, sh_raw.scene_rest, sh_raw.obj_rest, sh_raw.scene_deform, sh_raw.obj_deform, sh_raw.func_eval,
                   tuple(geo4) if geo4[0] is not None else None, bg_image)
            with torch.no_grad():
                r = _C.rasterize_gaussians_rawsh(s.bg, means3D, opacities, scales, rotations, s.scale_modifier, s.viewmatrix, s.projmatrix, s.tanfovx, s.tanfovy,
                                                 s.image_height, s.image_width, raw, empty(flow_points), empty(semantic), s.sh_degree, s.campos, s.inv_depth,
                                                 s.debug, training=False)
            return r[1], r[4], r[2], r[3], r[8], r[9]
        return _RasterizeGaussiansRawSH.apply(means3D, means2D, opacities, scales, rotations, empty(flow_points), empty(semantic),
                                              sh_raw.scene_dc, sh_raw.obj_dc, sh_raw.scene_rest, sh_raw.obj_rest, sh_raw.scene_deform,
                                              sh_raw.obj_deform, sh_raw.func_eval, self.raster_settings, factor_sink,
                                              getattr(sh_raw, "scene_xyz", None), getattr(sh_raw, "scene_scaling", None),
                                              getattr(sh_raw, "scene_rotation", None), getattr(sh_raw, "scene_opacity", None),
                                              getattr(sh_raw, "grad_arena", None), bg_image, getattr(sh_raw, "adam", None))

    def forward(self, means3D, means2D, opacities, shs=None, colors_precomp=None, scales=None, rotations=None,
                cov3D_precomp=None, flow_points=None, semantic=None):
        # validation identical to ref :214-218 (neither shs nor colours is allowed)
        if shs is not None and colors_precomp is not None:
            raise Exception('Cannot provice both shs and colors_precomp')
        has_sr = scales is not None and rotations is not None
        any_sr = scales is not None or rotations is not None
        if (not has_sr and cov3D_precomp is None) or (any_sr and cov3D_precomp is not None):
            raise Exception('Please provide exactly one of either scale/rotation pair or precomputed 3D covariance!')
        # absent optional inputs travel as empty CPU tensors, like the reference (:220-236)
        empty = lambda t: torch.Tensor([]) if t is None else t
        return rasterize_gaussians(means3D, means2D, empty(shs), empty(colors_precomp), opacities, empty(scales),
                                   empty(rotations), empty(cov3D_precomp), empty(flow_points), empty(semantic),
                                   self.raster_settings)
