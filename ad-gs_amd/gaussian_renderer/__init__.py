"""`gaussian_renderer.render()` for AD-GS on MI355X.

Signature and result dictionary are the reference's (gaussian_renderer/__init__.py:18-115) so that train.py / render.py
keep calling it unchanged; the body is organised around the HIP path: one fused deformation pass (both time stamps when
a flow target exists), the raw-SH entry of the rasterizer when the model hands out its SH tensors un-concatenated, and the
background composite.

`pc` is any object with the reference GaussianModel's render-time getters (scene/gaussian_model.py:88-231): get_xyz,
get_deformed_xyz(t), get_deformed_pkg(t), get_scaling, get_obj_mask, active_sh_degree.  `env_map` needs
get_image_background(camera) (scene/env.py:44-76; adgs.env.EnvironmentMap) or may be None (black background).
"""
import math

import torch

from diff_gaussian_rasterization import GaussianRasterizationSettings, GaussianRasterizer


_BLACK = {}


def _camera_matrices(cam, device):
    """The camera's view / projection matrices and centre as CONTIGUOUS tensors on `device`, the same three tensors at every call while the
    camera does not change.  The reference's Camera builds world_view_transform as a transposed (non-contiguous) tensor (scene/cameras.py:76-82):
    taking `.contiguous()` per render would hand the library a fresh temporary each time -- and the library recognises a camera by the
    addresses of its matrices (the tile order and the depth-slab bounds of its previous render: include/adgs_rasterizer.h,
    adgs_frame_status.order_hint).  Kept on the camera object itself; rebuilt when a matrix is replaced or written in place."""
    src = (cam.world_view_transform, cam.full_proj_transform, cam.camera_center)
    sig = tuple((t.data_ptr(), t._version, str(t.device)) for t in src) + (str(device),)
    ent = getattr(cam, "_adgs_matrices", None)
    if ent is not None and ent[0] == sig:
        return ent[1]
    mats = tuple((t if t.device == device else t.to(device)).contiguous() for t in src)
    try:
        cam._adgs_matrices = (sig, mats)
    except AttributeError:              # a camera type without instance attributes: per-call temporaries (no hints)
        pass
    return mats


def _camera_settings(cam, pc, pipe, scale_modifier, device):
    """Everything the rasterizer needs to know about the view (the background colour is black: the sky comes from env_map).
    The reference's Camera keeps its matrices on the GPU (scene/cameras.py:77-80); a camera object that holds host tensors costs
    three blocking uploads at its first render here (kept on the camera afterwards: _camera_matrices)."""
    black = _BLACK.get(device)
    if black is None:
        black = _BLACK[device] = torch.zeros(3, dtype=torch.float32, device=device)
    view, proj, center = _camera_matrices(cam, device)
    return GaussianRasterizationSettings(
        int(cam.image_height), int(cam.image_width), math.tan(0.5 * cam.FoVx), math.tan(0.5 * cam.FoVy),
        black, scale_modifier, view, proj, pc.active_sh_degree, center, False, pipe.inv_depth, pipe.debug)


def _deformed_state(pc, t, flow_pkg, full_rows=False):
    """(deform_pkg, flow_points): the Gaussians at the camera time and, for a flow target, their positions at its time stamp.
    full_rows: the caller reads the scene rows of the deformed tensors (override_color goes through the plain rasterizer entry), so a
    model on the raw-scene path must materialise them."""
    kw = dict(full_rows=True) if (full_rows and getattr(pc, "raw_scene", False)) else {}
    if flow_pkg is None:
        return pc.get_deformed_pkg(t, **kw), None
    flow_t = flow_pkg[0]
    if getattr(pc, "supports_fused_flow", False):          # adgs.model: both time stamps in one pass over the deformation rows
        pkg = pc.get_deformed_pkg(t, flow_time=flow_t, **kw)
        return pkg, pkg["flow_xyz"]
    return pc.get_deformed_pkg(t, **kw), pc.get_deformed_xyz(flow_t)


def _rasterize(rasterizer, pc, pkg, means2D, override_color, flow_points, semantic, sh_factor_sink=None, bg_image=None):
    """bg_image: the environment-map background; on the raw-SH path it is composited in the blend epilogue (the first output is then the
    final `render`), otherwise the caller composites."""
    scales = pkg["scales"] if "scales" in pkg else pc.get_scaling
    shs = None if override_color is not None else pkg["shs"]
    if shs is not None and not torch.is_tensor(shs):
        # a RawSH: the rasterizer reads dc / rest / deformation rows in place, the [N,16,3] tensor is never built
        return rasterizer.forward_rawsh(pkg["xyz"], means2D, pkg["opacity"], shs, scales, pkg["rotation"], flow_points=flow_points,
                                        semantic=semantic, factor_sink=None if sh_factor_sink is None else sh_factor_sink(pkg["xyz"]),
                                        bg_image=bg_image), bg_image is not None
    if sh_factor_sink is not None:
        raise RuntimeError("sh_factor_sink needs the raw-SH path (a model whose get_deformed_pkg hands out a RawSH)")
    return rasterizer(means3D=pkg["xyz"], means2D=means2D, opacities=pkg["opacity"], shs=shs, colors_precomp=override_color, scales=scales,
                      rotations=pkg["rotation"], flow_points=flow_points, semantic=semantic), False


class _LazyResult(dict):
    """render()'s result dictionary when the scene range never went through the deformation pass (model.raw_scene: the rasterizer
    read the raw scene tensors): the reference's 'xyz' / 'rotation' / 'opacity' (/ 'scales', 'flow_xyz') entries
    (gaussian_renderer/__init__.py:99-115) are produced on first access by a full deformation pass instead of being stored --
    train.py / render.py never read them."""

    def __init__(self, thunks):
        super().__init__()
        self._thunks = dict(thunks)

    def __missing__(self, key):
        if key in self._thunks:
            self[key] = self._thunks.pop(key)()
            return dict.__getitem__(self, key)
        raise KeyError(key)

    def __contains__(self, key):
        return dict.__contains__(self, key) or key in self._thunks

    def get(self, key, default=None):
        return self[key] if key in self else default

    def keys(self):
        return list(dict.keys(self)) + list(self._thunks)


_ZEROS = {}


def screenspace_points(n, device):
    """The reference's `torch.zeros_like(pc.get_xyz, requires_grad=True)` (gaussian_renderer/__init__.py:27-31): a fresh autograd
    leaf of zeros whose only purpose is to receive dL/dmeans2D in `.grad`.  Nothing ever writes to it, so every call hands out a
    new leaf over the SAME zero-filled storage: no `cat` of the positions and no fill kernel per frame."""
    key = (int(n), str(device))
    z = _ZEROS.get(key)
    if z is None:
        if len(_ZEROS) > 8:
            _ZEROS.clear()
        z = _ZEROS[key] = torch.zeros((int(n), 3), dtype=torch.float32, device=device)
    return z.detach().requires_grad_(True)


def render(viewpoint_camera, pc, env_map, pipe, scaling_modifier=1.0, override_color=None, flow_pkg=None, render_objmask=False,
           sh_factor_sink=None):
    """sh_factor_sink (extension, default off): adgs.dp.FactoredSHExchange.sink_for -- data-parallel training exchanges the SH
    gradients in factored form; the backward of this render then leaves them to FactoredSHExchange.reduce()."""
    n_pts = pc.get_pts_num if hasattr(pc, "get_pts_num") else pc.get_xyz.shape[0]
    device = (pc._scene_xyz if hasattr(pc, "_scene_xyz") else pc.get_xyz).device
    # the densification statistics read the gradient of the screen-space means from this tensor (.grad[:, :2])
    means2D = screenspace_points(n_pts, device)

    rasterizer = GaussianRasterizer(raster_settings=_camera_settings(viewpoint_camera, pc, pipe, scaling_modifier, device))
    pkg, flow_points = _deformed_state(pc, viewpoint_camera.time, flow_pkg, full_rows=override_color is not None)
    semantic = (pc.obj_mask_float if hasattr(pc, "obj_mask_float") else pc.get_obj_mask.float()[..., None]) if render_objmask else None
    # the environment-map background first: on the raw-SH path the blend epilogue composites it (`render = C + T * background`,
    # gaussian_renderer/__init__.py:93-94) and the blend backward returns dL/dbackground = T * dL/drender -- no element-wise pass
    background = env_map.get_image_background(viewpoint_camera) if env_map is not None else None
    (first, radii, depth, img_opacity, img_flow, img_semantic), composited = _rasterize(rasterizer, pc, pkg, means2D, override_color, flow_points,
                                                                                         semantic, sh_factor_sink, bg_image=background)
    if composited:
        rendered, foreground = first, None                      # 'foreground' on demand: render - (1 - O) * background
    else:
        foreground = first
        if background is None:
            background = torch.zeros_like(foreground)
        rendered = foreground + (1.0 - img_opacity) * background
    shs = pkg.get("shs")
    if shs is not None and not torch.is_tensor(shs) and getattr(shs, "scene_xyz", None) is not None:
        # raw-scene path: rows [0, Ns) of the deformed tensors were never written; hand them out lazily (a full deformation pass)
        from adgs import deform as _deform
        t_cam, flow_t = viewpoint_camera.time, (None if flow_pkg is None else flow_pkg[0])
        full = lambda key: (lambda: _deform.get_deformed_pkg(pc, t_cam, want=(key,))[key])
        out = _LazyResult({k: full(k) for k in ("xyz", "rotation", "opacity", "scales")})
        if flow_t is not None:
            out._thunks["flow_xyz"] = lambda: _deform.get_deformed_xyz(pc, flow_t)
        out["shs"] = shs
        opacity_entry = {}
    else:
        out = dict(pkg)                                         # the reference also hands back xyz / rotation / shs / opacity
        opacity_entry = dict(opacity=pkg["opacity"])
    out.update(opacity_entry)
    out.update(render=rendered, viewspace_points=means2D, visibility_filter=radii > 0, radii=radii,
               depth=depth.squeeze(0), img_opacity=img_opacity.squeeze(0),
               background=background, img_flow=img_flow if flow_points is not None else None,
               img_semantic=img_semantic if semantic is not None else None)
    if foreground is not None:
        out["foreground"] = foreground
    else:                                                       # composited in the epilogue: the reference's 'foreground' entry on first access
        if not isinstance(out, _LazyResult):
            lazy = _LazyResult({})
            lazy.update(out)
            out = lazy
        out._thunks["foreground"] = lambda: rendered - (1.0 - img_opacity) * background
    return out
