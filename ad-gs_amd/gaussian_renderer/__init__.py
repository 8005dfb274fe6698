"""`gaussian_renderer.render()` for AD-GS on MI355X -- same signature and result dict as the
reference (gaussian_renderer/__init__.py:18-115), built on the HIP rasterizer.

`pc` is any object with the reference GaussianModel's render-time getters
(scene/gaussian_model.py:88-231): get_xyz, get_deformed_xyz(t), get_deformed_pkg(t),
get_scaling, get_obj_mask, active_sh_degree.  `env_map` needs get_image_background(camera)
(scene/env.py:44-76) or may be None (black background).
"""
import math

import torch

from diff_gaussian_rasterization import GaussianRasterizationSettings, GaussianRasterizer


def render(viewpoint_camera, pc, env_map, pipe, scaling_modifier=1.0, override_color=None, flow_pkg=None, render_objmask=False):
    xyz0 = pc.get_xyz
    device = xyz0.device
    # gradient carrier for the screen-space means (densification statistics read .grad[:, :2])
    screenspace_points = torch.zeros_like(xyz0, dtype=xyz0.dtype, requires_grad=True, device=device) + 0
    try:
        screenspace_points.retain_grad()
    except Exception:
        pass

    settings = GaussianRasterizationSettings(
        image_height=int(viewpoint_camera.image_height),
        image_width=int(viewpoint_camera.image_width),
        tanfovx=math.tan(viewpoint_camera.FoVx * 0.5),
        tanfovy=math.tan(viewpoint_camera.FoVy * 0.5),
        bg=torch.zeros(3, device=device, dtype=torch.float32),
        scale_modifier=scaling_modifier,
        viewmatrix=viewpoint_camera.world_view_transform.to(device),
        projmatrix=viewpoint_camera.full_proj_transform.to(device),
        sh_degree=pc.active_sh_degree,
        campos=viewpoint_camera.camera_center.to(device),
        prefiltered=False,
        inv_depth=pipe.inv_depth,
        debug=pipe.debug)
    rasterizer = GaussianRasterizer(raster_settings=settings)

    flow_points = None
    if flow_pkg is not None and getattr(pc, "supports_fused_flow", False):
        # one pass over the deformation rows for both time stamps
        deform_pkg = pc.get_deformed_pkg(viewpoint_camera.time, flow_time=flow_pkg[0])
        flow_points = deform_pkg['flow_xyz']
    else:
        if flow_pkg is not None:
            flow_points = pc.get_deformed_xyz(flow_pkg[0])      # world positions at the other time stamp
        deform_pkg = pc.get_deformed_pkg(viewpoint_camera.time)
    semantic = pc.get_obj_mask.float()[..., None] if render_objmask else None

    shs_in = deform_pkg['shs'] if override_color is None else None
    if shs_in is not None and not torch.is_tensor(shs_in):
        # raw-SH fast path (pc.get_deformed_pkg returned a RawSH: the [N,16,3] tensor is never materialised)
        foreground, radii, depth, img_opacity, img_flow, img_semantic = rasterizer.forward_rawsh(
            deform_pkg['xyz'], screenspace_points, deform_pkg['opacity'], shs_in,
            deform_pkg['scales'] if 'scales' in deform_pkg else pc.get_scaling, deform_pkg['rotation'],
            flow_points=flow_points, semantic=semantic)
    else:
        foreground, radii, depth, img_opacity, img_flow, img_semantic = rasterizer(
            means3D=deform_pkg['xyz'], means2D=screenspace_points, shs=shs_in, colors_precomp=override_color,
            opacities=deform_pkg['opacity'], scales=deform_pkg['scales'] if 'scales' in deform_pkg else pc.get_scaling,
            rotations=deform_pkg['rotation'], flow_points=flow_points, semantic=semantic)

    if env_map is not None:
        background = env_map.get_image_background(viewpoint_camera)
    else:
        background = torch.zeros_like(foreground)
    rendered_image = foreground + (1.0 - img_opacity) * background

    res = {
        "render": rendered_image,
        "viewspace_points": screenspace_points,
        "visibility_filter": radii > 0,
        "radii": radii,
        "depth": depth.squeeze(0),
        "opacity": deform_pkg['opacity'],
        "img_opacity": img_opacity.squeeze(0),
        "foreground": foreground,
        "background": background,
        "img_flow": img_flow if flow_points is not None else None,
        "img_semantic": img_semantic if semantic is not None else None,
    }
    res.update(deform_pkg)
    return res
