"""Deterministic synthetic scenes for parity tests and bench.py.

Everything is generated on the CPU with a seeded torch.Generator and returned as
CPU float32 tensors; callers move them to the device.  Camera conventions follow
the reference: `viewmatrix`/`projmatrix` are the TRANSPOSED 4x4 (row-vector
convention, scene/cameras.py:77-80), `projmatrix = view @ proj`
(utils/graphics_utils.py:60-80 for the projection).
"""
import math

import torch

# BASELINE.json configs -> (P, W, H, focal_px, sh_degree, n_objects, seed)
CONFIGS = {
    "C1": dict(P=10_000, W=400, H=300, focal=300.0, sh_degree=0, n_objects=0, seed=0),
    "C2": dict(P=300_000, W=1242, H=375, focal=721.5, sh_degree=3, n_objects=0, seed=1),
    "C3": dict(P=1_000_000, W=1920, H=1280, focal=2050.0, sh_degree=3, n_objects=8, seed=2),
    "C4": dict(P=1_000_000, W=1920, H=1280, focal=2050.0, sh_degree=3, n_objects=8, seed=3),
    "C5": dict(P=3_000_000, W=1920, H=1280, focal=2050.0, sh_degree=3, n_objects=16, seed=4),
    # not a BASELINE.json config: a small dynamic scene for multi-rank dry runs of bench.py (tests/test_gpu_bench_multirank.py)
    "T3": dict(P=20_000, W=320, H=208, focal=340.0, sh_degree=3, n_objects=2, seed=7),
}


def projection_matrix(znear, zfar, fovx, fovy):
    """utils/graphics_utils.py:60-80 (returns the un-transposed P)."""
    tan_y = math.tan(fovy / 2)
    tan_x = math.tan(fovx / 2)
    top = tan_y * znear
    bottom = -top
    right = tan_x * znear
    left = -right
    P = torch.zeros(4, 4)
    P[0, 0] = 2.0 * znear / (right - left)
    P[1, 1] = 2.0 * znear / (top - bottom)
    P[0, 2] = (right + left) / (right - left)
    P[1, 2] = (top + bottom) / (top - bottom)
    P[3, 2] = 1.0
    P[2, 2] = zfar / (zfar - znear)
    P[2, 3] = -(zfar * znear) / (zfar - znear)
    return P


def make_camera(W, H, focal, cam_seed=None):
    """Camera at the origin looking down +z (view = identity); an optional seed
    yaw/pitch/translation-jitters it (used for the multi-camera configs C4/C5)."""
    fovx = 2.0 * math.atan(W / (2.0 * focal))
    fovy = 2.0 * math.atan(H / (2.0 * focal))
    w2c = torch.eye(4)
    if cam_seed is not None:
        g = torch.Generator().manual_seed(100 + int(cam_seed))
        yaw, pitch = ((torch.rand(2, generator=g) - 0.5) * 0.08).tolist()
        t = ((torch.rand(3, generator=g) - 0.5) * torch.tensor([1.0, 0.2, 1.0])).tolist()
        cy, sy, cp, sp = math.cos(yaw), math.sin(yaw), math.cos(pitch), math.sin(pitch)
        Ry = torch.tensor([[cy, 0, sy], [0, 1, 0], [-sy, 0, cy]])
        Rx = torch.tensor([[1, 0, 0], [0, cp, -sp], [0, sp, cp]])
        w2c[:3, :3] = Rx @ Ry
        w2c[:3, 3] = torch.tensor(t)
    view = w2c.transpose(0, 1).contiguous()                      # world_view_transform
    proj = projection_matrix(0.01, 100.0, fovx, fovy).transpose(0, 1)
    full = (view.unsqueeze(0).bmm(proj.unsqueeze(0))).squeeze(0).contiguous()
    campos = view.inverse()[3, :3].contiguous()
    return dict(W=W, H=H, tanfovx=math.tan(fovx * 0.5), tanfovy=math.tan(fovy * 0.5),
                viewmatrix=view.float(), projmatrix=full.float(), campos=campos.float(), fovx=fovx, fovy=fovy)


def make_scene(P, W, H, focal, sh_degree=3, seed=0, n_objects=0, with_flow=True, with_semantic=True,
               scale_mult=0.004, near_frac=0.01):
    """Static (already deformed/activated) rasterizer inputs, SURVEY.md 8(d)."""
    g = torch.Generator().manual_seed(int(seed))
    cam = make_camera(W, H, focal)
    tx, ty = cam["tanfovx"], cam["tanfovy"]
    z = torch.rand(P, generator=g) * 78.0 + 2.0
    n_near = int(P * near_frac)
    if n_near > 0:
        z[:n_near] = torch.rand(n_near, generator=g) * 5.2 - 5.0          # U(-5, 0.2): near-culled
    x = (torch.rand(P, generator=g) * 2.2 - 1.1) * z * tx
    y = (torch.rand(P, generator=g) * 2.2 - 1.1) * z * ty
    xyz = torch.stack([x, y, z], dim=1)
    zc = z.abs().clamp_min(0.5)
    log_scale = torch.log(scale_mult * zc)[:, None] + 0.6 * torch.randn(P, 3, generator=g)
    scales = torch.exp(log_scale)
    rot = torch.nn.functional.normalize(torch.randn(P, 4, generator=g), dim=1)
    opacity = torch.sigmoid(1.5 * torch.randn(P, 1, generator=g))
    M = (sh_degree + 1) ** 2 if sh_degree >= 0 else 0
    Mfull = 16 if sh_degree == 3 else M
    shs = torch.zeros(P, Mfull, 3)
    shs[:, 0] = torch.randn(P, 3, generator=g)
    if Mfull > 1:
        shs[:, 1:] = 0.15 * torch.randn(P, Mfull - 1, 3, generator=g)
    obj_mask = torch.zeros(P, dtype=torch.bool)
    if n_objects > 0:
        n_obj = P // 5
        obj_mask[P - n_obj:] = True        # objects last, as in GaussianModel (scene || obj)
        centers = torch.stack([
            (torch.rand(n_objects, generator=g) * 1.6 - 0.8) * 30.0 * tx,
            torch.full((n_objects,), 1.5),
            torch.rand(n_objects, generator=g) * 50.0 + 8.0], dim=1)
        which = torch.randint(0, n_objects, (n_obj,), generator=g)
        offs = torch.randn(n_obj, 3, generator=g)
        offs = offs / offs.norm(dim=1, keepdim=True).clamp_min(1e-6) * (torch.rand(n_obj, 1, generator=g) ** (1 / 3)) * 2.0
        xyz[P - n_obj:] = centers[which] + offs
    scene = dict(cam)
    scene.update(P=P, sh_degree=sh_degree, means3D=xyz.float().contiguous(), scales=scales.float().contiguous(),
                 rotations=rot.float().contiguous(), opacities=opacity.float().contiguous(), shs=shs.float().contiguous(),
                 bg=torch.zeros(3), obj_mask=obj_mask)
    if with_semantic:
        scene["semantic"] = obj_mask.float()[:, None].contiguous()
    if with_flow:
        scene["flow_points"] = (xyz + 0.05 * torch.randn(P, 3, generator=g)).float().contiguous()
    return scene


def make_config_scene(name, **overrides):
    cfg = dict(CONFIGS[name])
    cfg.update(overrides)
    return make_scene(cfg["P"], cfg["W"], cfg["H"], cfg["focal"], sh_degree=cfg["sh_degree"], seed=cfg["seed"],
                      n_objects=cfg["n_objects"])


def make_upstream_grads(scene, seed=0, D_S=1):
    """Upstream image gradients ~ N(0,1)/(H*W)  (SURVEY.md 8(d))."""
    g = torch.Generator().manual_seed(1000 + int(seed))
    H, W = scene["H"], scene["W"]
    n = float(H * W)
    return dict(color=torch.randn(3, H, W, generator=g) / n, depth=torch.randn(1, H, W, generator=g) / n,
                img_opacity=torch.randn(1, H, W, generator=g) / n, flow=torch.randn(3, H, W, generator=g) / n,
                semantic=torch.randn(D_S, H, W, generator=g) / n)


# cam0 -> world rotation of a z-up world whose +x is the canonical camera's viewing direction (x right, y down, z forward ->
# X forward, Y left, Z up), and its quaternion (w, x, y, z)
Z_UP = torch.tensor([[0.0, 0.0, 1.0], [-1.0, 0.0, 0.0], [0.0, -1.0, 0.0]])
Z_UP_QUAT = (0.5, -0.5, 0.5, -0.5)


def to_z_up_world(scene):
    """The same scene expressed in a driving-dataset world frame (z up, the canonical camera looks along +x) instead of the
    canonical camera's own frame: positions, flow points and the Gaussians' orientations are rotated, the camera becomes
    `camera_to_z_up(camera)`.  The rendered images are the same up to rounding; what changes is everything that depends on WORLD
    directions -- the environment map (scene/env.py:63-76: elevation = angle to the world's xy-plane, so a camera looking along
    world +z stares at the map's pole)."""
    out = dict(scene)
    A = Z_UP
    out["means3D"] = (scene["means3D"] @ A.t()).contiguous()
    if "flow_points" in scene:
        out["flow_points"] = (scene["flow_points"] @ A.t()).contiguous()
    aw, ax, ay, az = Z_UP_QUAT
    q = scene["rotations"]
    w, x, y, z = q[:, 0], q[:, 1], q[:, 2], q[:, 3]
    out["rotations"] = torch.stack([aw * w - ax * x - ay * y - az * z, aw * x + ax * w + ay * z - az * y,
                                    aw * y - ax * z + ay * w + az * x, aw * z + ax * y - ay * x + az * w], 1).contiguous()
    cam = camera_to_z_up(scene)
    out.update({k: cam[k] for k in ("viewmatrix", "projmatrix", "campos")})
    return out


def camera_to_z_up(cam):
    """make_camera()'s dict with the world rotated as in to_z_up_world (world -> camera: first back into the canonical frame)."""
    out = dict(cam)
    w2c = cam["viewmatrix"].t().double().clone()
    w2c[:3, :3] = w2c[:3, :3] @ Z_UP.t().double()
    view = w2c.t().contiguous()
    proj = projection_matrix(0.01, 100.0, cam["fovx"], cam["fovy"]).transpose(0, 1).double()
    out["viewmatrix"] = view.float()
    out["projmatrix"] = (view @ proj).float().contiguous()
    out["campos"] = view.inverse()[3, :3].float().contiguous()
    return out


class CameraObject:
    """The attributes gaussian_renderer.render() reads from the reference Camera (scene/cameras.py:17-100)."""

    def __init__(self, W, H, fovx, fovy, view, full_proj, center, time):
        self.image_width, self.image_height = W, H
        self.FoVx, self.FoVy = fovx, fovy
        self.world_view_transform, self.full_proj_transform, self.camera_center = view, full_proj, center
        self.time = time


def camera_object(cam, time=0.37):
    """Wrap a make_camera()/make_scene() dict as a Camera-like object."""
    return CameraObject(cam["W"], cam["H"], cam["fovx"], cam["fovy"], cam["viewmatrix"], cam["projmatrix"], cam["campos"], time)
