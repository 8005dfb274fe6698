"""Dense, differentiable torch (CPU, float64) restatement of the rasterizer forward.

Test infrastructure: a THIRD, structurally different implementation (all pixels x
all Gaussians, no tiles lists, autograd instead of hand-written backward) used to
pin the C++ oracle (oracle/raster_oracle.cpp):
  * forward images must agree with the oracle;
  * torch.autograd of this forward must agree with the oracle's hand-derived
    backward wherever the reference's backward is the true derivative (no frustum
    clamp active, grad_img_opacity = 0 -- the two documented quirks are pinned by
    their own tests).
Hard gates (cull, alpha<1/255, power>0, T stop, tile-rect membership) are taken
from the forward values and treated as constants, as the reference does.

Follows RAST/cuda_rasterizer/forward.cu:20-402 (math) with the conventions of
SURVEY.md section 8(a).
"""
import math

import numpy as np
import torch


def _f(x):
    """The reference's constants are float literals: use their float32 values exactly."""
    return float(np.float32(x))


SH_C0 = _f(0.28209479177387814)
SH_C1 = _f(0.4886025119029199)
SH_C2 = [_f(v) for v in (1.0925484305920792, -1.0925484305920792, 0.31539156525252005, -1.0925484305920792, 0.5462742152960396)]
SH_C3 = [_f(v) for v in (-0.5900435899266435, 2.890611442640554, -0.4570457994644658, 0.3731763325901154, -0.4570457994644658,
                         1.445305721320277, -0.5900435899266435)]
EPS7 = _f(0.0000001)
LOWPASS = _f(0.3)
CLAMP13 = _f(1.3)
NEAR = _f(0.2)
ALPHA_MAX = _f(0.99)
ALPHA_MIN = float(np.float32(1.0) / np.float32(255.0))
T_STOP = _f(0.0001)
EIG_FLOOR = _f(0.1)


def eval_sh(deg, sh, dirs):
    """forward.cu:20-71; sh [P,M,3], dirs [P,3] normalised."""
    x, y, z = dirs[:, 0:1], dirs[:, 1:2], dirs[:, 2:3]
    res = SH_C0 * sh[:, 0]
    if deg > 0:
        res = res - SH_C1 * y * sh[:, 1] + SH_C1 * z * sh[:, 2] - SH_C1 * x * sh[:, 3]
        if deg > 1:
            xx, yy, zz, xy, yz, xz = x * x, y * y, z * z, x * y, y * z, x * z
            res = (res + SH_C2[0] * xy * sh[:, 4] + SH_C2[1] * yz * sh[:, 5] + SH_C2[2] * (2 * zz - xx - yy) * sh[:, 6]
                   + SH_C2[3] * xz * sh[:, 7] + SH_C2[4] * (xx - yy) * sh[:, 8])
            if deg > 2:
                res = (res + SH_C3[0] * y * (3 * xx - yy) * sh[:, 9] + SH_C3[1] * xy * z * sh[:, 10]
                       + SH_C3[2] * y * (4 * zz - xx - yy) * sh[:, 11] + SH_C3[3] * z * (2 * zz - 3 * xx - 3 * yy) * sh[:, 12]
                       + SH_C3[4] * x * (4 * zz - xx - yy) * sh[:, 13] + SH_C3[5] * z * (xx - yy) * sh[:, 14]
                       + SH_C3[6] * x * (xx - 3 * yy) * sh[:, 15])
    return torch.clamp_min(res + 0.5, 0.0)


def quat_to_R(q):
    """forward.cu:126-138 (un-normalised r,x,y,z), returned as the usual row-major rotation."""
    r, x, y, z = q[:, 0], q[:, 1], q[:, 2], q[:, 3]
    R = torch.stack([
        1 - 2 * (y * y + z * z), 2 * (x * y - r * z), 2 * (x * z + r * y),
        2 * (x * y + r * z), 1 - 2 * (x * x + z * z), 2 * (y * z - r * x),
        2 * (x * z - r * y), 2 * (y * z + r * x), 1 - 2 * (x * x + y * y)], dim=1).reshape(-1, 3, 3)
    return R


def render_dense(means3D, means2D, opacities, shs, colors_precomp, scales, rotations, cov3D_precomp, flow_points, semantic,
                 bg, viewmatrix, projmatrix, campos, tanfovx, tanfovy, H, W, sh_degree, scale_modifier=1.0, inv_depth=False):
    """Returns (color[3,H,W], radii[P], depth[1,H,W], img_opacity[1,H,W], img_flow[3,H,W], img_semantic[D_S,H,W])."""
    dt = torch.float64
    P = means3D.shape[0]
    # the reference ABI passes these as C floats
    tanfovx, tanfovy, scale_modifier = _f(tanfovx), _f(tanfovy), _f(scale_modifier)
    V = viewmatrix.to(dt)     # transposed convention: p' = p_row @ V
    PM = projmatrix.to(dt)
    ones = torch.ones(P, 1, dtype=dt)
    p_h = torch.cat([means3D, ones], 1)
    p_view = (p_h @ V)[:, :3]
    p_hom = p_h @ PM
    p_w = 1.0 / (p_hom[:, 3:4] + EPS7)
    p_proj = p_hom[:, :3] * p_w
    if means2D is not None:
        # the reference returns d/d(NDC xy) in means2D.grad (backward.cu:515-516,634-635)
        p_proj = p_proj + torch.cat([means2D[:, :2], torch.zeros(P, 1, dtype=dt)], 1)
    visible = p_view[:, 2] > NEAR
    focal_x = W / (2.0 * tanfovx)
    focal_y = H / (2.0 * tanfovy)
    if cov3D_precomp is None:
        R = quat_to_R(rotations)
        S = torch.diag_embed(scale_modifier * scales)
        Mm = R @ S
        Sigma = Mm @ Mm.transpose(1, 2)
    else:
        c = cov3D_precomp
        Sigma = torch.stack([c[:, 0], c[:, 1], c[:, 2], c[:, 1], c[:, 3], c[:, 4], c[:, 2], c[:, 4], c[:, 5]], 1).reshape(-1, 3, 3)
    tz = p_view[:, 2]
    limx, limy = CLAMP13 * tanfovx, CLAMP13 * tanfovy
    txc = torch.clamp(p_view[:, 0] / tz, -limx, limx) * tz
    tyc = torch.clamp(p_view[:, 1] / tz, -limy, limy) * tz
    zero = torch.zeros_like(tz)
    J = torch.stack([focal_x / tz, zero, -(focal_x * txc) / (tz * tz),
                     zero, focal_y / tz, -(focal_y * tyc) / (tz * tz)], 1).reshape(-1, 2, 3)
    Rw = V[:3, :3].transpose(0, 1)           # world->cam rotation (rows r: v[r], v[4+r], v[8+r])
    T = J @ Rw
    cov = T @ Sigma @ T.transpose(1, 2)
    a = cov[:, 0, 0] + LOWPASS
    b = cov[:, 0, 1]
    c = cov[:, 1, 1] + LOWPASS
    det = a * c - b * b
    visible = visible & (det != 0)
    det_safe = torch.where(det != 0, det, torch.ones_like(det))
    conic = torch.stack([c / det_safe, -b / det_safe, a / det_safe], 1)
    mid = 0.5 * (a + c)
    lam = mid + torch.sqrt(torch.clamp_min(mid * mid - det, EIG_FLOOR))
    radius = torch.ceil(3.0 * torch.sqrt(lam)).detach()
    px = ((p_proj[:, 0] + 1.0) * W - 1.0) * 0.5
    py = ((p_proj[:, 1] + 1.0) * H - 1.0) * 0.5
    gx, gy = (W + 15) // 16, (H + 15) // 16
    pxd, pyd = px.detach(), py.detach()
    rminx = torch.clamp(torch.trunc((pxd - radius) / 16), 0, gx)
    rminy = torch.clamp(torch.trunc((pyd - radius) / 16), 0, gy)
    rmaxx = torch.clamp(torch.trunc((pxd + radius + 15) / 16), 0, gx)
    rmaxy = torch.clamp(torch.trunc((pyd + radius + 15) / 16), 0, gy)
    visible = visible & ((rmaxx - rminx) * (rmaxy - rminy) > 0)
    radii = torch.where(visible, radius, torch.zeros_like(radius)).to(torch.int32)
    if colors_precomp is not None:
        feat = colors_precomp
    elif shs is not None:
        d = means3D - campos.to(dt)[None]
        d = d / d.norm(dim=1, keepdim=True)
        feat = eval_sh(sh_degree, shs, d)
    else:
        feat = None
    # depth order, ties by index (stable sort in the reference)
    depth32 = tz.detach().to(torch.float32)
    order = torch.argsort(depth32, stable=True)
    order = order[visible[order]]
    ys, xs = torch.meshgrid(torch.arange(H, dtype=dt), torch.arange(W, dtype=dt), indexing="ij")
    pixx, pixy = xs.reshape(-1, 1), ys.reshape(-1, 1)            # [X,1]
    tilex, tiley = torch.floor(pixx / 16), torch.floor(pixy / 16)
    o = order
    dx = px[o][None, :] - pixx
    dy = py[o][None, :] - pixy
    power = -0.5 * (conic[o, 0][None] * dx * dx + conic[o, 2][None] * dy * dy) - conic[o, 1][None] * dx * dy
    alpha = torch.clamp_max(opacities[o, 0][None] * torch.exp(power), ALPHA_MAX)
    member = ((tilex >= rminx[o][None]) & (tilex < rmaxx[o][None]) & (tiley >= rminy[o][None]) & (tiley < rmaxy[o][None]))
    gate = member & (power.detach() <= 0) & (alpha.detach() >= ALPHA_MIN)
    a_eff = torch.where(gate, alpha, torch.zeros_like(alpha))
    Tincl = torch.cumprod(1 - a_eff, dim=1)
    # stop at the first (gated) entry whose test_T < 1e-4, exclusive
    stop = gate & (Tincl.detach() < T_STOP)
    stopped = torch.cumsum(stop.to(torch.int32), dim=1) > 0
    a_eff = torch.where(stopped, torch.zeros_like(a_eff), a_eff)
    Tincl = torch.cumprod(1 - a_eff, dim=1)
    Texcl = torch.cat([torch.ones(Tincl.shape[0], 1, dtype=dt), Tincl[:, :-1]], 1)
    w = a_eff * Texcl
    Tfinal = Tincl[:, -1] if Tincl.shape[1] > 0 else torch.ones(H * W, dtype=dt)
    def img(v):   # v [P,C]
        return (w @ v[o]).transpose(0, 1).reshape(-1, H, W)
    color = (img(feat) + Tfinal[None].reshape(1, H, W) * bg.to(dt)[:, None, None]) if feat is not None else torch.zeros(3, H, W, dtype=dt)
    dval = (1.0 / (tz + EPS7)) if inv_depth else tz
    depth = img(dval[:, None])
    img_opacity = (1 - Tfinal).reshape(1, H, W)
    img_flow = img(flow_points) if flow_points is not None else torch.zeros(3, H, W, dtype=dt)
    img_sem = img(semantic) if semantic is not None else torch.zeros(0, H, W, dtype=dt)
    return color, radii, depth, img_opacity, img_flow, img_sem
