"""NumPy oracle of the factored SH-gradient expansion (include/adgs_exchange.h).

TEST INFRASTRUCTURE ONLY: only tests/ may import this module.  The product path (ad-gs_amd/) never does.

Restates, per camera, the coefficient gradients of the reference's computeColorFromSH backward
(submodules/depth-diff-gaussian-rasterization/cuda_rasterizer/backward.cu:44-112: dRGBdsh_k * dL_dRGB with the clamp
mask of :38-41 already applied to dL_dRGB) and the chain through `shs = cat(dc + f_shs(t), rest)`
(scene/gaussian_model.py:198-205) into shs_deform_param (f_shs linear: utils/func_utils.py:121-156), summed over cameras.
Pinned by tests/test_oracle_exchange.py against the line-by-line C++ oracle's dL_dsh (oracle/raster_oracle.cpp).
"""
import numpy as np

SH_C0 = 0.28209479177387814
SH_C1 = 0.4886025119029199
SH_C2 = [1.0925484305920792, -1.0925484305920792, 0.31539156525252005, -1.0925484305920792, 0.5462742152960396]
SH_C3 = [-0.5900435899266435, 2.890611442640554, -0.4570457994644658, 0.3731763325901154, -0.4570457994644658, 1.445305721320277,
         -0.5900435899266435]


def sh_coef_factors(deg, dirs):
    """dRGB/dsh_k for unit directions dirs[P,3] -> [P,16] (backward.cu:44-112); 0 above the active degree."""
    d = np.asarray(dirs, np.float64)
    x, y, z = d[:, 0], d[:, 1], d[:, 2]
    out = np.zeros((d.shape[0], 16), np.float64)
    out[:, 0] = SH_C0
    if deg > 0:
        out[:, 1], out[:, 2], out[:, 3] = -SH_C1 * y, SH_C1 * z, -SH_C1 * x
        if deg > 1:
            xx, yy, zz, xy, yz, xz = x * x, y * y, z * z, x * y, y * z, x * z
            out[:, 4] = SH_C2[0] * xy
            out[:, 5] = SH_C2[1] * yz
            out[:, 6] = SH_C2[2] * (2.0 * zz - xx - yy)
            out[:, 7] = SH_C2[3] * xz
            out[:, 8] = SH_C2[4] * (xx - yy)
            if deg > 2:
                out[:, 9] = SH_C3[0] * y * (3.0 * xx - yy)
                out[:, 10] = SH_C3[1] * xy * z
                out[:, 11] = SH_C3[2] * y * (4.0 * zz - xx - yy)
                out[:, 12] = SH_C3[3] * z * (2.0 * zz - 3.0 * xx - 3.0 * yy)
                out[:, 13] = SH_C3[4] * x * (4.0 * zz - xx - yy)
                out[:, 14] = SH_C3[5] * z * (xx - yy)
                out[:, 15] = SH_C3[6] * x * (xx - 3.0 * yy)
    return out


def expand(cams, W, C, P, Ns, row0, xyz_head, D, M):
    """cams = [(rgb[P,3], xyz_tail[P-row0,3] | None, campos[3])], W [n,C] dense basis weights.
    Returns float64 (scene_dc[Ns,1,3], obj_dc, scene_rest[Ns,M-1,3], obj_rest, scene_deform[Ns,3,C], obj_deform)."""
    dsh = np.zeros((P, 16, 3), np.float64)
    dsp = np.zeros((P, 3, max(C, 0)), np.float64)
    for c, (rgb, tail, campos) in enumerate(cams):
        rgb = np.asarray(rgb, np.float64).reshape(P, 3)
        means = np.zeros((P, 3), np.float64)
        if row0 > 0:
            means[:row0] = np.asarray(xyz_head, np.float64).reshape(-1, 3)[:row0]
        if row0 < P:
            means[row0:] = np.asarray(tail, np.float64).reshape(P - row0, 3)
        live = np.any(rgb != 0, axis=1)
        o = means - np.asarray(campos, np.float64)[None]
        with np.errstate(invalid="ignore", divide="ignore"):
            dirs = o / np.linalg.norm(o, axis=1, keepdims=True)
        coef = sh_coef_factors(D, dirs)
        coef[~live] = 0.0
        dsh += coef[:, :, None] * rgb[:, None, :]
        if C > 0:
            dsp += (SH_C0 * rgb)[:, :, None] * np.asarray(W, np.float64)[c][None, None, :C]
    dc, rest = dsh[:, :1], dsh[:, 1:M]
    return dc[:Ns], dc[Ns:], rest[:Ns], rest[Ns:], dsp[:Ns], dsp[Ns:]
