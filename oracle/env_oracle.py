"""CPU restatement (NumPy) of the environment-map background of scene/env.py:11-76:
pixel -> camera ray (K^-1, normalise) -> world ray (world_view_transform[:3,:3] @ ray, normalise)
-> (azimuth, elevation) (utils/graphics_utils.py:95-100) -> scaled by (1/pi, 2/pi)
-> bilinear grid_sample (align_corners=True, zero padding) of grid_map -> sigmoid.
Plus the gradient w.r.t. grid_map for an upstream dL/dbackground.

TEST INFRASTRUCTURE ONLY.  Pinned against the reference's own Python (tests/golden/make_env_golden.py ->
tests/golden/env_golden.npz: backgrounds and reference-autograd gradients).
"""
import numpy as np


def sample_coords(H, W, focal, R, Hm, Wm, dtype=np.float64):
    """Continuous texel coordinates (ix, iy) [H, W] of every pixel."""
    f = dtype(focal)
    xs, ys = np.meshgrid(np.arange(W, dtype=dtype), np.arange(H, dtype=dtype), indexing="xy")
    ray = np.stack([(xs - dtype(W) / 2) / f, (ys - dtype(H) / 2) / f, np.ones_like(xs)], -1)     # K^-1 [x, y, 1]
    ray = ray / np.maximum(np.linalg.norm(ray, axis=-1, keepdims=True), 1e-12)
    ray = ray @ np.asarray(R, dtype).T                                                            # R @ ray
    ray = ray / np.maximum(np.linalg.norm(ray, axis=-1, keepdims=True), 1e-12)
    az = np.arctan2(ray[..., 1], ray[..., 0]); el = np.arctan2(ray[..., 2], np.hypot(ray[..., 0], ray[..., 1]))
    gx, gy = az * dtype(1.0 / np.pi), el * dtype(2.0 / np.pi)
    return (gx + 1) / 2 * (Wm - 1), (gy + 1) / 2 * (Hm - 1)


def _corners(ix, iy, Hm, Wm):
    x0, y0 = np.floor(ix).astype(np.int64), np.floor(iy).astype(np.int64)
    fx, fy = ix - x0, iy - y0
    for dy, wy in ((0, 1 - fy), (1, fy)):
        for dx, wx in ((0, 1 - fx), (1, fx)):
            xx, yy = x0 + dx, y0 + dy
            ok = (xx >= 0) & (xx < Wm) & (yy >= 0) & (yy < Hm)                                    # padding_mode='zeros'
            yield xx, yy, wx * wy, ok


def background(grid_map, H, W, focal, R, dtype=np.float64):
    """grid_map [C, Hm, Wm] -> sigmoid(bilinear sample) [C, H, W]."""
    gm = np.asarray(grid_map, dtype)
    C, Hm, Wm = gm.shape
    ix, iy = sample_coords(H, W, focal, R, Hm, Wm, dtype)
    raw = np.zeros((C, H, W), dtype)
    for xx, yy, w, ok in _corners(ix, iy, Hm, Wm):
        raw += np.where(ok, w, 0)[None] * gm[:, np.clip(yy, 0, Hm - 1), np.clip(xx, 0, Wm - 1)]
    return 1.0 / (1.0 + np.exp(-raw))


def background_grad(grid_map, H, W, focal, R, g_bg, dtype=np.float64):
    """d(sum g_bg * background) / d grid_map."""
    gm = np.asarray(grid_map, dtype)
    C, Hm, Wm = gm.shape
    bg = background(gm, H, W, focal, R, dtype)
    g_raw = np.asarray(g_bg, dtype) * bg * (1 - bg)
    ix, iy = sample_coords(H, W, focal, R, Hm, Wm, dtype)
    out = np.zeros_like(gm)
    for xx, yy, w, ok in _corners(ix, iy, Hm, Wm):
        for c in range(C):
            np.add.at(out[c], (np.clip(yy, 0, Hm - 1), np.clip(xx, 0, Wm - 1)), np.where(ok, w, 0) * g_raw[c])
    return out
