"""CPU restatement (NumPy, float64 or float32) of utils/loss_utils.py:20-68: l1_loss and ssim
(11x11 Gaussian window, sigma 1.5, zero padding, C1 = 0.01^2, C2 = 0.03^2), plus the hand-derived
gradient of both means w.r.t. the first image.

TEST INFRASTRUCTURE ONLY (tests/, smoke(), bench cpu baseline).  Pinned against the reference's own
Python (imported in this container, tests/golden/make_loss_golden.py -> tests/golden/loss_golden.npz:
values and reference-autograd gradients).
"""
import numpy as np


def gaussian_window(window_size=11, sigma=1.5, dtype=np.float32):
    """loss_utils.py:26-28: python-float exp, float32 tensor, divided by its float32 sum."""
    g = np.array([np.exp(-(x - window_size // 2) ** 2 / float(2 * sigma ** 2)) for x in range(window_size)], dtype=np.float32)
    return (g / g.sum(dtype=np.float32)).astype(dtype)


def _conv(x, w2):
    """Depthwise 'same' correlation with zero padding (F.conv2d(x, window, padding=5, groups=C)); x [..., H, W]."""
    k = w2.shape[0]; r = k // 2
    H, W = x.shape[-2:]
    xp = np.zeros(x.shape[:-2] + (H + 2 * r, W + 2 * r), dtype=x.dtype)
    xp[..., r:r + H, r:r + W] = x
    out = np.zeros_like(x)
    for i in range(k):
        for j in range(k):
            out += w2[i, j] * xp[..., i:i + H, j:j + W]
    return out


def l1_ssim(img1, img2, dtype=np.float64):
    """Returns (l1, ssim, d l1 / d img1, d ssim / d img1)."""
    x1, x2 = np.asarray(img1, dtype), np.asarray(img2, dtype)
    g = gaussian_window(dtype=np.float32)
    w2 = np.outer(g, g).astype(np.float32).astype(dtype)          # _1D_window.mm(_1D_window.t()).float()
    C1, C2 = dtype(0.01 ** 2), dtype(0.03 ** 2)
    mu1, mu2 = _conv(x1, w2), _conv(x2, w2)
    e11, e22, e12 = _conv(x1 * x1, w2), _conv(x2 * x2, w2), _conv(x1 * x2, w2)
    s1, s2, s12 = e11 - mu1 * mu1, e22 - mu2 * mu2, e12 - mu1 * mu2
    A1, A2, B1, B2 = 2 * mu1 * mu2 + C1, 2 * s12 + C2, mu1 * mu1 + mu2 * mu2 + C1, s1 + s2 + C2
    m = (A1 * A2) / (B1 * B2)
    n = x1.size
    # partials of the map w.r.t. mu1, E[x1^2], E[x1 x2] (the other two window means belong to the constant image)
    num, den = A1 * A2, B1 * B2
    d_mu1 = ((2 * mu2 * A2 - 2 * mu2 * A1) * den - num * (2 * mu1 * B2 - 2 * mu1 * B1)) / (den * den)
    d_e11 = -num / (B1 * B2 * B2)
    d_e12 = 2 * A1 / den
    g_ssim = (_conv(d_mu1, w2) + 2 * x1 * _conv(d_e11, w2) + x2 * _conv(d_e12, w2)) / n       # window is symmetric
    g_l1 = np.sign(x1 - x2) / n
    return np.abs(x1 - x2).mean(), m.mean(), g_l1, g_ssim


def depth_loss(pred, gt, mask=None, dtype=np.float64):
    """utils/loss_utils.py:70-75 over utils/depth_utils.py:3-45.  Returns (loss, d loss / d pred) -- the gradient includes the
    dependence of the least-squares scale and shift on the prediction, as the reference's autograd does."""
    p, g = np.asarray(pred, dtype), np.asarray(gt, dtype)
    m = np.ones_like(p) if mask is None or np.size(mask) == 0 else np.asarray(mask, dtype)
    a00, a01, a11 = (m * p * p).sum(), (m * p).sum(), m.sum()
    b0, b1 = (m * p * g).sum(), (m * g).sum()
    # the reference evaluates the determinant on fp32 sums: an exactly rank-deficient system is detected there
    f = np.float32
    det32 = f(a00) * f(a11) - f(a01) * f(a01)
    det = a00 * a11 - a01 * a01
    if det32 == 0:
        return np.abs(0 * p - g)[m > 0].sum() / a11 if mask is not None and np.size(mask) else np.abs(g).mean(), np.zeros_like(p)
    s, t = (a11 * b0 - a01 * b1) / det, (-a01 * b0 + a00 * b1) / det
    d = s * p + t - g
    sg = np.sign(d)
    loss = (np.abs(d) * m).sum() / a11
    A, B = (m * sg * p).sum(), (m * sg).sum()
    s_a00, s_a01, s_b0 = -s * a11 / det, (-b1 + 2 * a01 * s) / det, a11 / det
    t_a00, t_a01, t_b0 = (b1 - t * a11) / det, (-b0 + 2 * a01 * t) / det, -a01 / det
    grad = m * (sg * s + (A * s_a00 + B * t_a00) * 2 * p + (A * s_a01 + B * t_a01) + (A * s_b0 + B * t_b0) * g) / a11
    return loss, grad


def flow_loss(img_flow, flow, flow_vis, opacity, K, R, T, dist, dtype=np.float64):
    """utils/loss_utils.py:86-106 (get_flow_loss) over utils/flow_utils.py:5-10 (flow_points_project).
    img_flow [3,H,W] rendered 3-D flow points, flow [2,H,W] target pixel coordinates, flow_vis [H,W], opacity [H,W] or None.
    Returns (loss, d loss / d img_flow, d loss / d opacity or None); loss 0 when no pixel is selected (:92-93)."""
    f = np.asarray(img_flow, dtype); fl = np.asarray(flow, dtype); vis = np.asarray(flow_vis, dtype)
    K, R, T = np.asarray(K, dtype), np.asarray(R, dtype), np.asarray(T, dtype)
    _, H, W = f.shape
    sel = (vis > 0.5) & (fl[0] <= W - 1.0) & (fl[0] >= 0.0) & (fl[1] <= H - 1.0) & (fl[1] >= 0.0)
    n = int(sel.sum())
    g_f = np.zeros_like(f); g_o = None if opacity is None or np.size(opacity) == 0 else np.zeros((H, W), dtype)
    if n == 0:
        return 0.0, g_f, g_o
    w = sel.astype(dtype)
    if g_o is not None:
        w = w * np.asarray(opacity, dtype)
    M = K @ R
    p = np.einsum("ij,jhw->ihw", M, f) + (K @ T)[:, None, None]
    mask = p[2] > dist
    z = np.maximum(p[2], dist)
    u, v = p[0] / z, p[1] / z
    w = w * mask
    du, dv = u - fl[0], v - fl[1]
    per = (np.abs(du) / W + np.abs(dv) / H)
    loss = (per * w).sum() / n
    if g_o is not None:
        g_o = per * sel * mask / n
    # d/dp of u = x / z (z > dist where the weight is non-zero): (1/z, 0, -x/z^2)
    gu, gv = np.sign(du) * w / (W * n), np.sign(dv) * w / (H * n)
    gp = np.stack([gu / z, gv / z, -(gu * p[0] + gv * p[1]) / (z * z)])
    g_f = np.einsum("ij,ihw->jhw", M, gp)
    return loss, g_f, g_o


def bce_clip_loss(pred, target, lo=1e-3, hi=1.0 - 1e-3, invert=False, dtype=np.float64):
    """train.py:95-103: mean binary cross entropy of clip(pred, lo, hi) (or 1 - clip(...) when `invert`) against `target`;
    log terms clamped at -100 like torch.  Returns (loss, d loss / d pred)."""
    x = np.asarray(pred, dtype); t = np.asarray(target, dtype)
    c = np.clip(x, dtype(lo), dtype(hi))
    inside = (x >= dtype(lo)) & (x <= dtype(hi))
    q = 1.0 - c if invert else c
    lq, l1q = np.maximum(np.log(q), -100.0), np.maximum(np.log(1.0 - q), -100.0)
    loss = -(t * lq + (1.0 - t) * l1q).mean()
    dq = -(t / q - (1.0 - t) / (1.0 - q)) / x.size
    return loss, (-dq if invert else dq) * inside


def group_var_loss(x, idx, dtype=np.float64):
    """train.py:104-106 / 111-113: mean(sum(var(x[idx], dim=1), dim=-1)) with torch.var's unbiased estimator; x [N, ..., C],
    idx [G, K].  Returns (loss, d loss / d x)."""
    x = np.asarray(x, dtype)
    idx = np.asarray(idx, np.int64)
    G, K = idx.shape
    v = x[idx]                                         # [G, K, ..., C]
    mean = v.mean(axis=1, keepdims=True)
    var = ((v - mean) ** 2).sum(axis=1) / (K - 1)      # [G, ..., C]
    per = var.sum(axis=-1)                             # [G, ...]
    loss = per.mean()
    g = np.zeros_like(x)
    np.add.at(g, idx, 2.0 * (v - mean) / ((K - 1) * per.size))
    return float(loss), g


def sigma_loss(log_sigma, frame_gap, dtype=np.float64):
    """train.py:108-110: mean(|frame_gap / mean(exp(log_sigma), dim=-1)|); log_sigma [N, 2].  Returns (loss, d loss / d log_sigma)."""
    s = np.asarray(log_sigma, dtype)
    e = np.exp(s)
    m = e.mean(axis=-1)
    q = frame_gap / m
    loss = np.abs(q).mean()
    g = (-np.sign(q) * q / m / s.shape[0])[:, None] * (e / s.shape[1])
    return float(loss), g
