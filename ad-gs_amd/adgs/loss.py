"""Photometric loss on the HIP path: drop-ins for
utils/loss_utils.py `l1_loss` (:20-21) and `ssim` (:37-68) as used at train.py:79-80.

`l1_ssim(image, gt)` evaluates both in ONE forward kernel and back-propagates both in ONE backward
kernel (adgs_l1_ssim_forward / _backward, include/adgs_loss.h); `l1_loss` / `ssim` keep the reference's
names and signatures on top of it.  Images are [..., C, H, W] fp32 on a HIP device, `gt` is a constant.
There is no CPU fallback.
"""
import ctypes
import threading

import torch

from . import _lib

SLOTS = 256        # ADGS_LOSS_SLOTS


def _stream(dev):
    return _lib.stream_ptr(dev)


class _Slice:
    """Ownership of one slice of a _WorkArena (or of nothing: a fresh buffer).  The slot is free again when the token is dropped -- at once
    for terms whose backward does not read the slice, with the autograd context for the terms whose backward does (depth, flow).  A token
    dropped before done() (a launch failed between the sum kernel and the finish kernel that re-zeroes the slot rows) marks its slice
    dirty: it is zero-filled before it is handed out again."""
    __slots__ = ("arena", "i", "clean")

    def __init__(self, arena, i):
        self.arena, self.i, self.clean = arena, i, False

    def done(self):
        self.clean = True

    def __del__(self):
        a = self.arena
        if a is not None:
            if not self.clean:
                a.dirty.add(self.i)
            a.busy[self.i] = False


class _WorkArena:
    """Zero-initialised work buffers of the loss kernels (include/adgs_loss.h) without a fill per call: the kernel that consumes a
    buffer's slot rows leaves them zero, so a buffer only has to be zeroed when it is created.  One arena per (device, stream, layout):
    N slices; a slice is busy while its _Slice token lives (a training iteration holds two across its backward: depth and flow).  When
    every slice is busy -- gradient accumulation over more terms than the ring holds -- or under stream capture (an arena created there
    would have its zero fill baked into the graph) the caller gets a fresh zero-filled buffer instead.  take() is guarded by a lock (two
    threads may share a stream)."""
    N = 64

    def __init__(self, device, doubles):
        self.buf = torch.zeros(self.N, doubles, dtype=torch.float64, device=device)
        self.busy = [False] * self.N
        self.dirty = set()
        self.next = 0
        self.lock = threading.Lock()

    def take(self):
        with self.lock:
            for k in range(self.N):
                i = (self.next + k) % self.N
                if not self.busy[i]:
                    break
            else:
                return None
            self.next = (i + 1) % self.N
            self.busy[i] = True
            spoiled = i in self.dirty
            self.dirty.discard(i)
        if spoiled:
            self.buf[i].zero_()
        return self.buf[i], _Slice(self, i)


_ARENAS = {}


def _work(device, doubles):
    """(zeroed work buffer of `doubles` doubles for one loss term, its _Slice token: call done() after the term's launches)"""
    if not torch.cuda.is_current_stream_capturing():
        key = (device, _lib.stream_ptr(device).value, doubles)
        a = _ARENAS.get(key)
        if a is None:
            if len(_ARENAS) > 48:
                _ARENAS.clear()
            a = _ARENAS[key] = _WorkArena(device, doubles)
        got = a.take()
        if got is not None:
            return got
    t = _Slice(None, -1)
    return torch.zeros(doubles, dtype=torch.float64, device=device), t


class _L1SSIM(torch.autograd.Function):
    @staticmethod
    def forward(ctx, image, gt):
        if not image.is_cuda or not gt.is_cuda:
            raise RuntimeError("l1_ssim: tensors must be on a HIP device; there is no CPU path")
        if image.shape != gt.shape or image.dim() < 3:
            raise ValueError("l1_ssim: image and gt must have the same [..., C, H, W] shape")
        img, ref = image.contiguous().float(), gt.contiguous().float()
        H, W = img.shape[-2:]
        planes = img.numel() // (H * W) if H * W else 0
        n = img.numel()
        need = ctx.needs_input_grad[0]
        sums, tok = _work(img.device, 2 * SLOTS)                   # spread atomics (include/adgs_loss.h); consumed by adgs_l1_ssim_means below
        maps = [torch.empty_like(img) for _ in range(3)] if need else [None] * 3
        means = torch.empty(2, dtype=torch.float32, device=img.device) if n else torch.zeros(2, dtype=torch.float32, device=img.device)
        if n:
            with torch.cuda.device(img.device):
                _lib.check(_lib.lib().adgs_l1_ssim_forward(planes, H, W, img.data_ptr(), ref.data_ptr(), sums.data_ptr(),
                                                           *[m.data_ptr() if m is not None else None for m in maps], _stream(img.device)),
                           "adgs_l1_ssim_forward")
                _lib.check(_lib.lib().adgs_l1_ssim_means(sums.data_ptr(), n, means.data_ptr(), _stream(img.device)), "adgs_l1_ssim_means")
        tok.done()
        if need:
            ctx.save_for_backward(img, ref, *maps)
        ctx.dims = (planes, H, W)
        return means[0], means[1]

    @staticmethod
    def backward(ctx, g_l1, g_ssim):
        img, ref, d_mu1, d_e11, d_e12 = ctx.saved_tensors
        planes, H, W = ctx.dims
        out = torch.empty_like(img)
        gl = g_l1.reshape(1).float().contiguous() if g_l1 is not None else None
        gs = g_ssim.reshape(1).float().contiguous() if g_ssim is not None else None
        if img.numel():
            with torch.cuda.device(img.device):
                _lib.check(_lib.lib().adgs_l1_ssim_backward(planes, H, W, img.data_ptr(), ref.data_ptr(), d_mu1.data_ptr(), d_e11.data_ptr(),
                                                            d_e12.data_ptr(), gl.data_ptr() if gl is not None else None,
                                                            gs.data_ptr() if gs is not None else None, out.data_ptr(), _stream(img.device)),
                           "adgs_l1_ssim_backward")
        return out, None


def l1_ssim(image, gt):
    """(mean |image - gt|, mean SSIM(image, gt)) -- both differentiable w.r.t. `image`."""
    return _L1SSIM.apply(image, gt.detach())


def l1_loss(network_output, gt):
    """utils/loss_utils.py:20-21."""
    return l1_ssim(network_output, gt)[0]


def ssim(img1, img2, window_size=11, size_average=True):
    """utils/loss_utils.py:37-68 for the arguments AD-GS uses (11x11 window, mean over everything)."""
    if window_size != 11 or not size_average:
        raise NotImplementedError("the HIP ssim implements window_size=11, size_average=True (train.py:80)")
    return l1_ssim(img1, img2)[1]


def photometric_loss(image, gt, lambda_dssim, lambda_l1=1.0):
    """train.py:79-80,112: (1 - lambda_dssim) * lambda_l1 * L1 + lambda_dssim * (1 - SSIM); returns (loss, Ll1, dssim_loss)."""
    l1, s = l1_ssim(image, gt)
    dssim = 1.0 - s
    return (1.0 - lambda_dssim) * lambda_l1 * l1 + lambda_dssim * dssim, l1, dssim


DEPTH_WORK_DOUBLES = 256 * 8 + 16        # ADGS_DEPTH_WORK_DOUBLES


class _DepthLoss(torch.autograd.Function):
    @staticmethod
    def forward(ctx, pred, gt, mask):
        if not pred.is_cuda:
            raise RuntimeError("get_depth_loss: tensors must be on a HIP device; there is no CPU path")
        p, g = pred.contiguous().float(), gt.contiguous().float()
        m = None if mask is None else mask.contiguous().float()
        if p.shape != g.shape or (m is not None and m.shape != p.shape):
            raise ValueError("get_depth_loss: prediction, target and mask must have the same shape")
        work, ctx.token = _work(p.device, DEPTH_WORK_DOUBLES)
        out = torch.empty(1, dtype=torch.float32, device=p.device) if p.numel() else torch.zeros(1, dtype=torch.float32, device=p.device)
        with torch.cuda.device(p.device):
            _lib.check(_lib.lib().adgs_depth_loss_forward(p.numel(), p.data_ptr(), g.data_ptr(), m.data_ptr() if m is not None else None,
                                                          work.data_ptr(), out.data_ptr(), _stream(p.device)), "adgs_depth_loss_forward")
        ctx.token.done()
        ctx.save_for_backward(p, g, work, *([m] if m is not None else []))
        return out[0]

    @staticmethod
    def backward(ctx, g_loss):
        p, g, work, *rest = ctx.saved_tensors
        m = rest[0] if rest else None
        out = torch.empty_like(p)
        gl = g_loss.reshape(1).float().contiguous()
        with torch.cuda.device(p.device):
            _lib.check(_lib.lib().adgs_depth_loss_backward(p.numel(), p.data_ptr(), g.data_ptr(), m.data_ptr() if m is not None else None,
                                                           work.data_ptr(), gl.data_ptr(), out.data_ptr(), _stream(p.device)), "adgs_depth_loss_backward")
        return out, None, None


def get_depth_loss(pred, gt, mask=None):
    """utils/loss_utils.py:70-75: L1 between the scale/shift-aligned prediction (utils/depth_utils.py:3-45, closed-form least
    squares) and the target, averaged over the mask; differentiable w.r.t. `pred` through the scale and the shift.  No host
    synchronisation (the reference's `if det == 0` is one)."""
    return _DepthLoss.apply(pred, gt.detach(), None if mask is None else mask.detach())


AUX_WORK_DOUBLES = 256 * 2 + 2           # ADGS_AUX_WORK_DOUBLES


class _FlowCam:
    """K, R, T of the flow target (flow_pkg[1:4], train.py:68-71) as kernel arguments.  Tensors on the render device are handed to the
    `_devcam` entry points as device pointers (the kernels form K R and K T themselves): nothing is read back, and -- unlike a host-side
    copy keyed by storage address, which the caching allocator re-issues to the next iteration's `a.cuda()` -- nothing can go stale.
    CPU tensors go by value through the host entry points."""

    def __init__(self, K, R, T, device):
        ts = (K, R, T)
        for t, n, name in zip(ts, (9, 9, 3), "KRT"):
            if t.numel() != n:
                raise ValueError("flow camera: %s must have %d elements" % (name, n))
        self.on_device = all(t.is_cuda and t.device == device for t in ts)
        if self.on_device:
            self.keep = tuple(t.detach().contiguous().float() for t in ts)          # alive until the backward has been enqueued
            self.args = tuple(t.data_ptr() for t in self.keep)
        else:                                                                          # a tensor on another device is read back (correct, slow)
            self.keep = None
            self.args = tuple((ctypes.c_float * n)(*[float(x) for x in t.detach().reshape(-1).tolist()]) for t, n in zip(ts, (9, 9, 3)))

    def forward_fn(self):
        return _lib.lib().adgs_flow_loss_forward_devcam if self.on_device else _lib.lib().adgs_flow_loss_forward

    def backward_fn(self):
        return _lib.lib().adgs_flow_loss_backward_devcam if self.on_device else _lib.lib().adgs_flow_loss_backward


class _FlowLoss(torch.autograd.Function):
    @staticmethod
    def forward(ctx, img_flow, img_opacity, flow, flow_vis, K, R, T, dist):
        if not img_flow.is_cuda:
            raise RuntimeError("get_flow_loss: tensors must be on a HIP device; there is no CPU path")
        f = img_flow.contiguous().float()
        fl, vis = flow.contiguous().float(), flow_vis.contiguous().float()
        op = None if img_opacity is None else img_opacity.contiguous().float()
        H, W = fl.shape[1], fl.shape[2]
        if f.shape != (3, H, W) or fl.shape[0] != 2 or vis.shape != (H, W) or (op is not None and op.numel() != H * W):
            raise ValueError("get_flow_loss: expected img_flow [3,H,W], flow [2,H,W], flow_vis [H,W], img_opacity [H,W]")
        cam = _FlowCam(K, R, T, f.device)
        work, ctx.token = _work(f.device, AUX_WORK_DOUBLES)
        out = torch.empty(1, dtype=torch.float32, device=f.device) if H * W else torch.zeros(1, dtype=torch.float32, device=f.device)
        with torch.cuda.device(f.device):
            _lib.check(cam.forward_fn()(H, W, f.data_ptr(), fl.data_ptr(), vis.data_ptr(), op.data_ptr() if op is not None else None,
                                        cam.args[0], cam.args[1], cam.args[2], float(dist), work.data_ptr(), out.data_ptr(), _stream(f.device)),
                       "adgs_flow_loss_forward")
        ctx.token.done()
        ctx.save_for_backward(f, fl, vis, work, *([op] if op is not None else []))
        ctx.cam, ctx.dist, ctx.op_shape = cam, float(dist), None if img_opacity is None else img_opacity.shape
        return out[0]

    @staticmethod
    def backward(ctx, g_loss):
        f, fl, vis, work, *rest = ctx.saved_tensors
        op = rest[0] if rest else None
        H, W = fl.shape[1], fl.shape[2]
        g_f = torch.empty_like(f)
        g_op = torch.empty(H, W, dtype=torch.float32, device=f.device) if op is not None else None
        gl = g_loss.reshape(1).float().contiguous()
        with torch.cuda.device(f.device):
            _lib.check(ctx.cam.backward_fn()(H, W, f.data_ptr(), fl.data_ptr(), vis.data_ptr(), op.data_ptr() if op is not None else None,
                                             ctx.cam.args[0], ctx.cam.args[1], ctx.cam.args[2], ctx.dist, work.data_ptr(), gl.data_ptr(), g_f.data_ptr(),
                                             g_op.data_ptr() if g_op is not None else None, _stream(f.device)), "adgs_flow_loss_backward")
        return g_f, (g_op.reshape(ctx.op_shape) if g_op is not None else None), None, None, None, None, None, None


def get_flow_loss(img_flow, flow_pkg, img_opacity=None, dist=1e-3):
    """utils/loss_utils.py:86-106: mean over the pixels with a valid flow target of the normalised L1 distance between the
    re-projected rendered flow point and the target, weighted by the accumulated opacity; flow_pkg = (_, K, R, T, flow, flow_vis)
    as in the reference (train.py:68-71).  One reduction pass + one elementwise backward; the reference's nonzero() (a host
    synchronisation) and gathers are gone.  Always returns a tensor: 0 with zero gradients where the reference returns 0.0."""
    _, K, R, T, flow, flow_vis = flow_pkg
    return _FlowLoss.apply(img_flow, img_opacity, flow.detach(), flow_vis.detach(), K, R, T, dist)


class _BceClip(torch.autograd.Function):
    @staticmethod
    def forward(ctx, pred, target, lo, hi, invert, positive_target):
        if not pred.is_cuda:
            raise RuntimeError("bce_clip_loss: tensors must be on a HIP device; there is no CPU path")
        p, t = pred.contiguous().float(), target.contiguous().float()
        if p.numel() != t.numel():
            raise ValueError("bce_clip_loss: prediction and target must have the same number of elements")
        work, tok = _work(p.device, AUX_WORK_DOUBLES)
        out = torch.empty(1, dtype=torch.float32, device=p.device) if p.numel() else torch.zeros(1, dtype=torch.float32, device=p.device)
        with torch.cuda.device(p.device):
            _lib.check(_lib.lib().adgs_bce_clip_forward(p.numel(), p.data_ptr(), t.data_ptr(), float(lo), float(hi), int(bool(invert)),
                                                        int(bool(positive_target)), work.data_ptr(), out.data_ptr(), _stream(p.device)), "adgs_bce_clip_forward")
        tok.done()
        ctx.save_for_backward(p, t)
        ctx.args, ctx.shape = (float(lo), float(hi), int(bool(invert)), int(bool(positive_target))), pred.shape
        return out[0]

    @staticmethod
    def backward(ctx, g_loss):
        p, t = ctx.saved_tensors
        out = torch.empty_like(p)
        gl = g_loss.reshape(1).float().contiguous()
        lo, hi, inv, pos = ctx.args
        with torch.cuda.device(p.device):
            _lib.check(_lib.lib().adgs_bce_clip_backward(p.numel(), p.data_ptr(), t.data_ptr(), lo, hi, inv, pos, gl.data_ptr(), out.data_ptr(),
                                                         _stream(p.device)), "adgs_bce_clip_backward")
        return out.reshape(ctx.shape), None, None, None, None, None


def bce_clip_loss(pred, target, lo=1e-3, hi=1.0 - 1e-3, invert=False, positive_target=False):
    """mean BCE(q, t) with q = clip(pred, lo, hi) (1 - clip(...) with `invert`) and t = target ((target > 0) with `positive_target`)."""
    return _BceClip.apply(pred, target.detach(), lo, hi, invert, positive_target)


def obj_loss(img_semantic, gt_semantic):
    """train.py:95-98: binary_cross_entropy(clip(img_semantic, 1e-3, 1 - 1e-3)[0], (gt_semantic > 0).float())."""
    return bce_clip_loss(img_semantic[0] if img_semantic.dim() == 3 else img_semantic, gt_semantic, positive_target=True)


def sky_loss(img_opacity, gt_sky):
    """train.py:100-103: binary_cross_entropy(1 - clip(img_opacity, 1e-3, 1 - 1e-3), gt_sky)."""
    return bce_clip_loss(img_opacity, gt_sky, invert=True)


# ---------------------------------------------------------------- neighbourhood regularisers (train.py:104-113)
def _validate_near_idx(idx, N):
    """`param[obj_near_idx]` raises IndexError in the reference when an index lies outside [-N, N) (a stale obj_near_idx after a prune).
    A kernel cannot raise, so every index tensor is checked ONCE per (tensor object, in-place version, N): one min/max read-back when
    set_obj_near_idx / densify_and_prune installs a new tensor (every 10 iterations at most), none afterwards.  The record lives ON the
    tensor object (an attribute: it dies with the object and cannot be inherited by another tensor at the same address).  Skipped under
    stream capture (no read-back possible): there the kernels' own guards apply (NaN loss, no gradient for the affected groups)."""
    if idx.numel() == 0 or torch.cuda.is_current_stream_capturing():
        return
    if getattr(idx, "_adgs_validated", None) == (idx._version, N):
        return
    lo, hi = (int(v) for v in torch.aminmax(idx))
    if lo < -N or hi >= N:
        raise IndexError("obj_near_idx: index %d is out of bounds for dimension 0 with size %d (a stale neighbour index after densify / prune? "
                         "call set_obj_near_idx())" % (lo if lo < -N else hi, N))
    idx._adgs_validated = (idx._version, N)


class _GroupVar(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, idx, inner):
        if not x.is_cuda or not idx.is_cuda:
            raise RuntimeError("group variance loss: tensors must be on a HIP device; there is no CPU path")
        if idx.dim() != 2 or idx.dtype != torch.int64:
            raise ValueError("obj_near_idx must be an int64 [G, K] tensor")
        _validate_near_idx(idx, x.shape[0])
        xs, ix = x.contiguous().float(), idx.contiguous()
        N = xs.shape[0]
        D = xs.numel() // max(N, 1)
        G, K = ix.shape
        work, tok = _work(xs.device, AUX_WORK_DOUBLES)
        out = torch.empty(1, dtype=torch.float32, device=xs.device) if (G and D) else torch.zeros(1, dtype=torch.float32, device=xs.device)
        if G and D:
            with torch.cuda.device(xs.device):
                _lib.check(_lib.lib().adgs_group_var_forward(N, G, K, D, int(inner), xs.data_ptr(), ix.data_ptr(), work.data_ptr(), out.data_ptr(),
                                                             _stream(xs.device)), "adgs_group_var_forward")
        tok.done()
        ctx.save_for_backward(xs, ix)
        ctx.dims, ctx.shape = (N, G, K, D, int(inner)), x.shape
        return out[0]

    @staticmethod
    def backward(ctx, g_loss):
        xs, ix = ctx.saved_tensors
        N, G, K, D, inner = ctx.dims
        out = torch.zeros_like(xs)
        gl = g_loss.reshape(1).float().contiguous()
        if G and D:
            with torch.cuda.device(xs.device):
                _lib.check(_lib.lib().adgs_group_var_backward(N, G, K, D, inner, xs.data_ptr(), ix.data_ptr(), gl.data_ptr(), out.data_ptr(),
                                                              _stream(xs.device)), "adgs_group_var_backward")
        return out.reshape(ctx.shape), None, None


def reg_loss(xyz_deform_param, obj_near_idx):
    """train.py:104-106: mean(sum(var(xyz_deform_param[obj_near_idx], dim=1), dim=-1)); xyz_deform_param [N,3,C], obj_near_idx [G,K]."""
    return _GroupVar.apply(xyz_deform_param, obj_near_idx, xyz_deform_param.shape[-1])


def reg_sigma_loss(gs_time_sigma, obj_near_idx):
    """train.py:111-113: mean(sum(var(gs_time_sigma[obj_near_idx], dim=1), dim=-1)); gs_time_sigma [N,2]."""
    return _GroupVar.apply(gs_time_sigma, obj_near_idx, gs_time_sigma.shape[-1])


class _SigmaLoss(torch.autograd.Function):
    @staticmethod
    def forward(ctx, log_sigma, frame_gap):
        if not log_sigma.is_cuda:
            raise RuntimeError("sigma_loss: tensors must be on a HIP device; there is no CPU path")
        if log_sigma.dim() != 2 or log_sigma.shape[1] != 2:
            raise ValueError("gs_time_sigma must be [N, 2]")
        ls = log_sigma.contiguous().float()
        work, tok = _work(ls.device, AUX_WORK_DOUBLES)
        out = torch.empty(1, dtype=torch.float32, device=ls.device) if ls.shape[0] else torch.zeros(1, dtype=torch.float32, device=ls.device)
        if ls.shape[0]:
            with torch.cuda.device(ls.device):
                _lib.check(_lib.lib().adgs_sigma_loss_forward(ls.shape[0], ls.data_ptr(), float(frame_gap), work.data_ptr(), out.data_ptr(),
                                                              _stream(ls.device)), "adgs_sigma_loss_forward")
        tok.done()
        ctx.save_for_backward(ls)
        ctx.gap = float(frame_gap)
        return out[0]

    @staticmethod
    def backward(ctx, g_loss):
        (ls,) = ctx.saved_tensors
        out = torch.empty_like(ls)
        gl = g_loss.reshape(1).float().contiguous()
        if ls.shape[0]:
            with torch.cuda.device(ls.device):
                _lib.check(_lib.lib().adgs_sigma_loss_backward(ls.shape[0], ls.data_ptr(), ctx.gap, gl.data_ptr(), out.data_ptr(), _stream(ls.device)),
                           "adgs_sigma_loss_backward")
        return out, None


def sigma_loss(gs_time_sigma, frame_gap):
    """train.py:108-110: mean(|frame_gap / mean(exp(gs_time_sigma), dim=-1)|)."""
    return _SigmaLoss.apply(gs_time_sigma, frame_gap)


_WEIGHTS = {}


class _WeightedSum(torch.autograd.Function):
    @staticmethod
    def forward(ctx, weights, *terms):
        ctx.save_for_backward(weights)
        ctx.n = len(terms)
        return torch.dot(torch.stack([t.reshape(()) for t in terms]).to(weights.dtype), weights)

    @staticmethod
    def backward(ctx, g):
        (weights,) = ctx.saved_tensors
        gw = g * weights                          # one kernel; the terms' gradients are views of it
        return (None,) + tuple(gw.unbind(0))


def weighted_total(terms):
    """sum_i w_i * L_i of scalar loss terms, [(w_i, L_i), ...] -- what train.py:112-115 writes as a chain of python scalar products
    and sums (two kernels per term forward, two more backward: ~40 launches of 2 - 4 us each per iteration) as one stack, one dot
    product and one scaling in the backward.  Same value up to the order of the float32 additions."""
    terms = [(float(w), t) for w, t in terms if t is not None and not (isinstance(t, (int, float)) and t == 0)]
    if not terms:
        return 0.0
    ref = next(t for _, t in terms if torch.is_tensor(t))
    key = (tuple(w for w, _ in terms), ref.device)
    w = _WEIGHTS.get(key)
    if w is None:                                  # the lambdas are constants of a run: one upload
        if len(_WEIGHTS) > 64:
            _WEIGHTS.clear()
        w = _WEIGHTS[key] = torch.tensor(key[0], dtype=torch.float32, device=ref.device)
    ts = [t if torch.is_tensor(t) else torch.tensor(float(t), dtype=torch.float32, device=ref.device) for _, t in terms]
    return _WeightedSum.apply(w, *ts)


# ---------------------------------------------------------------- the image terms of train.py:78-99 as ONE autograd node
class _ImageLosses(torch.autograd.Function):
    """L1, SSIM, depth loss, flow loss, object BCE, sky BCE -- the same kernels, the same work buffers as the six functions above, behind
    one autograd node: six Python-level Function calls forward and six backward (each ~30 us of host time around a 5 - 90 us kernel)
    become two, so the short kernels of this section no longer wait for the host between them (tools/iteration_gaps.py: ~100 us of
    idle GPU per iteration in front of the rasterizer's backward).  Output: a [6] tensor (Ll1, ssim, depth, flow, obj, sky)."""

    @staticmethod
    def forward(ctx, image, depth, img_flow, img_opacity, img_semantic, gt_image, gt_depth, flow, flow_vis, cam, dist, gt_semantic, gt_sky):
        dev = image.device
        if not image.is_cuda:
            raise RuntimeError("image_losses: tensors must be on a HIP device; there is no CPU path")
        L = _lib.lib()
        st = _stream(dev)
        f32 = lambda t: t.contiguous().float()
        img, ref = f32(image), f32(gt_image)
        H, W = img.shape[-2:]
        if img.shape != ref.shape or img.dim() != 3:
            raise ValueError("image_losses: image and gt_image must be [C, H, W]")
        npix = H * W
        dep, gdep = f32(depth).reshape(H, W), f32(gt_depth).reshape(H, W)
        fl_img, op = f32(img_flow), f32(img_opacity).reshape(H, W)
        fl, vis = f32(flow), f32(flow_vis)
        sem = f32(img_semantic[0] if img_semantic.dim() == 3 else img_semantic).reshape(H, W)
        gsem, gsky = f32(gt_semantic).reshape(H, W), f32(gt_sky).reshape(H, W)
        if fl_img.shape != (3, H, W) or fl.shape != (2, H, W) or vis.shape != (H, W):
            raise ValueError("image_losses: expected img_flow [3,H,W], flow [2,H,W], flow_vis [H,W]")
        terms = torch.empty(6, dtype=torch.float32, device=dev)
        maps = [torch.empty_like(img) for _ in range(3)]
        sums, tok_sums = _work(dev, 2 * SLOTS)
        w_depth, tok_depth = _work(dev, DEPTH_WORK_DOUBLES)
        w_flow, tok_flow = _work(dev, AUX_WORK_DOUBLES)
        w_obj, tok_obj = _work(dev, AUX_WORK_DOUBLES)
        w_sky, tok_sky = _work(dev, AUX_WORK_DOUBLES)
        p0 = terms.data_ptr()
        with torch.cuda.device(dev):
            _lib.check(L.adgs_l1_ssim_forward(img.shape[0], H, W, img.data_ptr(), ref.data_ptr(), sums.data_ptr(), *[m.data_ptr() for m in maps], st), "adgs_l1_ssim_forward")
            _lib.check(L.adgs_l1_ssim_means(sums.data_ptr(), img.numel(), p0, st), "adgs_l1_ssim_means")
            _lib.check(L.adgs_depth_loss_forward(npix, dep.data_ptr(), gdep.data_ptr(), None, w_depth.data_ptr(), p0 + 8, st), "adgs_depth_loss_forward")
            _lib.check(cam.forward_fn()(H, W, fl_img.data_ptr(), fl.data_ptr(), vis.data_ptr(), op.data_ptr(), cam.args[0], cam.args[1], cam.args[2], float(dist),
                                        w_flow.data_ptr(), p0 + 12, st), "adgs_flow_loss_forward")
            _lib.check(L.adgs_bce_clip_forward(npix, sem.data_ptr(), gsem.data_ptr(), 1e-3, 1.0 - 1e-3, 0, 1, w_obj.data_ptr(), p0 + 16, st), "adgs_bce_clip_forward")
            _lib.check(L.adgs_bce_clip_forward(npix, op.data_ptr(), gsky.data_ptr(), 1e-3, 1.0 - 1e-3, 1, 0, w_sky.data_ptr(), p0 + 20, st), "adgs_bce_clip_forward")
        for tok in (tok_sums, tok_depth, tok_flow, tok_obj, tok_sky):
            tok.done()
        ctx.save_for_backward(img, ref, *maps, dep, gdep, w_depth, fl_img, fl, vis, op, w_flow, sem, gsem, gsky)
        ctx.tokens, ctx.cam, ctx.dist = (tok_depth, tok_flow), cam, float(dist)
        ctx.shapes = (image.shape, depth.shape, img_flow.shape, img_opacity.shape, img_semantic.shape)
        return terms

    @staticmethod
    def backward(ctx, g):
        (img, ref, d_mu1, d_e11, d_e12, dep, gdep, w_depth, fl_img, fl, vis, op, w_flow, sem, gsem, gsky) = ctx.saved_tensors
        dev = img.device
        L = _lib.lib()
        st = _stream(dev)
        H, W = img.shape[-2:]
        npix = H * W
        g = g.contiguous().float()
        p0 = g.data_ptr()
        g_img, g_dep, g_fl = torch.empty_like(img), torch.empty_like(dep), torch.empty_like(fl_img)
        g_op, g_op2, g_sem = torch.empty_like(op), torch.empty_like(op), torch.empty_like(sem)
        with torch.cuda.device(dev):
            _lib.check(L.adgs_l1_ssim_backward(img.shape[0], H, W, img.data_ptr(), ref.data_ptr(), d_mu1.data_ptr(), d_e11.data_ptr(), d_e12.data_ptr(),
                                               p0, p0 + 4, g_img.data_ptr(), st), "adgs_l1_ssim_backward")
            _lib.check(L.adgs_depth_loss_backward(npix, dep.data_ptr(), gdep.data_ptr(), None, w_depth.data_ptr(), p0 + 8, g_dep.data_ptr(), st), "adgs_depth_loss_backward")
            _lib.check(ctx.cam.backward_fn()(H, W, fl_img.data_ptr(), fl.data_ptr(), vis.data_ptr(), op.data_ptr(), ctx.cam.args[0], ctx.cam.args[1], ctx.cam.args[2], ctx.dist,
                                             w_flow.data_ptr(), p0 + 12, g_fl.data_ptr(), g_op.data_ptr(), st), "adgs_flow_loss_backward")
            _lib.check(L.adgs_bce_clip_backward(npix, sem.data_ptr(), gsem.data_ptr(), 1e-3, 1.0 - 1e-3, 0, 1, p0 + 16, g_sem.data_ptr(), st), "adgs_bce_clip_backward")
            _lib.check(L.adgs_bce_clip_backward(npix, op.data_ptr(), gsky.data_ptr(), 1e-3, 1.0 - 1e-3, 1, 0, p0 + 20, g_op2.data_ptr(), st), "adgs_bce_clip_backward")
        g_op.add_(g_op2)                               # img_opacity feeds the flow loss and the sky loss
        s_img, s_dep, s_fl, s_op, s_sem = ctx.shapes
        if len(s_sem) == 3:                            # [D_S, H, W]: only channel 0 enters the object loss (train.py:95-98)
            full = torch.zeros(s_sem, dtype=torch.float32, device=dev) if s_sem[0] > 1 else None
            g_sem_out = g_sem.reshape(1, H, W) if full is None else full
            if full is not None:
                full[0].copy_(g_sem)
        else:
            g_sem_out = g_sem.reshape(s_sem)
        return (g_img.reshape(s_img), g_dep.reshape(s_dep), g_fl.reshape(s_fl), g_op.reshape(s_op), g_sem_out) + (None,) * 8


def image_losses(image, gt_image, depth, gt_depth, img_flow, flow_pkg, img_opacity, img_semantic, gt_semantic, gt_sky, dist=1e-3):
    """The six image terms of a training iteration in one autograd node: returns (Ll1, ssim, depth_loss, flow_loss, obj_loss, sky_loss) --
    bit for bit what l1_ssim, get_depth_loss (no mask), get_flow_loss, obj_loss and sky_loss return for the same arguments
    (utils/loss_utils.py:20-106, train.py:78-99; flow_pkg = (_, K, R, T, flow, flow_vis) as in train.py:68-71)."""
    _, K, R, T, flow, flow_vis = flow_pkg
    cam = _FlowCam(K, R, T, image.device)
    terms = _ImageLosses.apply(image, depth, img_flow, img_opacity, img_semantic, gt_image.detach(), gt_depth.detach(), flow.detach(), flow_vis.detach(),
                               cam, float(dist), gt_semantic.detach(), gt_sky.detach())
    return tuple(terms.unbind(0))
