"""Photometric loss on the HIP path: drop-ins for
utils/loss_utils.py `l1_loss` (:20-21) and `ssim` (:37-68) as used at train.py:79-80.

`l1_ssim(image, gt)` evaluates both in ONE forward kernel and back-propagates both in ONE backward
kernel (adgs_l1_ssim_forward / _backward, include/adgs_loss.h); `l1_loss` / `ssim` keep the reference's
names and signatures on top of it.  Images are [..., C, H, W] fp32 on a HIP device, `gt` is a constant.
There is no CPU fallback.
"""
import ctypes

import torch

from . import _lib

SLOTS = 256        # ADGS_LOSS_SLOTS


def _stream(dev):
    return ctypes.c_void_p(torch.cuda.current_stream(dev).cuda_stream)


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
        sums = torch.zeros(SLOTS, 2, dtype=torch.float64, device=img.device)      # spread atomics (include/adgs_loss.h)
        maps = [torch.empty_like(img) for _ in range(3)] if need else [None] * 3
        if n:
            with torch.cuda.device(img.device):
                _lib.check(_lib.lib().adgs_l1_ssim_forward(planes, H, W, img.data_ptr(), ref.data_ptr(), sums.data_ptr(),
                                                           *[m.data_ptr() if m is not None else None for m in maps], _stream(img.device)),
                           "adgs_l1_ssim_forward")
        means = (sums.sum(0) / max(n, 1)).float()
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
        work = torch.zeros(DEPTH_WORK_DOUBLES, dtype=torch.float64, device=p.device)
        out = torch.zeros(1, dtype=torch.float32, device=p.device)
        with torch.cuda.device(p.device):
            _lib.check(_lib.lib().adgs_depth_loss_forward(p.numel(), p.data_ptr(), g.data_ptr(), m.data_ptr() if m is not None else None,
                                                          work.data_ptr(), out.data_ptr(), _stream(p.device)), "adgs_depth_loss_forward")
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
