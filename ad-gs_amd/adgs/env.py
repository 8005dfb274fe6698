"""Environment-map background on the HIP path: drop-in for the render-time
surface of scene/env.py `EnvironmentMap` -- `get_image_background(cam)` (env.py:43-66), `training_setup`,
`save_weights` / `load_weights` -- over adgs_envmap_forward / _backward (include/adgs_envmap.h).

The reference keeps a [H, W, 3] ray tensor per camera (29 MB at 1920x1280) and runs a dozen torch kernels per
call; here the rays are recomputed inside the one kernel.  `grid_map` keeps the reference's [1, C, R, R] layout
and is the only trainable input.  There is no CPU fallback.
"""
import ctypes
import math

import torch
from torch import nn

from . import _lib
from .optim import FusedAdam, ADAM_TILE


def fov2focal(fov, pixels):
    """utils/graphics_utils.py: pixels / (2 tan(fov / 2))."""
    return pixels / (2 * math.tan(fov / 2))


class _EnvBackground(torch.autograd.Function):
    @staticmethod
    def forward(ctx, grid_map, H, W, focal, R9, marked=None):
        if not grid_map.is_cuda:
            raise RuntimeError("EnvironmentMap: grid_map must be on a HIP device; there is no CPU path")
        gm = grid_map.contiguous().float()
        C, Hm, Wm = gm.shape[-3:]
        out = torch.empty(C, H, W, dtype=torch.float32, device=gm.device)
        Rarr = (ctypes.c_float * 9)(*R9)
        with torch.cuda.device(gm.device):
            _lib.check(_lib.lib().adgs_envmap_forward(C, Hm, Wm, gm.data_ptr(), H, W, float(focal), Rarr, out.data_ptr(),
                                                      _lib.stream_ptr(gm.device)), "adgs_envmap_forward")
        ctx.save_for_backward(out)
        ctx.meta = (tuple(grid_map.shape), C, Hm, Wm, H, W, float(focal), Rarr)
        ctx.marked = marked
        return out

    @staticmethod
    def backward(ctx, g):
        (out,) = ctx.saved_tensors
        shape, C, Hm, Wm, H, W, focal, Rarr = ctx.meta
        mk = ctx.marked
        gg = mk.take() if mk is not None else None        # a buffer the optimizer left all zero (adgs.optim.MarkedGradient)
        if gg is None or tuple(gg.shape) != shape or gg.device != out.device:
            gg = torch.zeros(shape, dtype=torch.float32, device=out.device)       # dense, like grid_sample's backward
        g = g.contiguous().float()
        with torch.cuda.device(out.device):
            _lib.check(_lib.lib().adgs_envmap_backward_marked(C, Hm, Wm, H, W, focal, Rarr, out.data_ptr(), g.data_ptr(), gg.data_ptr(),
                                                              mk.marks.data_ptr() if mk is not None else None, ADAM_TILE,
                                                              _lib.stream_ptr(out.device)), "adgs_envmap_backward")
        if mk is not None:
            mk.issued(gg)
        return gg, None, None, None, None, None


def image_background(grid_map, H, W, focal, R):
    """sigmoid(bilinear(grid_map)) along the pixel rays of a pinhole camera: [C, H, W].  R = world_view_transform[:3,:3]
    (anything convertible to nine floats, row-major)."""
    R9 = [float(v) for v in (R.detach().cpu().reshape(-1).tolist() if torch.is_tensor(R) else [x for row in R for x in row])]
    return _EnvBackground.apply(grid_map, int(H), int(W), float(focal), R9, None)


class EnvironmentMap:
    def __init__(self, resolution, num_channel=3, use_cache=True, device="cuda", sparse_grad=False):
        """sparse_grad (extension; the default False is the reference's behaviour step for step): the backward marks the 256-element
        tiles of the map it writes into in the optimizer's tile map and reuses one gradient buffer the optimizer keeps zero outside
        them, so a step costs neither a fill pass nor a scan of the dense 3 x 8192^2 gradient (0.8 GB each).  Valid while the map's
        gradient comes from get_image_background only and is consumed by `self.optimizer.step(zero_grad=True)` (or step() followed by
        zero_grad(), which only saves the scan); any gradient that reaches the optimizer as another tensor is handled densely."""
        self.sparse_grad = bool(sparse_grad)
        self.resolution = resolution
        grid_map = (torch.rand((1, num_channel, resolution, resolution), dtype=torch.float32, device=device) * 2.0 - 1.0) * 1e-4
        self.grid_map = nn.Parameter(grid_map.requires_grad_(True))
        self.optimizer = None
        self.use_cache = use_cache
        self._cam_cache = {}                  # cam_id -> (focal, nine host floats): the camera is constant, read it back once

    def _camera(self, cam, use_cache):
        key = getattr(cam, "cam_id", None)
        if use_cache and key in self._cam_cache:
            return self._cam_cache[key]
        focal = fov2focal(cam.FoVx, cam.image_width)
        R9 = [float(v) for v in cam.world_view_transform[:3, :3].detach().cpu().reshape(-1).tolist()]
        if use_cache and key is not None:
            self._cam_cache[key] = (focal, R9)
        return focal, R9

    def get_image_background(self, cam, use_cache=True, return_grid=False):
        focal, R9 = self._camera(cam, use_cache and self.use_cache)
        marked = self.optimizer.marked_gradient(self.grid_map) if (self.sparse_grad and self.optimizer is not None and torch.is_grad_enabled()) else None
        bg = _EnvBackground.apply(self.grid_map, int(cam.image_height), int(cam.image_width), float(focal), R9, marked)
        if return_grid:
            dev = self.grid_map.device
            grid = torch.stack(torch.meshgrid(torch.arange(0, cam.image_width, dtype=torch.float32, device=dev),
                                              torch.arange(0, cam.image_height, dtype=torch.float32, device=dev), indexing="xy"), dim=-1)
            return bg, grid
        return bg

    def training_setup(self, training_args):
        """scene/env.py:78-83 with the fused Adam."""
        # the cameras of a sequence see a small part of the sphere: texels that never receive a gradient keep zero moments, and the
        # Adam step leaves them exactly where they are -- skip_dormant_tiles does that at 4 instead of 28 bytes per texel
        self.optimizer = FusedAdam([{"params": [self.grid_map], "lr": training_args.env_lr, "name": "env"}], lr=0.0, eps=1e-15, skip_dormant_tiles=True)

    def save_weights(self, weights_path):
        torch.save(self.grid_map, weights_path)

    def load_weights(self, weights_path):
        grid_map = torch.load(weights_path, map_location=self.grid_map.device)
        self.grid_map = nn.Parameter(grid_map.requires_grad_(True))
