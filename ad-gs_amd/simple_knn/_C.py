"""`simple_knn._C` drop-in (reference: submodules/simple-knn/ext.cpp:15-17, spatial.cu:15-26).

distCUDA2(points[P,3] float32 on a HIP device) -> float32[P]: mean of the squared
distances to the three nearest OTHER points.  HIP kernels only; no CPU path.
"""
import ctypes

import torch

from adgs import _lib


def distCUDA2(points):
    if not points.is_cuda:
        raise RuntimeError("distCUDA2: points must be on a HIP device; there is no CPU path")
    lib = _lib.lib()
    P = points.size(0)
    pts = points.contiguous().float()
    means = torch.zeros((P,), dtype=torch.float32, device=points.device)
    if P == 0:
        return means
    with torch.cuda.device(points.device):
        ws = torch.empty((int(lib.adgs_knn_workspace_bytes(P)),), dtype=torch.uint8, device=points.device)
        stream = _lib.stream_ptr(points.device)
        _lib.check(lib.adgs_knn_dist2(P, pts.data_ptr(), means.data_ptr(), ws.data_ptr(), stream), "adgs_knn_dist2")
    return means
