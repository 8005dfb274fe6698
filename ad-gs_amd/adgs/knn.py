"""K-nearest-neighbour index on the HIP path: drop-in for `pytorch3d.ops.knn_points` as the reference uses it
(scene/gaussian_model.py:23,825-833) and for GaussianModel.set_obj_near_idx.  HIP kernel only; no CPU path."""
import collections
import ctypes

import torch

from . import _lib

_KNN = collections.namedtuple("KNN", ["dists", "idx", "knn"])


def knn_points(p1, p2, K=1):
    """p1 [1,A,D], p2 [1,N,D] (D = 3 or 4) -> KNN(dists [1,A,K] squared distances ascending, idx [1,A,K] int64, knn None).
    The subset of pytorch3d.ops.knn_points the reference calls (batch of one, no lengths, no return_nn)."""
    if p1.dim() != 3 or p2.dim() != 3 or p1.shape[0] != 1 or p2.shape[0] != 1 or p1.shape[2] != p2.shape[2]:
        raise ValueError("knn_points: expected p1 [1,A,D] and p2 [1,N,D]")
    if not p1.is_cuda or not p2.is_cuda:
        raise RuntimeError("knn_points: tensors must be on a HIP device; there is no CPU path")
    A, D, N = p1.shape[1], p1.shape[2], p2.shape[1]
    a, p = p1[0].detach().contiguous().float(), p2[0].detach().contiguous().float()
    idx = torch.empty((A, K), dtype=torch.int64, device=p1.device)
    dists = torch.empty((A, K), dtype=torch.float32, device=p1.device)
    if A > 0:
        lib = _lib.lib()
        with torch.cuda.device(p1.device):
            ws = torch.empty(int(lib.adgs_knn_points_workspace_bytes(A, N, K)), dtype=torch.uint8, device=p1.device)
            _lib.check(lib.adgs_knn_points(A, a.data_ptr(), N, p.data_ptr(), D, K, idx.data_ptr(), dists.data_ptr(), ws.data_ptr(),
                                           _lib.stream_ptr(p1.device)), "adgs_knn_points")
    return _KNN(dists[None], idx[None], None)


def set_obj_near_idx(model, K=None):
    """GaussianModel.set_obj_near_idx (scene/gaussian_model.py:825-833): K nearest object Gaussians (in xyz, or xyz + time *
    scene_extent with the time mask) of N // K randomly drawn anchors; sets and returns model.obj_near_idx [N // K, K]."""
    if not getattr(model, "use_near_idx", False):
        return None
    K = model.near_num if K is None else K
    xyz = model._obj_xyz.detach()
    if getattr(model, "use_time_mask", False):
        xyz = torch.cat([xyz, model.gs_time * model.scene_extent], dim=-1)
    anchor = xyz[torch.randperm(xyz.shape[0], device=xyz.device)[:xyz.shape[0] // K]]
    model.obj_near_idx = knn_points(anchor[None], xyz[None], K=K).idx.squeeze(0)      # [G, K] also for a single anchor (the reference's .squeeze() would drop that axis too)
    return model.obj_near_idx
