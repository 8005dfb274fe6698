"""NumPy oracle of the K-nearest-neighbour index (include/adgs_knn_points.h).

TEST INFRASTRUCTURE ONLY: only tests/ may import this module.

Restates pytorch3d.ops.knn_points as the reference calls it (scene/gaussian_model.py:825-833; pytorch3d is an un-vendored
dependency, environment.yaml, not installable here: PARITY UNPINNED at that boundary).  Published semantics: squared
Euclidean distances, the K smallest per query sorted ascending.  Distances are accumulated in float32 over the dimensions
in order, ties go to the lower index.  tests/test_oracle_knn_points.py cross-checks it against scipy.spatial.cKDTree."""
import numpy as np


def knn_points(anchors, points, K, block=512):
    a = np.ascontiguousarray(anchors, np.float32); p = np.ascontiguousarray(points, np.float32)
    A, D = a.shape
    idx = np.zeros((A, K), np.int64); dist = np.zeros((A, K), np.float32)
    for s in range(0, A, block):
        q = a[s:s + block]
        d = np.zeros((q.shape[0], p.shape[0]), np.float32)
        for c in range(D):
            df = q[:, c:c + 1] - p[None, :, c]
            d = d + df * df
        if K * 8 >= p.shape[0]:
            order = np.argsort(d, axis=1, kind="stable")[:, :K]
        else:
            # the K smallest per row without sorting the row: everything up to the K-th smallest VALUE (all of its ties included), in
            # index order, then a stable sort of those few -- the same lists as the full stable argsort, ~10x faster at 100 000 points
            kth = np.partition(d, K - 1, axis=1)[:, K - 1]
            order = np.empty((q.shape[0], K), np.int64)
            for r in range(q.shape[0]):
                cand = np.flatnonzero(d[r] <= kth[r])
                order[r] = cand[np.argsort(d[r, cand], kind="stable")[:K]]
        idx[s:s + block] = order
        dist[s:s + block] = np.take_along_axis(d, order, axis=1)
    return dist, idx
