"""The kNN-index oracle against an independent exact implementation (scipy.spatial.cKDTree, float64): same neighbour sets and
order wherever the K+1 nearest distances are separated by more than float32 rounding."""
import numpy as np
import pytest
from scipy.spatial import cKDTree

from oracle import knn_points_oracle as ko


@pytest.mark.parametrize("N,D,K", [(3000, 3, 8), (2500, 4, 8), (500, 3, 1), (900, 4, 16)])
def test_knn_points_oracle_vs_ckdtree(N, D, K):
    rng = np.random.default_rng(N + D + K)
    pts = rng.normal(size=(N, D)).astype(np.float32)
    anchors = pts[rng.permutation(N)[:N // K]]
    dist, idx = ko.knn_points(anchors, pts, K)
    dd, ii = cKDTree(pts.astype(np.float64)).query(anchors.astype(np.float64), k=K + 1)
    dd, ii = dd.reshape(len(anchors), -1), ii.reshape(len(anchors), -1)
    gaps = np.diff(dd ** 2, axis=1).min(axis=1) > 1e-5             # rows without a near-tie among the K+1 nearest
    assert gaps.mean() > 0.9
    assert np.array_equal(idx[gaps], ii[gaps][:, :K])
    np.testing.assert_allclose(dist, (dd ** 2)[:, :K], rtol=1e-5, atol=1e-6)
    assert np.array_equal(idx[:, 0], np.array([np.where((pts == a).all(1))[0][0] for a in anchors]))      # an anchor's nearest point is itself
