"""GPU parity of the HIP K-nearest-neighbour index (adgs.knn.knn_points / set_obj_near_idx) vs the NumPy oracle: index
lists bit-exact (same float32 distance arithmetic, same tie rule), distances bit-exact."""
import os

import numpy as np
import pytest
import torch

from oracle import knn_points_oracle as ko

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("N,A,D,K", [(20000, 2500, 3, 8), (20000, 2500, 4, 8), (1000, 1000, 3, 1), (5000, 77, 4, 16), (3000, 300, 3, 32), (64, 8, 3, 8),
                                     (100_000, 2_500, 4, 8)])      # (the oracle's cost is A x N: 12 500 anchors took 79 s of the suite's 1200 s limit)
def test_knn_points_vs_oracle(N, A, D, K):
    from adgs.knn import knn_points
    rng = np.random.default_rng(N + A + D + K)
    pts = rng.normal(size=(N, D)).astype(np.float32)
    pts[N // 2:N // 2 + 20] = pts[:20]                      # exact duplicates: ties must resolve to the lower index
    anchors = pts[rng.permutation(N)[:A]]
    res = knn_points(torch.tensor(anchors, device="cuda")[None], torch.tensor(pts, device="cuda")[None], K=K)
    torch.cuda.synchronize()
    assert res.idx.shape == (1, A, K) and res.idx.dtype == torch.int64 and res.dists.shape == (1, A, K)
    dist, idx = ko.knn_points(anchors, pts, K)
    assert np.array_equal(res.idx[0].cpu().numpy(), idx)
    assert np.array_equal(res.dists[0].cpu().numpy(), dist)


@pytest.mark.parametrize("seed", range(int(os.environ.get("ADGS_TEST_SEED_BASE", "0")), int(os.environ.get("ADGS_TEST_SEED_BASE", "0")) + int(os.environ.get("ADGS_TEST_KNN_SEEDS", "12"))))
def test_knn_points_fuzz(seed):
    """Random sizes (fewer points than a block, K = N, heavy duplication on a coarse lattice -> many exact distance ties)."""
    from adgs.knn import knn_points
    rng = np.random.default_rng(6000 + seed)
    N = int(rng.choice([1, 2, 33, 64, 257, 1000, 4097, 9000]))
    D = int(rng.choice([3, 4]))
    K = int(min(N, rng.choice([1, 2, 7, 8, 16, 32])))
    A = int(rng.integers(1, N + 1))
    pts = rng.normal(size=(N, D)).astype(np.float32)
    if rng.integers(2):
        pts = np.round(pts * 2) / 2                       # lattice: many points coincide, many distances tie exactly
    anchors = pts[rng.permutation(N)[:A]] if rng.integers(2) else rng.normal(size=(A, D)).astype(np.float32)
    res = knn_points(torch.tensor(anchors, device="cuda")[None], torch.tensor(pts, device="cuda")[None], K=K)
    dist, idx = ko.knn_points(anchors, pts, K)
    assert np.array_equal(res.idx[0].cpu().numpy(), idx), (N, A, D, K)
    assert np.array_equal(res.dists[0].cpu().numpy(), dist), (N, A, D, K)


def test_set_obj_near_idx_and_errors():
    from adgs.knn import knn_points, set_obj_near_idx

    class M:
        pass
    m = M()
    g = torch.Generator(device="cuda").manual_seed(0)
    m._obj_xyz = torch.randn(4000, 3, device="cuda", generator=g)
    m.gs_time = torch.rand(4000, 1, device="cuda", generator=g)
    m.scene_extent, m.use_time_mask, m.use_near_idx, m.near_num = 20.0, True, True, 8
    torch.manual_seed(3)
    idx = set_obj_near_idx(m)
    assert idx.shape == (500, 8) and idx.dtype == torch.int64 and int(idx.min()) >= 0 and int(idx.max()) < 4000
    torch.manual_seed(3)
    perm = torch.randperm(4000, device="cuda")[:500]
    assert torch.equal(idx[:, 0], perm)                      # every anchor's nearest neighbour is itself
    xyz4 = torch.cat([m._obj_xyz, m.gs_time * 20.0], -1).cpu().numpy()
    assert np.array_equal(idx.cpu().numpy(), ko.knn_points(xyz4[perm.cpu().numpy()], xyz4, 8)[1])
    m.use_near_idx = False
    assert set_obj_near_idx(m) is None
    with pytest.raises(RuntimeError):
        knn_points(torch.zeros(1, 4, 3), torch.zeros(1, 9, 3), K=2)
    with pytest.raises(RuntimeError):
        knn_points(torch.zeros(1, 4, 3, device="cuda"), torch.zeros(1, 9, 3, device="cuda"), K=33)


@pytest.mark.parametrize("N,A,D,K,lattice", [(300, 300, 3, 8, True), (5000, 700, 4, 8, True), (6000, 6000, 3, 32, False), (200_000, 25_000, 4, 8, False)])
def test_slab_search_equals_brute_force(monkeypatch, N, A, D, K, lattice):
    """The two exact paths (slab search along the axis of largest extent / tiled brute force) return identical index lists and
    distances -- also on a coarse lattice (exact distance ties, equal axis coordinates) and at the training size (200 k object
    Gaussians in clusters, time as the fourth coordinate)."""
    from adgs.knn import knn_points
    rng = np.random.default_rng(77 + N + K)
    if N >= 100_000:      # clustered like the object Gaussians of C3: 8 clusters of radius ~2 m, time * scene_extent on axis 3
        centres = rng.uniform(-20, 20, size=(8, 3)).astype(np.float32)
        pts = (centres[rng.integers(0, 8, N)] + rng.normal(scale=1.0, size=(N, 3))).astype(np.float32)
        pts = np.concatenate([pts, (rng.random((N, 1)) * 20.0).astype(np.float32)], 1) if D == 4 else pts
    else:
        pts = rng.normal(size=(N, D)).astype(np.float32)
    if lattice:
        pts = np.round(pts * 2) / 2
    anchors = pts[rng.permutation(N)[:A]]
    a, p = torch.tensor(anchors, device="cuda")[None], torch.tensor(pts, device="cuda")[None]
    out = {}
    for mode in ("slab", "brute"):
        monkeypatch.setenv("ADGS_KNN_POINTS", mode)
        res = knn_points(a, p, K=K)
        torch.cuda.synchronize()
        out[mode] = (res.idx[0].cpu().numpy(), res.dists[0].cpu().numpy())
    assert np.array_equal(out["slab"][0], out["brute"][0])
    assert np.array_equal(out["slab"][1], out["brute"][1])
    if N <= 6000:
        dist, idx = ko.knn_points(anchors, pts, K)
        assert np.array_equal(out["slab"][0], idx) and np.array_equal(out["slab"][1], dist)
