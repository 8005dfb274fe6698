"""GPU tests of the hand-written scan / radix sort (bit-exact vs numpy) and the HIP 3-NN
kernel (bit-exact vs the CPU oracle, which itself is checked against brute force)."""
import ctypes
import os

import numpy as np
import pytest
import torch

from oracle import oracle

pytestmark = pytest.mark.gpu


def _stream():
    return ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)


@pytest.mark.parametrize("n", [1, 2, 255, 2048, 2049, 4096 * 3 + 5, 1_000_003, 5_000_017])
def test_exclusive_scan(n):
    from adgs import _lib
    lib = _lib.lib()
    g = torch.Generator().manual_seed(n)
    x = torch.randint(0, 50, (n,), generator=g, dtype=torch.int32)
    xd = x.cuda()
    out = torch.empty_like(xd)
    tmp = torch.empty(int(lib.adgs_test_scan_temp_bytes(n)), dtype=torch.uint8, device="cuda")
    assert lib.adgs_test_exclusive_scan_u32(xd.data_ptr(), out.data_ptr(), n, tmp.data_ptr(), _stream()) == 0
    ref = np.concatenate([[0], np.cumsum(x.numpy().astype(np.int64))[:-1]]).astype(np.uint32)
    np.testing.assert_array_equal(out.cpu().numpy().view(np.uint32), ref)
    # in place
    assert lib.adgs_test_exclusive_scan_u32(xd.data_ptr(), xd.data_ptr(), n, tmp.data_ptr(), _stream()) == 0
    np.testing.assert_array_equal(xd.cpu().numpy().view(np.uint32), ref)


@pytest.mark.parametrize("n,end_bit", [(1, 46), (300, 41), (2048, 43), (100_001, 46), (3_000_017, 46), (1_000_000, 64)])
def test_radix_sort_u64_is_stable(n, end_bit):
    from adgs import _lib
    lib = _lib.lib()
    rng = np.random.RandomState(n % 1000)
    tile_bits = max(end_bit - 32, 1)
    # few distinct depths so that ties are common: stability is observable through the values
    keys = (rng.randint(0, 1 << min(tile_bits, 14), size=n).astype(np.uint64) << np.uint64(32)) | rng.randint(0, 64, size=n).astype(np.uint64) * np.uint64(0x01010101)
    if end_bit == 64:
        keys = rng.randint(0, 2 ** 63, size=n, dtype=np.int64).astype(np.uint64) * np.uint64(2) + rng.randint(0, 2, size=n).astype(np.uint64)
    vals = np.arange(n, dtype=np.uint32)
    kd = torch.from_numpy(keys.view(np.int64)).cuda(); vd = torch.from_numpy(vals.view(np.int32)).cuda()
    ko = torch.empty_like(kd); vo = torch.empty_like(vd)
    tmp = torch.empty(int(lib.adgs_test_sort_temp_bytes(n)), dtype=torch.uint8, device="cuda")
    assert lib.adgs_test_sort_pairs_u64(kd.data_ptr(), ko.data_ptr(), vd.data_ptr(), vo.data_ptr(), n, end_bit, tmp.data_ptr(), _stream()) == 0
    mask = np.uint64((1 << end_bit) - 1) if end_bit < 64 else np.uint64(0xFFFFFFFFFFFFFFFF)
    order = np.argsort(keys & mask, kind="stable")
    np.testing.assert_array_equal(ko.cpu().numpy().view(np.uint64), keys[order])
    np.testing.assert_array_equal(vo.cpu().numpy().view(np.uint32), vals[order])


@pytest.mark.parametrize("n", [1, 77, 5000, 1_200_000])
def test_radix_sort_u32(n):
    from adgs import _lib
    lib = _lib.lib()
    rng = np.random.RandomState(n % 999)
    keys = rng.randint(0, 1 << 30, size=n).astype(np.uint32)
    keys[:: 3] = keys[0]
    vals = np.arange(n, dtype=np.uint32)
    kd = torch.from_numpy(keys.view(np.int32)).cuda(); vd = torch.from_numpy(vals.view(np.int32)).cuda()
    ko = torch.empty_like(kd); vo = torch.empty_like(vd)
    tmp = torch.empty(int(lib.adgs_test_sort_temp_bytes(n)), dtype=torch.uint8, device="cuda")
    assert lib.adgs_test_sort_pairs_u32(kd.data_ptr(), ko.data_ptr(), vd.data_ptr(), vo.data_ptr(), n, 30, tmp.data_ptr(), _stream()) == 0
    order = np.argsort(keys, kind="stable")
    np.testing.assert_array_equal(ko.cpu().numpy().view(np.uint32), keys[order])
    np.testing.assert_array_equal(vo.cpu().numpy().view(np.uint32), vals[order])


def _cloud(n, seed, clustered=False):
    g = torch.Generator().manual_seed(seed)
    p = torch.rand(n, 3, generator=g) * torch.tensor([40.0, 10.0, 80.0]) - torch.tensor([20.0, 2.0, 0.0])
    if clustered:
        p[: n // 2] = p[: n // 2] * 0.01 + torch.tensor([3.0, 1.0, 7.0])
    return p.float().contiguous()


@pytest.mark.parametrize("n,clustered", [(4, False), (5, False), (1000, False), (1025, True), (30_000, False), (30_000, True), (200_000, True)])
def test_knn_matches_oracle_bit_exact(n, clustered):
    from simple_knn._C import distCUDA2
    pts = _cloud(n, n, clustered)
    got = distCUDA2(pts.cuda()).cpu().numpy()
    ref = oracle.knn_dist2(pts.numpy())
    np.testing.assert_array_equal(got, ref)


def test_knn_duplicates_and_tiny_inputs():
    from simple_knn._C import distCUDA2
    pts = _cloud(2000, 3)
    pts[100:110] = pts[100]            # duplicates give zero distances
    got = distCUDA2(pts.cuda()).cpu().numpy()
    np.testing.assert_array_equal(got, oracle.knn_dist2(pts.numpy()))
    assert got[100] == 0.0
    assert distCUDA2(torch.zeros(0, 3, device="cuda")).numel() == 0
    # P < 4 leaves FLT_MAX terms in the mean
    small = distCUDA2(_cloud(3, 1).cuda()).cpu().numpy()
    np.testing.assert_array_equal(small, oracle.knn_dist2(_cloud(3, 1).numpy()))
    with pytest.raises(RuntimeError):
        distCUDA2(pts)                 # CPU tensor: loud failure


@pytest.mark.parametrize("seed", range(int(os.environ.get("ADGS_TEST_SEED_BASE", "0")), int(os.environ.get("ADGS_TEST_SEED_BASE", "0")) + int(os.environ.get("ADGS_TEST_SIMPLE_KNN_SEEDS", "10"))))
def test_dist2_fuzz_bit_exact(seed):
    """distCUDA2 on random sizes around the 1024-point box size, lattice clouds (coincident points, equal Morton codes, exact
    distance ties), flat clouds (one coordinate constant) and wildly different scales."""
    from simple_knn._C import distCUDA2
    rng = np.random.default_rng(3000 + seed)
    n = int(rng.choice([1, 2, 3, 4, 7, 1023, 1024, 1025, 2048, 3000, 5121, 20_000]))
    p = rng.normal(size=(n, 3)).astype(np.float32) * float(rng.choice([1e-3, 1.0, 1e3]))
    kind = int(rng.integers(4))
    if kind == 1:
        p = (np.round(p / np.abs(p).max() * 6) / 6).astype(np.float32)
    elif kind == 2:
        p[:, int(rng.integers(3))] = 0.25
    elif kind == 3:
        p[: n // 2] = p[: n // 2] * 0.001 + p[0]
    got = distCUDA2(torch.tensor(p).cuda()).cpu().numpy()
    np.testing.assert_array_equal(got, oracle.knn_dist2(p), err_msg=str((n, kind)))
