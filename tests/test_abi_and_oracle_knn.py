"""CPU-side checks: the C-ABI library loads and exports every symbol declared in include/*.h
(no compute calls without a GPU), and the kNN oracle equals brute force."""
import ctypes
import os
import re

import numpy as np
import pytest

from oracle import oracle

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared_symbols():
    names = set()
    for h in sorted(os.listdir(os.path.join(ROOT, "include"))):
        p = os.path.join(ROOT, "include", h)
        if not os.path.exists(p):
            continue
        txt = open(p).read()
        txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
        names |= set(re.findall(r"\b(adgs_[a-z0-9_]+)\s*\(", txt))
    names.discard("adgs_alloc_fn")
    return names


def test_library_exports_every_declared_symbol():
    from adgs import _lib
    lib = _lib.lib()
    decl = _declared_symbols()
    assert {"adgs_raster_forward", "adgs_raster_backward", "adgs_mark_visible", "adgs_knn_dist2", "adgs_last_error"} <= decl
    for name in sorted(decl):
        assert hasattr(lib, name), "libadgs_hip.so does not export %s" % name
    # every declared symbol has a ctypes signature (and vice versa)
    assert decl == set(_lib.SIGNATURES.keys())


def test_workspace_queries_run_without_gpu():
    from adgs import _lib
    lib = _lib.lib()
    assert lib.adgs_knn_workspace_bytes(0) > 0
    assert lib.adgs_knn_workspace_bytes(100000) > 100000 * 4 * 5
    assert lib.adgs_test_sort_temp_bytes(1 << 20) >= 256 * 4 * ((1 << 20) // 4096)      # one histogram column per 4096-key block


def test_operators_fail_loudly_on_cpu_tensors():
    import torch
    from diff_gaussian_rasterization import GaussianRasterizationSettings, GaussianRasterizer
    from simple_knn._C import distCUDA2
    s = GaussianRasterizationSettings(16, 16, 1.0, 1.0, torch.zeros(3), 1.0, torch.eye(4), torch.eye(4), 0, torch.zeros(3), False, True, False)
    r = GaussianRasterizer(s)
    with pytest.raises(RuntimeError):
        r(means3D=torch.zeros(4, 3), means2D=torch.zeros(4, 3), opacities=torch.ones(4, 1), shs=torch.zeros(4, 1, 3),
          scales=torch.ones(4, 3), rotations=torch.ones(4, 4))
    with pytest.raises(RuntimeError):
        r.markVisible(torch.zeros(4, 3))
    with pytest.raises(RuntimeError):
        distCUDA2(torch.zeros(8, 3))


@pytest.mark.parametrize("n,seed", [(4, 0), (50, 1), (1024, 2), (3000, 3), (9000, 4)])
def test_knn_oracle_equals_bruteforce(n, seed):
    rng = np.random.RandomState(seed)
    pts = (rng.rand(n, 3) * [30, 5, 60] - [15, 1, 0]).astype(np.float32)
    if n > 100:
        pts[: n // 3] = pts[: n // 3] * 0.02 + 4.0       # dense cluster: box pruning is exercised
    a = oracle.knn_dist2(pts)
    b = oracle.knn_dist2(pts, bruteforce=True)
    np.testing.assert_array_equal(a, b)
    assert np.all(a >= 0)


def test_knn_oracle_known_answer():
    # unit grid line: neighbours at distance 1,1,2 (interior) -> mean of squares = (1+1+4)/3
    pts = np.stack([np.arange(10, dtype=np.float32), np.zeros(10, np.float32), np.zeros(10, np.float32)], 1)
    d = oracle.knn_dist2(pts)
    np.testing.assert_allclose(d[3:7], 2.0, rtol=1e-6)
    np.testing.assert_allclose(d[0], (1 + 4 + 9) / 3.0, rtol=1e-6)


def test_ctypes_mirrors_have_the_size_of_the_c_structs():
    """The structs that cross the ABI by pointer: a Python mirror that falls behind the header reads or writes past the C object
    (adgs_sh_grads grew to 12 members while one mirror kept 7)."""
    import ctypes
    from adgs import _lib, deform
    from adgs.optim import AdamGroup, ShAdam
    from diff_gaussian_rasterization._C import ShSource, ShGrads
    lib = _lib.lib()
    mirrors = {0: ShSource, 1: ShGrads, 2: _lib.FrameStats, 3: _lib.FrameStatus, 4: deform.FuncEval, 5: AdamGroup, 6: ShAdam}
    for which, cls in mirrors.items():
        assert ctypes.sizeof(cls) == lib.adgs_test_abi_sizeof(which), (which, cls.__name__)
