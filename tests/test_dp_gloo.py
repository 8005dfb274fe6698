"""Multi-process (gloo, world_size 2, CPU) tests of the camera-parallel data-parallel layer
(ad-gs_amd/adgs/dp.py).  The rasterizer itself has no CPU path, so the per-camera
"render + loss" here is a small differentiable torch function (the DP layer is agnostic to it);
the parity statement is the one that matters for training: k cameras sharded over ranks with a
gradient all-reduce == single-process gradient accumulation over the same k cameras."""
import os
import socket
import sys

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for _p in (ROOT, os.path.join(ROOT, "ad-gs_amd")):
    if _p not in sys.path:
        sys.path.insert(0, _p)


def _free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close(); return p


def _params(seed=0):
    g = torch.Generator().manual_seed(seed)
    return [torch.randn(2000, 3, generator=g, requires_grad=True), torch.randn(2000, 16, 3, generator=g, requires_grad=True),
            torch.randn(7, generator=g, requires_grad=True), torch.randn(1_200_000, generator=g, requires_grad=True)]


def _camera_loss(params, cam):
    xyz, shs, small, big = params
    g = torch.Generator().manual_seed(100 + cam)
    w = torch.randn(2000, generator=g)
    return ((xyz.sum(1) * w).tanh().sum() + (shs[:, cam % 16] * w[:, None]).sum() * 0.1 + (small * (cam + 1)).sum()
            + (big[cam::7] ** 2).sum() * 1e-3)


def _worker(rank, world, port, cams, out_dir):
    for p in (ROOT, os.path.join(ROOT, "ad-gs_amd")):
        sys.path.insert(0, p)
    from adgs import dp
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    params = _params()
    total = dp.multi_camera_step(lambda c: _camera_loss(params, c), cams, params)
    # densification statistics and shared RNG
    acc = torch.full((10, 1), float(rank + 1)); den = torch.full((10, 1), 1.0); rad = torch.arange(10, dtype=torch.float32) * (rank + 1)
    dp.allreduce_densification_stats(acc, den, rad)
    # both reduction paths with explicit thresholds: in place (>= 1 MiB here), several flat buckets (16 KiB each), and a
    # parameter that has a gradient on one rank only
    extra = [torch.zeros(300_000, requires_grad=True), torch.zeros(3000, requires_grad=True), torch.zeros(5000, requires_grad=True),
             torch.zeros(11, requires_grad=True)]
    for i, p in enumerate(extra):
        if not (i == 3 and rank == 1):
            p.grad = torch.full_like(p, float((rank + 1) * (i + 1)))
    dp.allreduce_gradients(extra, in_place_bytes=1 << 20, bucket_bytes=16 << 10)
    # one flat buffer through both forms of the collective (ADGS_DP_COLLECTIVE): all_reduce and reduce-scatter + all-gather
    flats = {}
    for mode in ("all_reduce", "rs_ag"):
        os.environ["ADGS_DP_COLLECTIVE"] = mode
        f = torch.arange(dp.ARENA_QUANTUM, dtype=torch.float32) * (rank + 1)
        for w in dp.reduce_flat(f, None, {}):
            w.wait()
        flats[mode] = f
    os.environ.pop("ADGS_DP_COLLECTIVE")
    seed = dp.seed_all_ranks(1234 + rank)           # rank 0's value wins
    draw = torch.randn(4)
    torch.save(dict(grads=[p.grad.clone() for p in params], total=None if total is None else total.clone(), acc=acc, den=den, rad=rad,
                    seed=seed, draw=draw, mine=dp.shard_cameras(cams, rank, world), extra=[p.grad.clone() for p in extra], flats=flats), os.path.join(out_dir, "r%d.pt" % rank))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("cams", [[0, 1], [0, 1, 2], [5]])
def test_sharded_cameras_equal_single_process_accumulation(tmp_path, cams):
    world = 2
    port = _free_port()
    mp.spawn(_worker, args=(world, port, cams, str(tmp_path)), nprocs=world, join=True)
    res = [torch.load(os.path.join(str(tmp_path), "r%d.pt" % r)) for r in range(world)]
    # reference: one process, loss = mean over cameras
    params = _params()
    for c in cams:
        (_camera_loss(params, c) / len(cams)).backward()
    for r in range(world):
        for g, p in zip(res[r]["grads"], params):
            torch.testing.assert_close(g, p.grad, rtol=1e-5, atol=1e-6)
    # replicas are bit-identical after the all-reduce
    for a, b in zip(res[0]["grads"], res[1]["grads"]):
        assert torch.equal(a, b)
    assert sorted(res[0]["mine"] + res[1]["mine"]) == sorted(cams)
    assert res[0]["mine"] == cams[0::2] and res[1]["mine"] == cams[1::2]
    # stats: sums for the accumulators, max for the radii
    assert torch.equal(res[0]["acc"], torch.full((10, 1), 3.0)) and torch.equal(res[1]["den"], torch.full((10, 1), 2.0))
    assert torch.equal(res[0]["rad"], torch.arange(10, dtype=torch.float32) * 2)
    assert res[0]["seed"] == res[1]["seed"] == 1234 and torch.equal(res[0]["draw"], res[1]["draw"])
    from adgs import dp as _dp
    for r in range(world):
        for mode in ("all_reduce", "rs_ag"):
            assert torch.equal(res[r]["flats"][mode], torch.arange(_dp.ARENA_QUANTUM, dtype=torch.float32) * 3), (r, mode)
    for r in range(world):
        for i, g in enumerate(res[r]["extra"]):
            want = 3.0 * (i + 1) if i != 3 else 4.0          # the last one only has rank 0's gradient (1 * 4)
            assert torch.equal(g, torch.full_like(g, want)), (r, i)


def test_single_process_is_a_no_op():
    sys.path.insert(0, os.path.join(ROOT, "ad-gs_amd"))
    from adgs import dp
    params = _params()
    _camera_loss(params, 0).backward()
    before = [p.grad.clone() for p in params]
    dp.allreduce_gradients(params)
    for a, p in zip(before, params):
        assert torch.equal(a, p.grad)
    assert dp.shard_cameras([1, 2, 3], 0, 1) == [1, 2, 3]
