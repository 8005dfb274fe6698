"""The forward without a host decision in its middle (replaces the blocking read of rasterizer_impl.cu:288): the whole frame is
enqueued against a capacity, the device compares the totals, the host validates afterwards.

  * eager: a frame that does not fit the capacity hint is enqueued again with exact sizes before forward() returns -- results equal
    the oracle's, `eager_reruns` counts it;
  * HIP graph replay (adgs.graph): a captured frame equals the eager frame bit for bit; a replay whose frame outgrew the captured
    capacity is reported by adgs_get_frame_status, leaves an empty replay state (no fault), and a re-capture repairs it;
  * the forward state is keyed on (image buffer, geometry buffer, W, H, P): a backward over cloned state buffers falls back to
    today's environment and zeroes the accumulator lines itself.
"""
import os
import sys

import numpy as np
import pytest
import torch

from adgs import _lib, synthetic, graph

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
from tests.test_gpu_raster import compare, run_hip, dev  # noqa: E402

pytestmark = pytest.mark.gpu


def same(a, b, what=""):
    """Forward outputs are deterministic (bit-equal); gradients pass through float atomics whose order differs from run to run."""
    if a.dtype != torch.float32 or not a.is_floating_point():
        assert torch.equal(a, b), what
        return
    scale = float(b.abs().max()) if b.numel() else 0.0
    assert torch.allclose(a, b, rtol=2e-3, atol=2e-5 * scale + 1e-30), (what, float((a - b).abs().max()), scale)


def test_eager_frame_beyond_the_capacity_hint_is_enqueued_again():
    _lib.lib().adgs_test_set_capacity_hints(0, 0)      # as in a fresh process: the capacity is the floor P + 4096 pairs
    before = _lib.frame_status()["eager_reruns"]
    big = synthetic.make_scene(60000, 400, 300, 300.0, seed=61, scale_mult=0.02)      # large splats: many (cell, Gaussian) pairs
    compare(big, grads=synthetic.make_upstream_grads(big, 61))
    st = _lib.frame_status()
    assert st["eager_reruns"] > before, "the big frame was expected to exceed the decayed capacity hint"
    before = st["eager_reruns"]
    compare(big, grads=synthetic.make_upstream_grads(big, 61))                       # hint raised: fits now
    assert _lib.frame_status()["eager_reruns"] == before


def _static_step(sc, scale=None):
    """forward + backward on static tensors; returns (fn, leaves)."""
    from diff_gaussian_rasterization import GaussianRasterizationSettings, GaussianRasterizer
    s = GaussianRasterizationSettings(sc["H"], sc["W"], sc["tanfovx"], sc["tanfovy"], dev(sc["bg"]), 1.0, dev(sc["viewmatrix"]),
                                      dev(sc["projmatrix"]), sc["sh_degree"], dev(sc["campos"]), False, True, False)
    rast = GaussianRasterizer(s)
    leaf = {k: dev(sc[k]).clone().requires_grad_(True) for k in ("means3D", "opacities", "shs", "scales", "rotations")}
    means2D = torch.zeros(sc["P"], 3, device="cuda", requires_grad=True)
    flow, sem = dev(sc["flow_points"]), dev(sc["semantic"])
    g = synthetic.make_upstream_grads(sc, 3)
    ups = [dev(g[k]) for k in ("color", "depth", "img_opacity", "flow", "semantic")]

    def fn():
        color, radii, depth, op, fl, se = rast(means3D=leaf["means3D"], means2D=means2D, opacities=leaf["opacities"], shs=leaf["shs"],
                                               scales=leaf["scales"], rotations=leaf["rotations"], flow_points=flow, semantic=sem)
        torch.autograd.backward([color, depth, op, fl, se], ups)
        # detached: a result that keeps the autograd graph (and its AccumulateGrad nodes, bound to the stream they were created on)
        # alive across calls breaks a later capture of the same leaves (torch.cuda.graphs, "whole-network capture" notes)
        out = [color.detach(), depth.detach(), op.detach(), radii] + [leaf[k].grad for k in leaf] + [means2D.grad]
        for t in list(leaf.values()) + [means2D]:
            t.grad = None
        return out
    return fn, leaf


def test_graph_replay_equals_the_eager_frame_bit_for_bit():
    sc = synthetic.make_scene(10000, 400, 300, 300.0, sh_degree=3, seed=62, n_objects=2)
    fn, _ = _static_step(sc)
    eager = [t.clone() for t in fn()]
    step = graph.GraphedStep(fn)
    for _ in range(3):
        got = step()
    torch.cuda.synchronize()
    assert step.validate(repair=False)
    for i, (a, b) in enumerate(zip(eager, got)):
        if i < 4:
            assert torch.equal(a, b), i          # colour, depth, accumulated opacity, radii
        else:
            same(a, b, i)


def test_graph_replay_of_the_deformation_and_raw_parameter_path():
    import bench
    cfg = synthetic.CONFIGS["T3"]
    sc = synthetic.make_config_scene("T3")
    device = torch.device("cuda", 0)
    pool = bench.camera_pool(cfg, 3)
    frames = bench.frame_pool(sc, cfg, pool, device, True)
    g = synthetic.make_upstream_grads(sc, 0)
    ups = [dev(g[k]) for k in ("color", "depth", "img_opacity", "flow", "semantic")]
    eager = []
    for f in frames:
        torch.autograd.backward(f.forward(), ups)
        eager.append([p.grad.clone() for p in f.parameters()])
        f.zero_grad()
    cache = bench.graphed_steps(frames, ups)
    for rep in range(2):
        for k in range(len(frames)):
            got = cache(k)
            torch.cuda.synchronize()
            for a, b in zip(eager[k], got):
                same(a, b, "camera %d, replay %d" % (k, rep))
    assert cache.validate(repair=False)


def test_graph_replay_that_outgrows_its_capacity_is_reported_and_repaired():
    sc = synthetic.make_scene(20000, 400, 300, 300.0, sh_degree=1, seed=63)
    fn, leaf = _static_step(sc)
    _lib.lib().adgs_test_set_capacity_hints(0, 0)
    fn()                                      # the capacity hint settles on this frame (+ 25 %)
    step = graph.GraphedStep(fn)
    step()
    torch.cuda.synchronize()
    assert step.validate(repair=False)
    ok_pairs = _lib.frame_status()["pairs"]
    with torch.no_grad():
        leaf["scales"].mul_(6.0)              # the same static tensors, six times larger splats: several times the pairs
    step()
    torch.cuda.synchronize()
    st = _lib.frame_status()
    assert st["overflow"] == 1 and st["pairs"] > st["capacity_pairs"] >= ok_pairs
    assert not step.validate(repair=True)     # reported, and re-captured after an eager frame
    got = step()
    torch.cuda.synchronize()
    assert step.validate(repair=False)
    got = [t.clone() for t in got]
    want = fn()
    torch.cuda.synchronize()
    for i, (a, b) in enumerate(zip(want, got)):
        if i < 4:
            assert torch.equal(a, b), i
        else:
            same(b, a, i)


def test_backward_over_cloned_state_buffers_does_not_trust_a_stale_forward_entry():
    """autograd may hand the backward COPIES of the saved state (save_on_cpu, checkpointing): same bytes, new addresses."""
    from diff_gaussian_rasterization import _C
    sc = synthetic.make_scene(4000, 200, 136, 150.0, seed=64)
    e = torch.Tensor([])
    args = (dev(sc["bg"]), dev(sc["means3D"]), e, dev(sc["opacities"]), dev(sc["scales"]), dev(sc["rotations"]), 1.0, e, dev(sc["viewmatrix"]),
            dev(sc["projmatrix"]), sc["tanfovx"], sc["tanfovy"], sc["H"], sc["W"], dev(sc["shs"]), dev(sc["flow_points"]), dev(sc["semantic"]), 3,
            dev(sc["campos"]), False, True, False)
    R, color, depth, op, radii, geom, binning, img, fl, se = _C.rasterize_gaussians(*args)
    g = synthetic.make_upstream_grads(sc, 64)

    def backward(geom_, bin_, img_):
        return _C.rasterize_gaussians_backward(args[0], args[1], radii, e, args[4], args[5], 1.0, e, args[8], args[9], args[10], args[11],
                                               dev(g["color"]), dev(g["depth"]), dev(g["flow"]), dev(g["semantic"]), args[16], args[15], args[14], 3,
                                               args[18], geom_, R, bin_, img_, op, dev(g["img_opacity"]), True, False)
    want = [t.clone() for t in backward(geom, binning, img)]
    for rep in range(2):                       # clones: not in the frame table -> accumulator lines zeroed by the backward itself
        got = backward(geom.clone(), binning.clone(), img.clone())
        for a, b in zip(want, got):
            same(b, a, rep)
