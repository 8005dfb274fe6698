"""The drop-in pieces together: render() (deformation, flow, semantic, environment map) -> the losses and regularisers of
train.py:78-115 -> backward -> densification statistics -> neighbour index / densify_and_prune -> fused Adam, for a few
iterations on a small synthetic scene (examples/train_iteration.py)."""
import importlib.util
import os

import pytest
import torch

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_a_few_training_iterations_run_and_reduce_the_loss():
    spec = importlib.util.spec_from_file_location("train_iteration", os.path.join(ROOT, "examples", "train_iteration.py"))
    ti = importlib.util.module_from_spec(spec); spec.loader.exec_module(ti)
    from adgs import synthetic
    synthetic.CONFIGS["T0"] = dict(P=6000, W=208, H=130, focal=150.0, sh_degree=3, n_objects=2, seed=9)
    try:
        cfg, model, cams, env_map = ti.build("T0", 256, torch.device("cuda", 0), n_cameras=2)
        assert model.obj_near_idx.shape == (model.get_obj_pts_num // 8, 8)
        with torch.no_grad():                       # make the targets reachable: the scene's own first renders
            from gaussian_renderer import render
            import types
            for cam in cams:
                pkg = render(cam, model, env_map, types.SimpleNamespace(inv_depth=True, debug=False), flow_pkg=cam.flow[0], render_objmask=True)
                cam.original_image = (pkg["render"] * 0.8 + 0.1).clamp(0, 1)
        state, clock = {}, ti.StageClock(True)
        ti.OPT.densification_interval = 8           # one densify_and_prune and one neighbour-index reset inside the run
        ti.OPT.near_idx_reset_interval = 5
        l1 = []
        for it in range(12):
            total = ti.iteration(it, model, cams, env_map, clock, state)
            assert torch.isfinite(total)
            l1.append(float(state["l1"]))
        torch.cuda.synchronize()
        stages = clock.summary()
    finally:
        del synthetic.CONFIGS["T0"]
    assert all(l == l for l in l1) and min(l1[-2:]) < min(l1[:2]), l1           # L1 to the targets goes down (two cameras alternate)
    assert set(ti.STAGES) <= set(stages) and all(v >= 0 for v in stages.values()), stages
    assert state.get("densified") == 1
    assert float(model.denom.sum()) > 0 and float(model.max_radii2D.max()) > 0
    for p in model.parameters():
        assert torch.isfinite(p).all()
    assert torch.isfinite(env_map.grid_map).all() and float(env_map.grid_map.abs().max()) > 1e-4


def test_densification_cycle_on_the_model_class():
    """training_setup -> a few render/backward/stats/Adam iterations -> densify_and_prune -> reset_opacity -> neighbour index ->
    save_ply/load_ply, all through SyntheticGaussianModel's reference-named methods, then the renderer again on the new set."""
    import tempfile
    import types
    from adgs import synthetic
    from adgs.model import SyntheticGaussianModel
    from gaussian_renderer import render
    dev = torch.device("cuda", 0)
    sc = synthetic.make_scene(5000, 208, 130, 150.0, sh_degree=3, seed=4, n_objects=2)
    model = SyntheticGaussianModel.from_scene(sc, dev, seed=2)
    model.raw_sh = True
    model.training_setup(percent_dense=0.01, scene_extent=20.0, object_extent=4.0, near_num=8)
    assert model.obj_near_idx.shape == (model.get_obj_pts_num // 8, 8)
    cam = synthetic.camera_object(sc, time=0.4)
    pipe = types.SimpleNamespace(inv_depth=True, debug=False)
    for it in range(3):
        pkg = render(cam, model, None, pipe, flow_pkg=(0.45,) + (None,) * 5, render_objmask=True)
        (pkg["render"].mean() + pkg["depth"].mean() * 0.1).backward()
        model.add_densification_stats(pkg)
        model.optimizer.step()
        model.optimizer.zero_grad(set_to_none=True)
    assert float(model.denom.sum()) > 0
    n0 = model.get_pts_num
    thr = float((model.xyz_gradient_accum / model.denom.clamp_min(1)).quantile(0.9))
    torch.manual_seed(0)
    info = model.densify_and_prune(thr, thr, 0.005, True)
    n1 = model.get_pts_num
    assert n1 != n0 and n1 == info["scene"][2] + info["obj"][2]
    assert model.xyz_gradient_accum.shape == (n1, 1) and model.max_radii2D.shape == (n1,) and model.gs_time.shape[0] == model.get_obj_pts_num
    assert model.obj_near_idx.shape == (model.get_obj_pts_num // 8, 8)
    model.reset_opacity()
    assert float(torch.sigmoid(model._scene_opacity).max()) <= 0.0100001
    pkg = render(cam, model, None, pipe, flow_pkg=(0.45,) + (None,) * 5, render_objmask=True)
    pkg["render"].mean().backward()
    model.optimizer.step()
    assert pkg["radii"].shape[0] == n1 and torch.isfinite(pkg["render"]).all()
    for p in model.parameters():
        assert torch.isfinite(p).all()
    with tempfile.TemporaryDirectory() as d:
        model.save_ply(d + "/point_cloud.ply")
        other = SyntheticGaussianModel(3, model.order_args)
        other._scene_xyz = torch.zeros(0, 3, device=dev)
        other.load_ply(d + "/point_cloud.ply")
        assert torch.equal(other._obj_shs_rest, model._obj_shs_rest) and torch.equal(other.xyz_deform_param, model.xyz_deform_param)


def test_data_parallel_training_example_on_one_gpu(monkeypatch):
    """examples/train_dp.py with three cameras accumulated on one GPU through the factored exchange, including a densify step."""
    import sys
    spec = importlib.util.spec_from_file_location("train_dp", os.path.join(ROOT, "examples", "train_dp.py"))
    td = importlib.util.module_from_spec(spec); spec.loader.exec_module(td)
    from adgs import synthetic
    synthetic.CONFIGS["T1"] = dict(P=6000, W=208, H=130, focal=150.0, sh_degree=3, n_objects=2, seed=9)
    monkeypatch.setattr(sys, "argv", ["train_dp.py", "--config", "T1", "--iters", "6", "--cams", "3", "--densify-every", "4"])
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK"):
        monkeypatch.delenv(k, raising=False)
    try:
        losses, same = td.main()
    finally:
        del synthetic.CONFIGS["T1"]
    assert same and all(l == l for l in losses) and len(losses) == 9


def test_side_stream_gives_the_same_results_as_the_default_stream():
    """Every launch, copy and allocation of the path follows torch's CURRENT stream (the library takes the stream as an argument, the
    mailbox read-back waits on it): the rasterizer and a few whole training iterations under `torch.cuda.stream(side)` must reproduce
    the default-stream run -- a launch that slipped onto the null stream would race with its neighbours here."""
    import sys
    import numpy as np
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    from test_gpu_raster import assert_close, run_hip
    from adgs import synthetic
    sc = synthetic.make_scene(5000, 310, 190, 170.0, sh_degree=3, seed=21, n_objects=2)
    g = synthetic.make_upstream_grads(sc, 21)
    ref = run_hip(sc, grads=g)
    side = torch.cuda.Stream()
    for _ in range(3):
        with torch.cuda.stream(side):
            got = run_hip(sc, grads=g)
        assert torch.equal(got["radii"], ref["radii"])
        for k in ("color", "depth", "img_opacity", "img_flow", "img_semantic"):
            assert torch.equal(got[k], ref[k]), k                    # the forward is deterministic
        for k, v in ref["grads"].items():
            if v is not None:
                assert_close("grad " + k, got["grads"][k].cpu().numpy(), v.cpu().numpy(), max_frac=2e-4)

    spec = importlib.util.spec_from_file_location("train_iteration", os.path.join(ROOT, "examples", "train_iteration.py"))
    ti = importlib.util.module_from_spec(spec); spec.loader.exec_module(ti)
    synthetic.CONFIGS["T2"] = dict(P=6000, W=208, H=130, focal=150.0, sh_degree=3, n_objects=2, seed=9)
    try:
        runs = []
        for stream in (None, side):
            ctx = torch.cuda.stream(stream) if stream is not None else torch.cuda.stream(torch.cuda.default_stream())
            with ctx:
                torch.manual_seed(0)                 # the neighbour-index anchors are drawn with torch.randperm
                cfg, model, cams, env_map = ti.build("T2", 256, torch.device("cuda", 0), n_cameras=2)
                state = {}
                losses = [float(ti.iteration(it, model, cams, env_map, ti.StageClock(False), state)) for it in range(6)]
                torch.cuda.current_stream().synchronize()
            runs.append(losses)
    finally:
        del synthetic.CONFIGS["T2"]
    assert np.allclose(runs[0], runs[1], rtol=2e-4, atol=1e-7), runs
