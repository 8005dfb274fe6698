"""The drop-in pieces together: render() (deformation, flow, semantic, environment map) -> fused L1+SSIM -> backward ->
densification statistics -> fused Adam, for a few iterations on a small synthetic scene (examples/train_iteration.py)."""
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
        cfg, model, cam, env_map, stats, targets = ti.build("T0", 256, torch.device("cuda", 0))
        with torch.no_grad():                       # make the target reachable: the scene's own first render
            from gaussian_renderer import render
            import types
            pkg = render(cam, model, env_map, types.SimpleNamespace(inv_depth=True, debug=False), flow_pkg=(cam.time + 0.05,) + (None,) * 5, render_objmask=True)
            targets["image"] = (pkg["render"] * 0.8 + 0.1).clamp(0, 1)
        losses = [float(ti.iteration(model, cam, env_map, stats, targets)[1]) for _ in range(12)]
    finally:
        del synthetic.CONFIGS["T0"]
    assert all(l == l for l in losses) and losses[-1] < losses[0], losses           # L1 to the target goes down
    assert float(stats["denom"].sum()) > 0 and float(stats["max_r"].max()) > 0
    for p in model.parameters():
        assert torch.isfinite(p).all()
    assert torch.isfinite(env_map.grid_map).all() and float(env_map.grid_map.abs().max()) > 1e-4
