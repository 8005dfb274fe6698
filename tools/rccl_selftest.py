#!/usr/bin/env python3
"""RCCL self-test on a ONE-GPU box: a one-rank "nccl" process group, the collectives of adgs.dp forced on, and the result of
the factored exchange compared with the plain one-process result.  Checks what the gloo tests cannot: that RCCL accepts
the buffer views the exchange hands it and that the stream ordering between the backward, the collectives and the
expansion kernel is right.  (Two ranks cannot share one GPU under RCCL, so world size 1 is all a 1-GPU box offers.)"""
import os
import sys
import time

import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "ad-gs_amd")):
    sys.path.insert(0, p)
os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29561")
from adgs import dp, synthetic  # noqa: E402
from adgs.model import SyntheticGaussianModel  # noqa: E402
from diff_gaussian_rasterization import GaussianRasterizationSettings, GaussianRasterizer  # noqa: E402


N_IT, N_TIMED = int(os.environ.get("SELFTEST_ITERS", "30")), 20


def main():
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(dev)
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
    P = int(os.environ.get("SELFTEST_P", "200000"))
    sc = synthetic.make_scene(P, 640, 400, 600.0, sh_degree=3, seed=7, n_objects=4)
    cams = [synthetic.make_camera(640, 400, 600.0, cam_seed=c) for c in range(2)]
    times = [0.2, 0.6]
    up = synthetic.make_upstream_grads(sc, 1)
    ups = [up[k].to(dev) for k in ("color", "depth", "img_opacity", "flow", "semantic")]

    def run(factored, force):
        model = SyntheticGaussianModel.from_scene(sc, dev, seed=1)
        model.raw_sh = True
        ex = dp.FactoredSHExchange(model)
        ex.force_collectives = force
        t0 = None
        for it in range(N_IT):
            if it == N_IT - N_TIMED:
                torch.cuda.synchronize(); t0 = time.perf_counter()
            model.zero_grad()
            for cam, t in zip(cams, times):
                s = GaussianRasterizationSettings(cam["H"], cam["W"], cam["tanfovx"], cam["tanfovy"], torch.zeros(3, device=dev), 1.0, cam["viewmatrix"].to(dev),
                                                  cam["projmatrix"].to(dev), 3, cam["campos"].to(dev), False, True, False)
                pkg = model.get_deformed_pkg(t, flow_time=t + 0.05)
                m2 = torch.zeros_like(pkg["xyz"], requires_grad=True)
                outs = GaussianRasterizer(s).forward_rawsh(pkg["xyz"], m2, pkg["opacity"], pkg["shs"], pkg["scales"], pkg["rotation"], flow_points=pkg["flow_xyz"],
                                                           semantic=model.get_obj_mask.float()[:, None].contiguous(),
                                                           factor_sink=ex.sink_for(pkg["xyz"]) if factored else None)
                torch.autograd.backward([outs[0], outs[2], outs[3], outs[4], outs[5]], ups)
            if factored:
                ex.reduce(times, [c["campos"].tolist() for c in cams])
            elif force:
                dp.allreduce_gradients(model.parameters(), force=True)
        torch.cuda.synchronize()
        return {n: getattr(model, n).grad.clone() for n in dp._SH_PARAMS + ("_scene_xyz", "_obj_xyz", "xyz_deform_param", "rotation_deform_param")}, (time.perf_counter() - t0) / N_TIMED

    ref, t_ref = run(False, False)
    for name, (fac, force) in dict(dense_rccl=(False, True), factored_local=(True, False), factored_rccl=(True, True)).items():
        got, t = run(fac, force)
        worst = max(float((got[k] - ref[k]).abs().max() / ref[k].abs().max().clamp_min(1e-30)) for k in ref)
        print("%-15s max rel deviation from the plain accumulation %.2e   %.2f ms/iteration (plain %.2f)" % (name, worst, t * 1e3, t_ref * 1e3))
        assert worst < 1e-4, name
    dist.destroy_process_group()
    print("rccl selftest ok")


if __name__ == "__main__":
    main()
