"""CPU checks of the factored SH-gradient exchange (include/adgs_exchange.h, adgs.dp.FactoredSHExchange).

1. The NumPy expansion oracle is pinned against the line-by-line C++ restatement of the reference backward: for one camera,
   expand(dL_dcolor * (1 - clamped)) must equal the oracle's dL_dsh (backward.cu:20-139).
2. The deform rows equal autograd through `dc + f_shs(t)` of the float64 torch restatement of get_func_result.
3. world_size-2 gloo: all-gathered factors + local expansion == all-reduce of the per-camera expanded gradients, and the
   replicas are bit-identical."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from oracle import exchange_oracle as xo
from oracle import oracle
from tests import torch_deform_ref as tr

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.parametrize("deg", [0, 1, 2, 3])
def test_expansion_oracle_equals_reference_backward_dL_dsh(deg):
    from adgs import synthetic
    sc = synthetic.make_scene(1500, 96, 64, 80.0, sh_degree=3, seed=3 + deg, n_objects=0)
    g = synthetic.make_upstream_grads(sc, 1)
    o = oracle.RasterOracle("f32")
    fwd = o.forward(sc["bg"], sc["means3D"], None, sc["opacities"], sc["scales"], sc["rotations"], 1.0, None, sc["viewmatrix"], sc["projmatrix"],
                    sc["tanfovx"], sc["tanfovy"], sc["H"], sc["W"], sc["shs"], None, None, deg, sc["campos"], False, True)
    st = o.state()
    rg = o.backward(g["color"], g["depth"], np.zeros((3, sc["H"], sc["W"]), np.float32), None, g["img_opacity"])
    vis = fwd["radii"] > 0
    assert vis.sum() > 200 and st["clamped"].sum() > 0, "the case must exercise visibility and the clamp mask"
    factor = rg["dL_dcolors"] * (1 - st["clamped"].astype(np.float32)) * vis[:, None]
    P = sc["P"]
    out = xo.expand([(factor, None, sc["campos"].numpy())], None, 0, P, P, P, sc["means3D"].numpy(), deg, 16)
    dsh = np.concatenate([out[0], out[2]], axis=1)
    scale = np.abs(rg["dL_dsh"]).max()
    np.testing.assert_allclose(dsh, rg["dL_dsh"], rtol=2e-5, atol=2e-6 * scale)
    assert np.abs(rg["dL_dsh"][:, (deg + 1) ** 2:]).max(initial=0.0) == 0.0


def _dense_w(times, oa, C):
    sys.path.insert(0, os.path.join(ROOT, "ad-gs_amd"))
    from adgs import dp
    return dp._dense_basis_weights(tuple(times), oa, C, "cpu").numpy()


@pytest.mark.parametrize("oa", [[0, 0, 0, 6, 0, 0], [9, 3, 2, 2, 0, 0]])
def test_deform_rows_equal_autograd_of_the_torch_restatement(oa):
    C = oa[0] + oa[2] + 2 * oa[3]
    times = [0.0, 0.37, 1.0]
    W = _dense_w(times, oa, C)
    rng = np.random.default_rng(0)
    P = 40
    rgbs = [rng.normal(size=(P, 3)) for _ in times]
    xyz = rng.normal(size=(P, 3)) + np.array([0, 0, 5.0])
    out = xo.expand([(r, None, np.zeros(3)) for r in rgbs], W, C, P, P, P, xyz, 0, 16)
    param = torch.zeros(P, 3, C, dtype=torch.float64, requires_grad=True)
    loss = sum((tr.get_func_result(t, param, oa) * torch.tensor(xo.SH_C0 * r)).sum() for t, r in zip(times, rgbs))
    loss.backward()
    # the product's basis weights are float32 (like the reference's), the restatement is float64
    np.testing.assert_allclose(out[4], param.grad.numpy(), rtol=1e-5, atol=1e-6 * np.abs(param.grad.numpy()).max())


# ---------------------------------------------------------------- gloo, world size 2
class _Model:
    """The attributes FactoredSHExchange reads from the reference GaussianModel."""

    def __init__(self, Ns, No, M, C, background=False, seed=0):
        g = torch.Generator().manual_seed(seed)
        r = lambda *s: torch.randn(*s, generator=g).requires_grad_(True)
        self._scene_xyz, self._obj_xyz = r(Ns, 3), r(No, 3)
        self._scene_shs_dc, self._obj_shs_dc = r(Ns, 1, 3), r(No, 1, 3)
        self._scene_shs_rest, self._obj_shs_rest = r(Ns, M - 1, 3), r(No, M - 1, 3)
        self.shs_deform_param_scene, self.shs_deform_param_obj = r(Ns, 3, C), r(No, 3, C)
        self._scene_opacity = r(Ns, 1)
        self.order_args = dict(shs=[0, 0, 0, C // 2, 0, 0], background=[0, 0, 1, 0, 0, 0] if background else [0] * 6)
        self.active_sh_degree = 3 if M == 16 else 1
        self.get_scene_pts_num, self.get_pts_num = Ns, Ns + No

    def parameters(self):
        return [self._scene_xyz, self._obj_xyz, self._scene_shs_dc, self._obj_shs_dc, self._scene_shs_rest, self._obj_shs_rest,
                self.shs_deform_param_scene, self.shs_deform_param_obj, self._scene_opacity]


def _oracle_expand(cams, W, C, P, Ns, row0, xyz_head, D, M, outs):
    res = xo.expand([(r.numpy(), None if t is None else t.numpy(), np.asarray(cp)) for r, t, cp in cams], None if W is None else W.numpy(), C, P, Ns,
                    row0, None if xyz_head is None else xyz_head.numpy(), D, M)
    for o, v in zip(outs, res):
        if o is not None:
            o.copy_(torch.tensor(v, dtype=torch.float32).reshape(o.shape))


def _camera(model, cam):
    """Synthetic per-camera backward: (time, campos, deformed means, colour-gradient factor, dense gradients)."""
    g = torch.Generator().manual_seed(500 + cam)
    P = model.get_pts_num
    t, campos = 0.1 + 0.2 * cam, [0.1 * cam, -0.2, -6.0]
    xyz = torch.cat([model._scene_xyz.detach(), model._obj_xyz.detach() + 0.05 * cam], 0)
    if any(model.order_args["background"]):
        xyz = xyz + 0.01 * cam
    rgb = torch.randn(P, 3, generator=g)
    rgb[torch.rand(P, generator=g) < 0.3] = 0.0                 # not visible from this camera
    dense = [torch.randn(p.shape, generator=g) for p in (model._scene_xyz, model._obj_xyz, model._scene_opacity)]
    return t, campos, xyz, rgb, dense


def _worker(rank, world, port, n_cams, background, out_dir):
    for p in (ROOT, os.path.join(ROOT, "ad-gs_amd")):
        sys.path.insert(0, p)
    from adgs import dp
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    model = _Model(50, 30, 16, 12, background)
    ex = dp.FactoredSHExchange(model, expand=_oracle_expand)
    cams = [_camera(model, c) for c in range(n_cams)]
    if n_cams == 3:
        ex.begin(n_cams)                      # announced camera count: the gather is issued by the last local append
    for c in dp.shard_cameras(list(range(n_cams)), rank, world):
        t, campos, xyz, rgb, dense = cams[c]
        sink = ex.sink_for(xyz)               # what forward_rawsh(factor_sink=...) + backward do on the HIP path
        sink.append(rgb)
        for p, gd in zip((model._scene_xyz, model._obj_xyz, model._scene_opacity), dense):
            p.grad = gd.clone() if p.grad is None else p.grad + gd
    ex.reduce([c[0] for c in cams], [c[1] for c in cams])
    torch.save([None if p.grad is None else p.grad.clone() for p in model.parameters()], os.path.join(out_dir, "r%d.pt" % rank))
    dist.barrier()
    dist.destroy_process_group()


def _free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close(); return p


@pytest.mark.parametrize("n_cams,background", [(2, False), (3, False), (1, False), (3, True)])
def test_gloo_factored_exchange_equals_dense_allreduce(tmp_path, n_cams, background):
    world = 2
    mp.spawn(_worker, args=(world, _free_port(), n_cams, background, str(tmp_path)), nprocs=world, join=True)
    res = [torch.load(os.path.join(str(tmp_path), "r%d.pt" % r)) for r in range(world)]
    # reference: every camera's gradients expanded on their own and summed (what a dense all-reduce would deliver)
    sys.path.insert(0, os.path.join(ROOT, "ad-gs_amd"))
    from adgs import dp
    model = _Model(50, 30, 16, 12, background)
    P, Ns = model.get_pts_num, model.get_scene_pts_num
    want = [torch.zeros_like(p, dtype=torch.float64) for p in model.parameters()]
    for c in range(n_cams):
        t, campos, xyz, rgb, dense = _camera(model, c)
        W = dp._dense_basis_weights((t,), model.order_args["shs"], 12, "cpu").numpy()
        one = xo.expand([(rgb.numpy(), xyz.numpy(), np.asarray(campos))], W, 12, P, Ns, 0, None, 3, 16)
        for i, v in zip((2, 3, 4, 5, 6, 7), one):
            want[i] += torch.tensor(v).reshape(want[i].shape)
        for i, gd in zip((0, 1, 8), dense):
            want[i] += gd.double()
    for r in range(world):
        for got, w in zip(res[r], want):
            torch.testing.assert_close(got.double(), w, rtol=1e-5, atol=1e-6)
    for a, b in zip(res[0], res[1]):
        assert torch.equal(a, b), "replicas must hold the same bits"
