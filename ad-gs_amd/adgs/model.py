"""Host-side mirror of the render-time interface of the reference GaussianModel
(scene/gaussian_model.py:27-231): the same attribute names for the raw parameters and the same
getters (`get_xyz`, `get_scaling`, `get_obj_mask`, `get_deformed_xyz(t)`, `get_deformed_pkg(t)`,
`active_sh_degree`), with the per-frame work done by the fused HIP deformation kernels
(adgs.deform) instead of ~30 small PyTorch kernels.  Training-side machinery of the reference
(densify, optimizer surgery, PLY I/O) is out of scope.
"""
import torch

from . import deform

DEFAULT_ORDER_ARGS = dict(xyz=[6, 5, 0, 6, 0, 0], rotation=[0, 0, 0, 0, 6, 5], shs=[0, 0, 0, 6, 0, 0], background=[0] * 6)

_RAW = ["_scene_xyz", "_obj_xyz", "_scene_shs_dc", "_obj_shs_dc", "_scene_shs_rest", "_obj_shs_rest", "_scene_scaling", "_obj_scaling",
        "_scene_rotation", "_obj_rotation", "_scene_opacity", "_obj_opacity", "xyz_deform_param", "rotation_deform_param",
        "shs_deform_param_scene", "shs_deform_param_obj", "background_deform_param", "gs_time_sigma"]


class SyntheticGaussianModel:
    def __init__(self, sh_degree, order_args):
        self.active_sh_degree = sh_degree
        self.max_sh_degree = sh_degree
        self.order_args = {k: list(v) for k, v in order_args.items()}
        self.use_time_mask = False
        self.gs_time = None

    @classmethod
    def from_scene(cls, sc, device, seed=0, order_args=None, use_time_mask=True, deform_scale=(0.05, 0.05, 0.01)):
        """Split an activated synthetic scene (adgs.synthetic.make_scene; objects are the LAST
        Gaussians) into the reference's scene/object raw parameters (inverse activations) and draw
        the deformation parameters of SURVEY.md 8(d)."""
        oa = order_args or DEFAULT_ORDER_ARGS
        m = cls(sc["sh_degree"], oa)
        g = torch.Generator().manual_seed(1234 + int(seed))
        obj = sc["obj_mask"]
        No = int(obj.sum())
        Ns = sc["P"] - No
        assert not obj[:Ns].any() and obj[Ns:].all(), "objects must be the trailing Gaussians"
        leaf = lambda t: t.to(device).contiguous().requires_grad_(True)
        sp = lambda t: (t[:Ns], t[Ns:])
        xs, xo = sp(sc["means3D"]); m._scene_xyz, m._obj_xyz = leaf(xs), leaf(xo)
        ss, so = sp(torch.log(sc["scales"])); m._scene_scaling, m._obj_scaling = leaf(ss), leaf(so)
        rs, ro = sp(sc["rotations"]); m._scene_rotation, m._obj_rotation = leaf(rs), leaf(ro)
        op = sc["opacities"].clamp(1e-6, 1 - 1e-6)
        os_, oo = sp(torch.log(op / (1 - op))); m._scene_opacity, m._obj_opacity = leaf(os_), leaf(oo)
        ds, dob = sp(sc["shs"][:, :1]); m._scene_shs_dc, m._obj_shs_dc = leaf(ds), leaf(dob)
        hs, ho = sp(sc["shs"][:, 1:]); m._scene_shs_rest, m._obj_shs_rest = leaf(hs), leaf(ho)
        u = lambda *s: torch.rand(*s, generator=g) * 2 - 1
        m.xyz_deform_param = leaf(u(No, 3, deform.get_param_num(oa["xyz"])) * deform_scale[0])
        m.rotation_deform_param = leaf(u(No, 4, deform.get_param_num(oa["rotation"])) * deform_scale[1])
        sd = u(sc["P"], 3, deform.get_param_num(oa["shs"])) * deform_scale[2]
        m.shs_deform_param_scene, m.shs_deform_param_obj = leaf(sd[:Ns]), leaf(sd[Ns:])
        m.background_deform_param = leaf(u(1, 3, deform.get_param_num(oa["background"])) * 1e-5)
        m.gs_time = torch.rand(No, 1, generator=g).to(device)
        m.gs_time_sigma = leaf(torch.full((No, 2), -1.0))          # exp(-1) ~ 0.37 of the sequence
        m.use_time_mask = use_time_mask
        return m

    # ---- reference getters (scene/gaussian_model.py:88-231) ----
    def parameters(self):
        return [getattr(self, n) for n in _RAW if getattr(self, n, None) is not None and getattr(self, n).numel() > 0]

    def zero_grad(self):
        for p in self.parameters():
            p.grad = None

    @property
    def get_scene_pts_num(self):
        return self._scene_xyz.shape[0]

    @property
    def get_obj_pts_num(self):
        return self._obj_xyz.shape[0]

    @property
    def get_pts_num(self):
        return self.get_scene_pts_num + self.get_obj_pts_num

    @property
    def get_xyz(self):
        return torch.cat([self._scene_xyz, self._obj_xyz], dim=0)

    @property
    def get_scaling(self):
        return deform.get_deformed_pkg(self, 0.0, want=("scales",))["scales"]

    @property
    def get_obj_mask(self):
        dev = self._scene_xyz.device
        return torch.cat([torch.zeros(self.get_scene_pts_num, dtype=torch.bool, device=dev),
                          torch.ones(self.get_obj_pts_num, dtype=torch.bool, device=dev)], dim=0)

    def get_deformed_xyz(self, t):
        return deform.get_deformed_xyz(self, t)

    raw_sh = False     # True: get_deformed_pkg leaves the SH coefficients in raw layout (RawSH) for forward_rawsh

    supports_fused_flow = True     # get_deformed_pkg(t, flow_time=...) also returns 'flow_xyz'

    def get_deformed_pkg(self, t, flow_time=None):
        """Reference keys 'xyz','rotation','shs','opacity' plus 'scales' (so that render() needs no
        separate get_scaling pass) and, with flow_time, 'flow_xyz' = get_deformed_xyz(flow_time)."""
        return deform.get_deformed_pkg(self, t, raw_sh=self.raw_sh, flow_time=flow_time)

    def deform_bytes_per_frame(self):
        """Algorithmic bytes of the deformation stage per frame: deformation
        parameters read forward + their gradients written backward."""
        n = sum(getattr(self, k).numel() for k in ("xyz_deform_param", "rotation_deform_param", "shs_deform_param_scene",
                                                   "shs_deform_param_obj", "background_deform_param"))
        return 2 * 4 * n
