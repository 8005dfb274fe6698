"""Host-side mirror of the render-time interface of the reference GaussianModel
(scene/gaussian_model.py:27-231): the same attribute names for the raw parameters and the same
getters (`get_xyz`, `get_scaling`, `get_obj_mask`, `get_deformed_xyz(t)`, `get_deformed_pkg(t)`,
`active_sh_degree`), with the per-frame work done by the fused HIP deformation kernels
(adgs.deform) instead of ~30 small PyTorch kernels.  The training-side methods that sit next to the hot
path keep their reference names too: training_setup (fused Adam with the reference's group names),
add_densification_stats, densify_and_prune, reset_opacity, set_obj_near_idx (adgs.densify / adgs.knn: HIP),
save_ply / load_ply (adgs.io: the reference's point_cloud.ply + deform.pth).
"""
import torch

from . import deform
from . import densify as _densify
from . import io as _io
from . import knn as _knn

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
        d = self.__dict__
        for n in _RAW:
            p = d.get(n)
            if p is not None:
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

    @property
    def obj_mask_float(self):
        """`get_obj_mask.float()[..., None]` (gaussian_renderer/__init__.py:111 of the reference builds it in every render: two fills, a
        cat and a cast, ~45 us of launches at 1 M Gaussians) -- a constant between two densifications: cached per (Ns, No, device)."""
        key = (int(self.get_scene_pts_num), int(self.get_obj_pts_num), str(self._scene_xyz.device))
        c = getattr(self, "_obj_mask_float", None)
        if c is None or c[0] != key:
            c = self._obj_mask_float = (key, self.get_obj_mask.float()[..., None].contiguous())
        return c[1]

    def get_deformed_xyz(self, t):
        return deform.get_deformed_xyz(self, t)

    raw_sh = False     # True: get_deformed_pkg leaves the SH coefficients in raw layout (RawSH) for forward_rawsh
    raw_scene = False  # True (with raw_sh): the scene range is not deformed/activated at all -- the rasterizer's preprocess reads the raw
    #                    scene tensors (adgs.deform.get_deformed_pkg(raw_scene=True)); rows [0, Ns) of 'xyz' / 'rotation' / 'opacity' /
    #                    'scales' / 'flow_xyz' are then uninitialised (gaussian_renderer.render() hides them behind lazy entries)

    supports_fused_flow = True     # get_deformed_pkg(t, flow_time=...) also returns 'flow_xyz'

    def get_deformed_pkg(self, t, flow_time=None, full_rows=False):
        """Reference keys 'xyz','rotation','shs','opacity' plus 'scales' (so that render() needs no
        separate get_scaling pass) and, with flow_time, 'flow_xyz' = get_deformed_xyz(flow_time).
        full_rows=True: every row of every tensor is valid even when raw_scene is on (callers that read them, e.g. override_color)."""
        return deform.get_deformed_pkg(self, t, raw_sh=self.raw_sh, flow_time=flow_time, raw_scene=self.raw_sh and self.raw_scene and not full_rows)

    def deform_bytes_per_frame(self):
        """Algorithmic bytes of the deformation stage per frame: deformation
        parameters read forward + their gradients written backward."""
        n = sum(getattr(self, k).numel() for k in ("xyz_deform_param", "rotation_deform_param", "shs_deform_param_scene",
                                                   "shs_deform_param_obj", "background_deform_param"))
        return 2 * 4 * n

    # ---- training-side methods under their reference names (scene/gaussian_model.py:338-400, 413-541, 825-867) ----
    percent_dense, scene_extent, object_extent = 0.01, 1.0, 1.0
    use_near_idx, near_num, obj_near_idx, optimizer = False, 0, None, None

    def training_setup(self, lrs=None, percent_dense=0.01, scene_extent=None, object_extent=None, near_num=0, adam_in_backward=False):
        """The optimizer of GaussianModel.training_setup (:338-372): one group per raw tensor, the reference's group names,
        Adam(lr=0, eps=1e-15) -- here the fused HIP Adam.  `lrs`: {group name: lr} (default 1e-3 each).
        adam_in_backward: FusedAdam(in_backward=True) -- optimizer.arm_backward() then lets the rasterizer's backward apply the step
        of the SH rest / SH deformation tensors itself (adgs.optim.BackwardEpilogue)."""
        from .optim import FusedAdam
        lrs = lrs or {}
        self.percent_dense = percent_dense
        self.scene_extent = self.scene_extent if scene_extent is None else scene_extent
        self.object_extent = self.object_extent if object_extent is None else object_extent
        dev = self._scene_xyz.device
        N = self.get_pts_num
        self.xyz_gradient_accum = torch.zeros((N, 1), device=dev)
        self.denom = torch.zeros((N, 1), device=dev)
        self.max_radii2D = torch.zeros((N,), device=dev)
        order = ["scene_xyz", "scene_shs_dc", "scene_shs_rest", "scene_opacity", "scene_scaling", "scene_rotation", "obj_xyz", "obj_shs_dc",
                 "obj_shs_rest", "obj_opacity", "obj_scaling", "obj_rotation", "deform_rotation", "deform_shs_scene", "deform_shs_obj", "deform_xyz",
                 "deform_background", "time_sigma"]
        groups = []
        for name in order:
            attr = _densify.GROUP_ATTR[name]
            t = getattr(self, attr)
            if not isinstance(t, torch.nn.Parameter):
                t = torch.nn.Parameter(t.detach().requires_grad_(True))
                setattr(self, attr, t)
            groups.append({"params": [t], "lr": float(lrs.get(name, 1e-3)), "name": name})
        self.optimizer = FusedAdam(groups, lr=0.0, eps=1e-15, in_backward=bool(adam_in_backward))
        self.near_num, self.use_near_idx = near_num, near_num > 0
        self.set_obj_near_idx()
        return self.optimizer

    def add_densification_stats(self, render_pkg):
        from .optim import add_densification_stats
        add_densification_stats(self.xyz_gradient_accum, self.denom, self.max_radii2D, render_pkg["viewspace_points"].grad, render_pkg["radii"])

    def densify_and_prune(self, max_scene_grad, max_obj_grad, min_opacity, prune_big_points):
        info = _densify.densify_and_prune(self, max_scene_grad, max_obj_grad, min_opacity, prune_big_points)
        self.set_obj_near_idx()
        return info

    def reset_opacity(self):
        _densify.reset_opacity(self)

    def set_obj_near_idx(self, K=None):
        return _knn.set_obj_near_idx(self, K)

    def save_ply(self, path):
        _io.save_ply(self, path)

    def load_ply(self, path):
        _io.load_ply(self, path, device=self._scene_xyz.device if torch.is_tensor(getattr(self, "_scene_xyz", None)) else "cuda")
