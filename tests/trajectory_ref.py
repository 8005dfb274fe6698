"""CPU reference TRAJECTORY of the AD-GS training iteration (test infrastructure) -- train.py:47-167 composed from the oracles:

    camera -> tests/chain_ref.run_chain (deform_oracle -> raster_oracle -> env_oracle composite)                  train.py:73
           -> oracle/loss_oracle: L1 + SSIM, depth, flow, object BCE, sky BCE on the chain's own images            :78-103
           -> the three regularisers over obj_near_idx                                                            :104-113
           -> weighted total (:112-115); its gradient = chain rule through the oracles' hand-derived backward passes (:116)
           -> densification statistics (:148-150), densify_and_prune / set_obj_near_idx / reset_opacity (:152-158, densify_oracle,
              knn_points_oracle)
           -> Adam(lr per group, eps 1e-15) in float64 (:163-167; pinned against torch.optim.Adam in tests/test_oracle_trajectory.py)

Parameters are held as float32 (what the optimizer of the reference holds), moments and the update arithmetic as float64.
The random draws of the reference (torch.normal in densify_and_split, torch.randperm in set_obj_near_idx) are INPUTS here: the GPU test
feeds both sides the same standard-normal / permutation draws.
"""
import numpy as np

from oracle import densify_oracle as dz
from oracle import env_oracle, knn_points_oracle, loss_oracle as lo
from tests import chain_ref

# optimizer group name (scene/gaussian_model.py:346-372) -> raw attribute name of tests/chain_ref.RAW_NAMES
GROUP_RAW = {
    "scene_xyz": "scene_xyz", "scene_shs_dc": "scene_shs_dc", "scene_shs_rest": "scene_shs_rest", "scene_opacity": "scene_opacity",
    "scene_scaling": "scene_scaling", "scene_rotation": "scene_rotation", "deform_shs_scene": "shs_deform_param_scene",
    "obj_xyz": "obj_xyz", "obj_shs_dc": "obj_shs_dc", "obj_shs_rest": "obj_shs_rest", "obj_opacity": "obj_opacity",
    "obj_scaling": "obj_scaling", "obj_rotation": "obj_rotation", "deform_xyz": "xyz_deform_param", "deform_rotation": "rotation_deform_param",
    "deform_shs_obj": "shs_deform_param_obj", "time_sigma": "gs_time_sigma", "deform_background": "background_deform_param"}
BETA1, BETA2, EPS = 0.9, 0.999, 1e-15


def adam_step(p, g, m, v, step, lr, beta1=BETA1, beta2=BETA2, eps=EPS):
    """One torch.optim.Adam step (no weight decay, no amsgrad) in float64; returns (new float32 parameter, m, v).
    torch/optim/adam.py _single_tensor_adam: exp_avg.lerp_(grad, 1 - beta1); exp_avg_sq.mul_(beta2).addcmul_(grad, grad, 1 - beta2);
    denom = exp_avg_sq.sqrt() / sqrt(1 - beta2^t) + eps; param -= lr / (1 - beta1^t) * exp_avg / denom."""
    g = np.asarray(g, np.float64)
    m = beta1 * m + (1.0 - beta1) * g
    v = beta2 * v + (1.0 - beta2) * g * g
    denom = np.sqrt(v) / np.sqrt(1.0 - beta2 ** step) + eps
    new = np.asarray(p, np.float64) - (lr / (1.0 - beta1 ** step)) * m / denom
    return new.astype(np.float32), m, v


class RefTrainer:
    """State in oracle/densify_oracle's layout: st["p"|"m"|"v"][group], st["gs_time"], st["xyz_gradient_accum"|"denom"|"max_radii2D"]."""

    def __init__(self, raw, lrs, order_args, use_time_mask, weights, sh_degree, scene_extent, object_extent, percent_dense, frame_gap, near_num,
                 env_grid=None, env_lr=0.0):
        self.st = dict(p={g: np.array(raw[r], np.float32) for g, r in GROUP_RAW.items()}, m={}, v={}, gs_time=np.array(raw["gs_time"], np.float32))
        for g, x in self.st["p"].items():
            self.st["m"][g] = np.zeros(x.shape, np.float64)
            self.st["v"][g] = np.zeros(x.shape, np.float64)
        N = self.n_scene + self.n_obj
        self.st.update(xyz_gradient_accum=np.zeros((N, 1), np.float32), denom=np.zeros((N, 1), np.float32), max_radii2D=np.zeros((N,), np.float32))
        self.steps = {g: 0 for g in GROUP_RAW}
        self.lrs, self.oa, self.use_time_mask, self.w, self.degree = dict(lrs), order_args, use_time_mask, dict(weights), sh_degree
        self.scene_extent, self.object_extent, self.percent_dense = scene_extent, object_extent, percent_dense
        self.frame_gap, self.near_num = frame_gap, near_num
        self.near_idx = None
        self.env = None if env_grid is None else dict(p=np.array(env_grid, np.float32), m=np.zeros(np.shape(env_grid), np.float64),
                                                      v=np.zeros(np.shape(env_grid), np.float64), step=0, lr=env_lr)
        self.precision = "f32"      # of the rasterizer oracle ("f64": finite-difference checks of the composed gradient)
        self.fresh = set()          # groups whose parameter tensor was replaced after the backward of this iteration: no gradient -> no step

    n_scene = property(lambda self: self.st["p"]["scene_xyz"].shape[0])
    n_obj = property(lambda self: self.st["p"]["obj_xyz"].shape[0])

    def raw(self):
        r = {rn: self.st["p"][g] for g, rn in GROUP_RAW.items()}
        r["gs_time"] = self.st["gs_time"]
        return r

    # ---- scene/gaussian_model.py:825-833
    def set_obj_near_idx(self, perm):
        """perm: the permutation torch.randperm(n_obj) drew on the other side."""
        K = self.near_num
        xyz = self.st["p"]["obj_xyz"]
        if self.use_time_mask:
            xyz = np.concatenate([xyz, (self.st["gs_time"] * np.float32(self.scene_extent)).astype(np.float32)], -1)
        anchors = xyz[np.asarray(perm)[:xyz.shape[0] // K]]
        self.near_idx = knn_points_oracle.knn_points(anchors, xyz, K)[1]
        return self.near_idx

    # ---- train.py:73-116
    def loss_and_grads(self, cam, t, flow_pkg, targets, env_cam=None):
        """cam: make_camera()-style dict of numpy arrays; flow_pkg = (t_flow, K, R, T, flow [2,H,W], vis [H,W]);
        targets: dict(image [3,H,W], depth [H,W], semantic [H,W], sky [H,W]); env_cam: dict(focal, R) when an environment map is trained.
        Returns dict(total, terms {name: value}, grads {group: float64 array or None}, env_grad, chain (run_chain's result))."""
        w = self.w
        H, W = targets["image"].shape[-2:]
        raw = self.raw()
        terms = {}
        t_flow, K, R, T, flow, vis = flow_pkg

        def ups(img):
            render = img["render"] if self.env is not None else np.asarray(img["color"], np.float64)
            l1, ssim, g_l1, g_ssim = lo.l1_ssim(render, targets["image"])
            dl, g_d = lo.depth_loss(np.asarray(img["depth"], np.float64)[0], targets["depth"])
            op = np.asarray(img["img_opacity"], np.float64)[0]
            fl, g_f, g_fo = lo.flow_loss(img["img_flow"], flow, vis, op, K, R, T, self.scene_extent * 1e-3)
            ol, g_sem = lo.bce_clip_loss(np.asarray(img["img_semantic"], np.float64)[0], (np.asarray(targets["semantic"]) > 0).astype(np.float64))
            sl, g_sky = lo.bce_clip_loss(op, targets["sky"], invert=True)
            terms.update(l1=float(l1), dssim=float(1.0 - ssim), depth=float(dl), flow=float(fl), obj=float(ol), sky=float(sl))
            return {"render" if self.env is not None else "color": w["l1"] * g_l1 - w["dssim"] * g_ssim, "depth": (w["depth"] * g_d)[None],
                    "img_opacity": (w["flow"] * g_fo + w["sky"] * g_sky)[None], "flow": w["flow"] * g_f, "semantic": (w["obj"] * g_sem)[None]}

        env = None if self.env is None else dict(grid_map=self.env["p"][0], focal=env_cam["focal"], R=env_cam["R"])
        sem = np.concatenate([np.zeros(self.n_scene, np.float32), np.ones(self.n_obj, np.float32)])[:, None]
        chain = chain_ref.run_chain(raw, self.oa, self.use_time_mask, t, t_flow, cam, H, W, self.degree, ups, semantic=sem, env=env, precision=self.precision)
        grads = {g: (None if chain["raw_grads"][rn] is None else np.array(chain["raw_grads"][rn], np.float64)) for g, rn in GROUP_RAW.items()}
        # regularisers (train.py:104-113)
        reg, g_reg = lo.group_var_loss(raw["xyz_deform_param"], self.near_idx)
        sig, g_sig = lo.sigma_loss(raw["gs_time_sigma"], self.frame_gap)
        rsig, g_rsig = lo.group_var_loss(raw["gs_time_sigma"], self.near_idx)
        terms.update(reg=reg, sigma=sig, reg_sigma=rsig)
        add = lambda g, x: x if grads[g] is None else grads[g] + x
        grads["deform_xyz"] = add("deform_xyz", w["reg"] * g_reg)
        grads["time_sigma"] = add("time_sigma", w["sigma"] * g_sig + w["reg_sigma"] * g_rsig)
        total = sum(w[k] * terms[k] for k in ("l1", "dssim", "depth", "flow", "obj", "sky", "reg", "sigma", "reg_sigma"))
        return dict(total=float(total), terms=terms, grads=grads, env_grad=chain["env_grad"], chain=chain)

    # ---- train.py:146-150
    def add_densification_stats(self, chain):
        radii = np.asarray(chain["radii"])
        vis = radii > 0
        self.st["max_radii2D"][vis] = np.maximum(self.st["max_radii2D"][vis], radii[vis].astype(np.float32))
        dz.add_densification_stats(self.st, chain["act_grads"]["dL_dmeans2D"], vis)

    def densify_and_prune(self, max_scene_grad, max_obj_grad, min_opacity, prune_big, normal_scene, normal_obj):
        """normal_*: the STANDARD normal draws [2 n_split, 3] of densify_and_split (:719-720); the std of each row is exp(scaling) of
        its parent, from THIS side's state."""
        a = dict(max_scene_grad=max_scene_grad, max_obj_grad=max_obj_grad, min_opacity=min_opacity, prune_big_points=prune_big,
                 percent_dense=self.percent_dense, scene_extent=self.scene_extent, object_extent=self.object_extent)
        sel = self.split_parents(max_scene_grad, max_obj_grad)
        samples = []
        for side, z in (("scene", normal_scene), ("obj", normal_obj)):
            n_side = self.st["p"][side + "_scaling"].shape[0]
            std = np.tile(np.exp(self.st["p"][side + "_scaling"][sel[side][:n_side]], dtype=np.float32), (2, 1))
            z = np.asarray(z, np.float32).reshape(-1, 3)
            assert z.shape == std.shape, "normal draws for %d split parents expected, got %s" % (int(sel[side].sum()), z.shape)
            samples.append(z * std)
        before = set(self.st["p"])
        dz.densify_and_prune(self.st, a, samples[0], samples[1])
        self.fresh = {g for g in before if g != "deform_background"}      # cat_tensors_to_optimizer / _prune_optimizer replace these tensors
        return sel

    def split_parents(self, max_scene_grad, max_obj_grad):
        """The rows densify_and_split will split (:715-731), {side: bool mask over the side's rows AFTER the clone step appended its rows}."""
        Ns = self.n_scene
        with np.errstate(divide="ignore", invalid="ignore"):
            grads = self.st["xyz_gradient_accum"].astype(np.float32) / self.st["denom"].astype(np.float32)
        grads[np.isnan(grads)] = 0.0
        grads = np.abs(grads[:, 0])
        out, n_clone = {}, {}
        for side, gsel, thr, ext in (("scene", grads[:Ns], max_scene_grad, self.scene_extent), ("obj", grads[Ns:], max_obj_grad, self.object_extent)):
            big = np.exp(self.st["p"][side + "_scaling"], dtype=np.float32).max(1) > np.float32(ext * self.percent_dense)
            hot = gsel >= np.float32(thr)
            n_clone[side] = int((hot & ~big).sum())
            out[side] = np.concatenate([hot & big, np.zeros(n_clone[side], bool)])
        self.last_counts = {s: (n_clone[s], int(out[s].sum())) for s in out}
        return out

    def reset_opacity(self):
        dz.reset_opacity(self.st)
        self.fresh |= {"scene_opacity", "obj_opacity"}

    # ---- train.py:163-167
    def optimizer_step(self, grads, env_grad=None):
        for g, x in grads.items():
            if x is None or g in self.fresh or self.st["p"][g].size == 0:
                continue
            self.steps[g] += 1
            self.st["p"][g], self.st["m"][g], self.st["v"][g] = adam_step(self.st["p"][g], x.reshape(self.st["p"][g].shape), self.st["m"][g],
                                                                         self.st["v"][g], self.steps[g], self.lrs[g])
        self.fresh = set()
        if self.env is not None and env_grad is not None:
            e = self.env
            e["step"] += 1
            p, e["m"], e["v"] = adam_step(e["p"][0], env_grad, e["m"][0], e["v"][0], e["step"], e["lr"])
            e["p"], e["m"], e["v"] = p[None], e["m"][None], e["v"][None]


def gap_threshold(values, quantile, rel_gap=1e-3):
    """A threshold near the `quantile` of the positive `values` that sits in the MIDDLE of a gap of at least `rel_gap` (relative) between
    two neighbouring values: no decision of `value >= threshold` is within rounding error of the threshold on either side of a parity
    comparison."""
    v = np.sort(np.asarray(values, np.float64)[np.asarray(values) > 0])
    if v.size < 2:
        return 1.0
    k = int(np.clip(round(quantile * (v.size - 1)), 0, v.size - 2))
    for d in range(v.size):
        for j in (k + d, k - d):
            if 0 <= j < v.size - 1 and v[j + 1] - v[j] > rel_gap * v[j + 1]:
                return float(0.5 * (v[j] + v[j + 1]))
    raise AssertionError("no gap")


# ---------------------------------------------------------------- a small dynamic case both sides start from
WEIGHTS = dict(l1=0.8, dssim=0.2, depth=0.1, flow=0.1, sky=0.05, obj=0.1, sigma=0.01, reg=0.5, reg_sigma=0.5)      # arguments/__init__.py:104-133 as train.py:112-115 combines them
LRS = {"scene_xyz": 1.6e-4, "obj_xyz": 1.6e-4, "scene_shs_dc": 2.5e-3, "obj_shs_dc": 2.5e-3, "scene_shs_rest": 1.25e-4, "obj_shs_rest": 1.25e-4,
       "scene_opacity": 0.05, "obj_opacity": 0.05, "scene_scaling": 5e-3, "obj_scaling": 5e-3, "scene_rotation": 1e-3, "obj_rotation": 1e-3,
       "deform_rotation": 1e-3, "deform_shs_scene": 1e-3, "deform_shs_obj": 1e-3, "deform_xyz": 1e-3, "deform_background": 1e-3, "time_sigma": 1e-3}


def build_case(P=6000, W=208, H=130, focal=150.0, n_objects=2, seed=9, n_cameras=2, model_seed=0):
    """(scene dict in the z-up world, CPU SyntheticGaussianModel, cameras): the scene, model and per-camera supervision of
    examples/train_iteration.build() for a small configuration, as CPU tensors / numpy arrays.  Camera k: dict(cam (numpy make_camera
    dict), time, flow_pkg (numpy), targets, env_cam)."""
    import sys, os
    import torch
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    for q in (root, os.path.join(root, "ad-gs_amd")):
        if q not in sys.path:
            sys.path.insert(0, q)
    from adgs import synthetic
    from adgs.model import SyntheticGaussianModel
    import bench
    cfg = dict(P=P, W=W, H=H, focal=focal, sh_degree=3, n_objects=n_objects, seed=seed)
    sc = synthetic.to_z_up_world(synthetic.make_scene(P, W, H, focal, sh_degree=3, seed=seed, n_objects=n_objects))
    model = SyntheticGaussianModel.from_scene(sc, device="cpu", seed=model_seed)
    g = torch.Generator().manual_seed(11)
    cams = []
    for cam, t in bench.camera_pool(cfg, n_cameras):
        cam = synthetic.camera_to_z_up(cam)
        image = torch.rand(3, H, W, generator=g)
        depth = torch.rand(H, W, generator=g) * 0.5 + 0.01
        semantic = (torch.rand(H, W, generator=g) > 0.8).float()
        sky = (torch.rand(H, W, generator=g) > 0.7).float()
        K = torch.tensor([[focal, 0.0, W / 2.0], [0.0, focal, H / 2.0], [0.0, 0.0, 1.0]])
        flow = torch.stack([torch.rand(H, W, generator=g) * (W - 1), torch.rand(H, W, generator=g) * (H - 1)])
        vis = (torch.rand(H, W, generator=g) > 0.3).float()
        R, T = cam["viewmatrix"][:3, :3].t().contiguous(), cam["viewmatrix"][3, :3].contiguous()
        cams.append(dict(cam_t=cam, cam={k: (v.numpy() if torch.is_tensor(v) else v) for k, v in cam.items()}, time=t,
                         flow_pkg_t=(t + 0.05, K, R, T, flow, vis), flow_pkg=(t + 0.05, K.numpy(), R.numpy(), T.numpy(), flow.numpy(), vis.numpy()),
                         targets_t=dict(image=image, depth=depth, semantic=semantic, sky=sky),
                         targets=dict(image=image.numpy(), depth=depth.numpy(), semantic=semantic.numpy(), sky=sky.numpy()),
                         env_cam=dict(focal=W / (2.0 * np.tan(cam["fovx"] / 2.0)), R=cam["viewmatrix"][:3, :3].numpy())))
    return cfg, sc, model, cams
