"""Fused Adam step on the HIP path.

Drop-in for the `torch.optim.Adam(l, lr=0.0, eps=1e-15)` the reference builds in
GaussianModel.training_setup (scene/gaussian_model.py:346-372) and steps at train.py:163-167:
same constructor arguments for what the reference uses, same `param_groups` (so the learning-rate
schedulers that write `group['lr']` keep working) and the same per-parameter state keys
(`step`, `exp_avg`, `exp_avg_sq`), which the densification code reads and rewrites
(scene/gaussian_model.py:413-459, 545-600: cat / prune of the optimizer state).

All groups are updated by ONE kernel launch (adgs_adam_step, include/adgs_optim.h); parameters
without a gradient are skipped exactly like torch does.  There is no CPU fallback.
"""
import ctypes

import torch

from . import _lib

MAX_GROUPS = 32


class AdamGroup(ctypes.Structure):
    _fields_ = [("param", ctypes.c_void_p), ("grad", ctypes.c_void_p), ("exp_avg", ctypes.c_void_p), ("exp_avg_sq", ctypes.c_void_p),
                ("numel", ctypes.c_int64), ("lr", ctypes.c_float), ("step", ctypes.c_int32), ("tile_active", ctypes.c_void_p),
                ("flags", ctypes.c_int32), ("reserved", ctypes.c_int32)]


class AdamSlot(ctypes.Structure):
    """adgs_adam_slot (include/adgs_optim.h)."""
    _fields_ = [("param", ctypes.c_void_p), ("exp_avg", ctypes.c_void_p), ("exp_avg_sq", ctypes.c_void_p), ("lr", ctypes.c_float), ("step", ctypes.c_int32)]


class ShAdam(ctypes.Structure):
    """adgs_sh_adam (include/adgs_optim.h)."""
    _fields_ = [(n, AdamSlot) for n in ("scene_rest", "obj_rest", "scene_deform", "obj_deform")] + \
               [("beta1", ctypes.c_float), ("beta2", ctypes.c_float), ("eps", ctypes.c_float), ("reserved", ctypes.c_int32)]


ADAM_TILE = 256            # ADGS_ADAM_TILE
TILES_MARKED, ZERO_GRAD = 1, 2          # adgs_adam_group.flags


class MarkedGradient:
    """What a gradient producer that marks the tiles it writes (adgs_envmap_backward_marked) shares with the optimizer of that
    parameter (FusedAdam.marked_gradient): the tile byte map, and a gradient buffer that is known to be all zero outside marked
    tiles.  Protocol: the producer calls take() for its output buffer (None: allocate zeros yourself) and issued(t) with the tensor
    it returns to autograd; the optimizer treats the tiles as marked only while p.grad is that buffer UNTOUCHED, zero-fills the
    tiles it updates when the step drops the gradient (step(zero_grad=True)), and puts the buffer back.

    "That buffer untouched" is decided by is_live(g): same storage object, same address, and the version counter the buffer had
    when it was issued.  An address alone proves nothing: with grad mode off autograd accumulates IN PLACE -- AccumulateGrad does
    `p.grad += new` for a second backward, the engine's input buffer may add a second loss term of the same backward into the
    first-arriving tensor -- and a non-marking source (image_background(), a regulariser on the map) would then put values into
    unmarked tiles of a buffer whose address still matches; every in-place write bumps the version counter that p.grad shares with
    the issued tensor (autograd installs `new_grad.detach()`).  The storage is held (not the tensor: a second reference to the
    tensor would make AccumulateGrad clone it instead of installing it) so that its address cannot be handed to another tensor
    while the record is live.  Anything else is a foreign gradient: dense scan, no recycling."""

    def __init__(self, marks):
        self.marks, self.buffer = marks, None
        self._live = None           # (storage, data_ptr, version at issue)

    def take(self):
        b, self.buffer = self.buffer, None
        self._live = None           # an issued gradient nobody stepped on is dropped here (its marks stay set: harmless, only cost)
        return b

    def issued(self, t):
        self._live = (t.untyped_storage(), t.data_ptr(), t._version)

    def is_live(self, g):
        if self._live is None:
            return False
        st, ptr, ver = self._live
        return g.data_ptr() == ptr and g._version == ver and g.untyped_storage()._cdata == st._cdata

    def retire(self):
        self._live = None


class BackwardClaim:
    """What one rasterizer backward was granted by BackwardEpilogue.claim: the adgs_sh_adam block, which tensors it covers (`fused`:
    name -> True) and the tensors that must stay alive until the kernels have been enqueued."""

    def __init__(self, struct, fused, keep, undo=None):
        self.struct, self.fused, self.keep, self._undo = struct, fused, keep, undo

    def rollback(self):
        """The backward the claim was made for did not run (the native call raised before any kernel was enqueued): the step counters go
        back, the tensors are not "already stepped" any more, and the optimizer is armed again as it was."""
        if self._undo is not None:
            self._undo()
            self._undo = None


class BackwardEpilogue:
    """The optimizer's side of the in-backward Adam step (FusedAdam(in_backward=True); include/adgs_optim.h: adgs_sh_adam).

    The reference's iteration is one backward followed by optimizer.step() (train.py:116, 163-167).  The kernels that produce the
    gradients of the SH `rest` tensors and of the SH deformation rows hold every element of them exactly once, and nothing else in
    AD-GS's loss reaches those tensors (the regularisers of train.py:101-110 act on xyz_deform_param and gs_time_sigma): armed for
    ONE backward (FusedAdam.arm_backward()), that backward applies the step to them where the gradient is produced -- 24 instead
    of 4 + 28 bytes per parameter -- and leaves their .grad None, so the step() that follows skips them exactly as torch.optim.Adam
    skips a parameter without a gradient.  Their `step` counters are advanced at the claim.

    Not for iterations that accumulate several backwards into one step (bench.py --config C4 / C5, data-parallel training): a second
    claim before step() raises, and so does a step() that finds a gradient on a tensor this backward has already stepped."""

    NAMES = ("scene_rest", "obj_rest", "scene_deform", "obj_deform")

    def __init__(self, opt):
        self.opt, self.armed, self.claimed = opt, False, []

    def claim(self, tensors, needs, factored=False):
        """Called by the rasterizer's backward.  tensors / needs: name -> tensor / needs_input_grad for NAMES.  Returns a BackwardClaim or
        None (not armed)."""
        if not self.armed:
            return None
        if factored:
            raise RuntimeError("FusedAdam.arm_backward(): the factored SH-gradient exchange (data-parallel training) sums the gradients of "
                               "several cameras before the step; the in-backward step is for the single-camera iteration")
        if self.claimed:
            raise RuntimeError("FusedAdam.arm_backward(): a second rasterizer backward before step() -- the in-backward Adam step is for "
                               "iterations of ONE backward (train.py:116, 163-167)")
        if torch.cuda.is_current_stream_capturing():
            raise RuntimeError("FusedAdam.arm_backward(): the in-backward Adam step cannot be captured into a graph")
        self.armed = False
        opt = self.opt
        where = {id(p): g for g in opt.param_groups for p in g["params"]}
        st_block, fused, keep, betas_eps = ShAdam(), {}, [], None
        pairs = (("scene_rest", "obj_rest"), ("scene_deform", "obj_deform"))
        for pair in pairs:
            # the two halves of a pair take the step together or not at all (adgs_raster_backward_rawsh)
            cand = [(n, tensors.get(n)) for n in pair]
            cand = [(n, t) for n, t in cand if t is not None and t.numel() != 0]
            ok = bool(cand) and all(needs.get(n) and id(t) in where and t.is_cuda and t.dtype == torch.float32 and t.is_contiguous() and t.grad is None
                                    for n, t in cand)
            if ok:
                keys = {(tuple(where[id(t)]["betas"]), float(where[id(t)]["eps"])) for _, t in cand}
                ok = len(keys) == 1 and (betas_eps is None or keys == {betas_eps})
                if ok:
                    betas_eps = next(iter(keys))
            if not ok:
                continue
            for n, t in cand:
                group = where[id(t)]
                st = opt._init_state(t)
                st["step"] += 1
                if not (st["exp_avg"].is_contiguous() and st["exp_avg_sq"].is_contiguous()):
                    st["exp_avg"], st["exp_avg_sq"] = st["exp_avg"].contiguous(), st["exp_avg_sq"].contiguous()
                setattr(st_block, n, AdamSlot(t.data_ptr(), st["exp_avg"].data_ptr(), st["exp_avg_sq"].data_ptr(), float(group["lr"]), int(st["step"])))
                fused[n] = True
                keep += [t, st["exp_avg"], st["exp_avg_sq"]]
                self.claimed.append(t)
        if not fused:
            return None
        (b1, b2), eps = betas_eps
        st_block.beta1, st_block.beta2, st_block.eps = float(b1), float(b2), float(eps)
        stepped = list(self.claimed)

        def undo():
            for t in stepped:
                opt.state[t]["step"] -= 1
            self.claimed = [t for t in self.claimed if not any(t is u for u in stepped)]
            self.armed = True
        return BackwardClaim(st_block, fused, keep, undo)


class FusedAdam(torch.optim.Optimizer):
    def __init__(self, params, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=0, amsgrad=False, skip_dormant_tiles=False, in_backward=False):
        """skip_dormant_tiles: keep, per parameter, one byte per 256 elements saying whether that tile has ever seen a non-zero
        gradient; a tile that has not is left untouched by the step -- which is exactly what Adam does to it (zero moments, zero
        update) at 4 instead of 28 bytes of traffic per element.  For parameters most of which never receive a gradient (the
        environment map).  Only valid while nothing but this optimizer writes the moments: moments that arrive from elsewhere
        (a loaded or edited state) are detected by their address and size and treated as active everywhere."""
        self.skip_dormant_tiles = bool(skip_dormant_tiles)
        if skip_dormant_tiles and in_backward:
            # the in-backward kernels update moments without marking tiles: a later ordinary step() would take every tile of such a tensor
            # for dormant and skip tiles whose moments are not zero (round-5 advisor finding)
            raise ValueError("FusedAdam: skip_dormant_tiles and in_backward cannot be combined (the in-backward step does not keep the tile map)")
        if weight_decay != 0 or amsgrad:
            raise ValueError("FusedAdam implements the configuration the reference uses: no weight decay, no amsgrad")
        if not 0.0 <= lr or not 0.0 <= eps or not (0.0 <= betas[0] < 1.0 and 0.0 <= betas[1] < 1.0):
            raise ValueError("invalid Adam hyper-parameter")
        super().__init__(params, dict(lr=lr, betas=betas, eps=eps))
        self._tile_maps = {}        # id(param) -> (byte map, exp_avg address, exp_avg_sq address, numel)
        self._marked = {}           # id(param) -> MarkedGradient
        # in_backward=True: the rasterizer's backward may apply the step to the tensors only it produces gradients for (BackwardEpilogue);
        # opt-in per backward with arm_backward()
        self.backward_epilogue = BackwardEpilogue(self) if in_backward else None

    def arm_backward(self):
        """The NEXT rasterizer backward over this optimizer's SH tensors applies their Adam step itself (BackwardEpilogue); call it in an
        iteration whose step() follows its one backward unconditionally -- not before a densification, after which the reference's
        step finds no gradients and changes nothing (train.py:152-167)."""
        if self.backward_epilogue is None:
            raise RuntimeError("arm_backward() needs FusedAdam(in_backward=True)")
        if self.backward_epilogue.claimed:
            raise RuntimeError("arm_backward(): the previous armed backward has not been followed by step()")
        self.backward_epilogue.armed = True

    def _init_state(self, p):
        st = self.state[p]
        if len(st) == 0:
            st["step"] = torch.tensor(0.0)          # same state layout as torch.optim.Adam
            st["exp_avg"] = torch.zeros_like(p, memory_format=torch.preserve_format)
            st["exp_avg_sq"] = torch.zeros_like(p, memory_format=torch.preserve_format)
            if self.skip_dormant_tiles:
                self._tile_maps[id(p)] = (torch.zeros((p.numel() + ADAM_TILE - 1) // ADAM_TILE, dtype=torch.uint8, device=p.device),
                                          st["exp_avg"].data_ptr(), st["exp_avg_sq"].data_ptr(), p.numel())
        return st

    def marked_gradient(self, p):
        """The MarkedGradient of parameter p (needs skip_dormant_tiles): hand it to the one producer of p's gradient.  Other sources
        of gradient for p are allowed -- whether autograd sums them into a new tensor or in place into the producer's buffer, step()
        sees that the buffer is no longer the one that was issued (MarkedGradient.is_live) and scans it densely."""
        if not self.skip_dormant_tiles:
            raise ValueError("marked_gradient needs skip_dormant_tiles=True")
        mg = self._marked.get(id(p))
        if mg is None:
            self._init_state(p)
            mg = self._marked[id(p)] = MarkedGradient(self._tile_maps[id(p)][0])
        return mg

    @torch.no_grad()
    def step(self, closure=None, zero_grad=False):
        """One Adam step over every parameter that has a gradient.
        zero_grad=True: the reference's `optimizer.zero_grad(set_to_none=True)` right after the step (train.py:165) -- the
        gradients are dropped (`p.grad = None`), so a parameter that receives no gradient in a later iteration is skipped by the
        next step exactly as torch.optim.Adam skips it (no moment-driven update, `step` not incremented).
        zero_grad="zeros": `zero_grad(set_to_none=False)` -- the gradients stay installed and are zero-filled by the same kernel
        pass (for callers that accumulate into persistent gradient buffers)."""
        if zero_grad not in (False, True, "zeros"):
            raise ValueError("zero_grad must be False, True (set to None) or 'zeros' (zero-fill in place)")
        fill = zero_grad == "zeros"
        loss = None
        if closure is not None:
            with torch.enable_grad():
                loss = closure()
        if self.backward_epilogue is not None:
            epi = self.backward_epilogue
            for p in epi.claimed:
                if p.grad is not None:
                    raise RuntimeError("FusedAdam.step(): a tensor whose Adam step was applied inside the rasterizer's backward has received a "
                                       "gradient from elsewhere; stepping it again would apply this iteration twice")
            epi.claimed, epi.armed = [], False
        batches = {}       # (device, betas, eps) -> list of AdamGroup
        keep = []
        recycle = []       # (MarkedGradient, its gradient buffer): zero again after this step
        for group in self.param_groups:
            b1, b2 = group["betas"]
            for p in group["params"]:
                if p.grad is None:
                    continue
                if not p.is_cuda or p.dtype != torch.float32:
                    raise RuntimeError("FusedAdam: parameters must be fp32 tensors on a HIP device; there is no CPU path")
                if p.grad.is_sparse:
                    raise RuntimeError("FusedAdam does not support sparse gradients")
                if not p.is_contiguous():
                    raise RuntimeError("FusedAdam: parameters must be contiguous")
                g = p.grad if p.grad.is_contiguous() else p.grad.contiguous()
                st = self._init_state(p)
                st["step"] += 1
                if p.numel() == 0:
                    continue
                if not (st["exp_avg"].is_contiguous() and st["exp_avg_sq"].is_contiguous()):
                    st["exp_avg"], st["exp_avg_sq"] = st["exp_avg"].contiguous(), st["exp_avg_sq"].contiguous()
                tile_map = None
                if self.skip_dormant_tiles:
                    ent = self._tile_maps.get(id(p))
                    if ent is None or ent[1:] != (st["exp_avg"].data_ptr(), st["exp_avg_sq"].data_ptr(), p.numel()):
                        # moments this optimizer did not create as zeros (loaded / edited state): every tile counts as active
                        ent = (torch.ones((p.numel() + ADAM_TILE - 1) // ADAM_TILE, dtype=torch.uint8, device=p.device),
                               st["exp_avg"].data_ptr(), st["exp_avg_sq"].data_ptr(), p.numel())
                        self._tile_maps[id(p)] = ent
                    tile_map = ent[0]
                flags = 0
                mg = self._marked.get(id(p))
                if mg is not None:
                    if tile_map is mg.marks and g is p.grad and mg.is_live(g):
                        # the producer's own buffer: its marks say which tiles hold anything.  Dropping the gradient with the step =
                        # zero the updated tiles in the same pass; the buffer is all zero again and goes back to the producer.
                        flags = TILES_MARKED | (ZERO_GRAD if zero_grad is True else 0)
                        if zero_grad is True:
                            recycle.append((mg, g))
                    elif tile_map is not mg.marks:
                        # the tile map was replaced (foreign moments): the producer keeps marking the old one, which nobody reads
                        self._marked.pop(id(p))
                    mg.retire()
                ag = AdamGroup(p.data_ptr(), g.data_ptr(), st["exp_avg"].data_ptr(), st["exp_avg_sq"].data_ptr(), p.numel(), float(group["lr"]),
                               int(st["step"]), tile_map.data_ptr() if tile_map is not None else None, flags, 0)
                keep.append(g)
                batches.setdefault((p.device, float(b1), float(b2), float(group["eps"])), []).append((ag, p, g))
        lib = _lib.lib()
        for (dev, b1, b2, eps), items in batches.items():
            with torch.cuda.device(dev):
                stream = _lib.stream_ptr(dev)
                for i in range(0, len(items), MAX_GROUPS):
                    chunk = items[i:i + MAX_GROUPS]
                    arr = (AdamGroup * len(chunk))(*[c[0] for c in chunk])
                    _lib.check(lib.adgs_adam_step(arr, len(chunk), b1, b2, eps, int(fill), stream), "adgs_adam_step")
            for _, p, g in items:
                if fill and g is not p.grad:
                    p.grad.zero_()          # the kernel zeroed the contiguous copy
                elif zero_grad is True:
                    p.grad = None
        for mg, g in recycle:
            mg.buffer = g
        return loss


def add_densification_stats(xyz_gradient_accum, denom, max_radii2D, viewspace_grad, radii):
    """train.py:148-150 + GaussianModel.add_densification_stats (scene/gaussian_model.py:863-867) in one kernel, in place:
    for every Gaussian with radii > 0: max_radii2D = max(max_radii2D, radii); xyz_gradient_accum += ||viewspace_grad[:, :2]||;
    denom += 1.  No boolean-mask indexing, no host synchronisation.  `max_radii2D` may be None."""
    for t in (xyz_gradient_accum, denom, viewspace_grad, radii):
        if not t.is_cuda or not t.is_contiguous():
            raise RuntimeError("add_densification_stats: contiguous tensors on a HIP device are required; there is no CPU path")
    if radii.dtype != torch.int32 or viewspace_grad.dtype != torch.float32 or viewspace_grad.shape[-1] != 3:
        raise ValueError("add_densification_stats: radii must be int32 [N], viewspace_grad fp32 [N,3]")
    N = radii.numel()
    if xyz_gradient_accum.numel() != N or denom.numel() != N or (max_radii2D is not None and max_radii2D.numel() != N):
        raise ValueError("add_densification_stats: accumulator sizes do not match the number of Gaussians")
    with torch.cuda.device(radii.device):
        _lib.check(_lib.lib().adgs_densification_stats(N, radii.data_ptr(), viewspace_grad.data_ptr(), xyz_gradient_accum.data_ptr(), denom.data_ptr(),
                                                       max_radii2D.data_ptr() if max_radii2D is not None else None,
                                                       _lib.stream_ptr(radii.device)), "adgs_densification_stats")
