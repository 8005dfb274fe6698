"""Camera-parallel data parallelism for AD-GS training on one MI355X node.

The reference is single-GPU, one camera per iteration (train.py:55-61,74); there is no
collective anywhere in it.  Renders of different cameras share read-only Gaussian
parameters and are independent forward and backward, so the natural sharding is: one
process per GPU, full parameter replica per rank, the iteration's camera batch dealt
round-robin to the ranks, and ONE exchange step -- the sum of the per-camera parameter
gradients (plus the densification statistics) -- as an all-reduce over RCCL/xGMI
(SURVEY.md section 8(e)).  `torch.distributed` backend "nccl" is RCCL on ROCm; the CPU
tests use "gloo".  Two forms of the exchange: `allreduce_gradients` (every materialised gradient) and
`FactoredSHExchange` (the SH gradients travel as their [P,3] colour-gradient factor and are expanded locally:
290 MB instead of 777 MB of link traffic per rank at 8 GPUs).
"""
import ctypes

import torch
import torch.distributed as dist


def shard_cameras(cameras, rank, world_size):
    """Round-robin deal of this iteration's camera batch (C4: 3 cameras over 4 ranks -> rank 3 idles)."""
    return [c for i, c in enumerate(cameras) if i % world_size == rank]


def _flatten_bucket(params, max_bytes):
    """Greedy bucketing of parameters (by gradient size) into groups of at most max_bytes (one collective per group)."""
    buckets, cur, size = [], [], 0
    for t in params:
        nbytes = t.grad.numel() * t.grad.element_size()
        if cur and size + nbytes > max_bytes:
            buckets.append(cur); cur, size = [], 0
        cur.append(t); size += nbytes
    if cur:
        buckets.append(cur)
    return buckets


def allreduce_gradients(params, group=None, average=False, bucket_bytes=256 << 20, in_place_bytes=32 << 20, force=False):
    """Sum `.grad` of every parameter over all ranks (missing grads count as zero).

    Tensors of at least `in_place_bytes` are reduced in place, one async collective each (no flatten
    copy: at 1M Gaussians the SH gradient alone is 180 MB); the others are coalesced into flat buckets
    (one `cat`; afterwards every `.grad` of the bucket is re-pointed at its slice of the reduced flat buffer, so
    nothing is copied back).  xGMI is point-to-point, so few large collectives are preferred over many small ones:
    C3 ends up with 6 in-place reductions and one bucket.
    """
    allreduce_gradients_finish(allreduce_gradients_start(params, group, average, bucket_bytes, in_place_bytes, force))


def allreduce_gradients_start(params, group=None, average=False, bucket_bytes=256 << 20, in_place_bytes=32 << 20, force=False):
    """Issue the collectives of allreduce_gradients and return a handle for allreduce_gradients_finish: work enqueued in
    between runs concurrently with the reduction (it must not touch the gradients being reduced)."""
    if not dist.is_available() or not dist.is_initialized() or (dist.get_world_size(group) == 1 and not force):
        return None                              # force: issue the collectives even in a one-rank group (RCCL self-test on a 1-GPU box)
    world = dist.get_world_size(group)
    params = list(params)
    for p in params:
        if p.grad is None:
            p.grad = torch.zeros_like(p)
    big = [p for p in params if p.grad.numel() * p.grad.element_size() >= in_place_bytes]
    small = [p for p in params if p.grad.numel() * p.grad.element_size() < in_place_bytes]
    works = [dist.all_reduce(p.grad, op=dist.ReduceOp.SUM, group=group, async_op=True) for p in big]
    flats = []
    for bucket in _flatten_bucket(small, bucket_bytes):
        flat = torch.cat([p.grad.reshape(-1) for p in bucket])
        works.append(dist.all_reduce(flat, op=dist.ReduceOp.SUM, group=group, async_op=True))
        flats.append((flat, bucket))
    return dict(works=works, flats=flats, big=big, average=average, world=world)


def allreduce_gradients_finish(handle):
    if handle is None:
        return
    for w in handle["works"]:
        w.wait()
    for flat, bucket in handle["flats"]:
        if handle["average"]:
            flat.div_(handle["world"])
        # the reduced gradients ARE slices of the flat buffer from here on (no copy back: one launch per tensor saved)
        off = 0
        for p in bucket:
            n = p.grad.numel()
            p.grad = flat[off:off + n].view(p.grad.shape)
            off += n
    if handle["average"]:
        for p in handle["big"]:
            p.grad.div_(handle["world"])


def allreduce_densification_stats(xyz_gradient_accum, denom, max_radii2D, group=None):
    """Densification statistics must agree on every replica before densify_and_prune
    (scene/gaussian_model.py:863-867, train.py:151): sums for the accumulators, max for radii."""
    if not dist.is_available() or not dist.is_initialized() or dist.get_world_size(group) == 1:
        return
    dist.all_reduce(xyz_gradient_accum, op=dist.ReduceOp.SUM, group=group)
    dist.all_reduce(denom, op=dist.ReduceOp.SUM, group=group)
    dist.all_reduce(max_radii2D, op=dist.ReduceOp.MAX, group=group)


def seed_all_ranks(seed, group=None):
    """Densify draws torch.normal / randperm (scene/gaussian_model.py:720,731,832): every replica
    must draw the same samples, so rank 0's seed is broadcast and applied everywhere."""
    t = torch.tensor([int(seed)], dtype=torch.int64)
    if dist.is_available() and dist.is_initialized() and dist.get_world_size(group) > 1:
        dev = torch.device("cuda", torch.cuda.current_device()) if dist.get_backend(group) == "nccl" else torch.device("cpu")
        t = t.to(dev)
        dist.broadcast(t, src=0, group=group)
    torch.manual_seed(int(t.item()))
    return int(t.item())


def multi_camera_step(render_loss_fn, cameras, params, group=None):
    """One data-parallel iteration: every rank renders its share of `cameras` through
    `render_loss_fn(camera) -> scalar loss`, back-propagates, and the gradients are summed over
    ranks and divided by the number of cameras, so the result equals single-GPU gradient
    accumulation over the same cameras with loss = mean over cameras."""
    world = dist.get_world_size(group) if (dist.is_available() and dist.is_initialized()) else 1
    rank = dist.get_rank(group) if world > 1 else 0
    total = None
    for cam in shard_cameras(cameras, rank, world):
        loss = render_loss_fn(cam) / float(len(cameras))
        loss.backward()
        total = loss.detach() if total is None else total + loss.detach()
    allreduce_gradients(params, group=group)
    return total


# ------------------------------------------------------------------ factored SH-gradient exchange
_SH_PARAMS = ("_scene_shs_dc", "_obj_shs_dc", "_scene_shs_rest", "_obj_shs_rest", "shs_deform_param_scene", "shs_deform_param_obj")


def _dense_basis_weights(times, order_args_shs, C, device):
    """W[c][j] = d f_shs(t_c) / d param[..., j] (dense [n, C]); f_shs is linear in its parameters (utils/func_utils.py:121-156)."""
    from . import deform
    W = torch.zeros(len(times), max(C, 1), dtype=torch.float32)
    if C > 0:
        for c, t in enumerate(times):
            f = deform.make_func_eval(float(t), order_args_shs, C)
            for i in range(f.n_terms[0] + f.n_terms[1] + f.n_terms[2]):
                W[c, f.index[i]] = f.weight[i]
    return W.to(device)


class _ExpandCam(ctypes.Structure):
    """adgs_sh_expand_cam (include/adgs_exchange.h)."""
    _fields_ = [("rgb", ctypes.c_void_p), ("xyz_tail", ctypes.c_void_p), ("campos", ctypes.c_float * 3), ("reserved", ctypes.c_float)]


def _expand_grads():
    """adgs_sh_grads (include/adgs_rasterizer.h): ONE ctypes mirror for both users of the struct (the rasterizer backward and the
    expansion), so that its size cannot drift from the C side again; tests/test_abi_and_oracle_knn.py checks sizeof against the library."""
    from diff_gaussian_rasterization._C import ShGrads
    g = ShGrads()
    g.struct_bytes = ctypes.sizeof(ShGrads)
    return g


def hip_sh_grad_expand(cams, W, C, P, Ns, row0, xyz_head, D, M, outs, _cache=None):
    """adgs_sh_grad_expand (include/adgs_exchange.h): cams = [(rgb[P,3], xyz_tail[P-row0,3] | None, campos 3 floats)],
    outs = six tensors (or None) in _SH_PARAMS order, fully written.  HIP only -- there is no CPU path.
    _cache: a dict owned by the caller; the ctypes camera array is rebuilt only when a pointer or a position changes."""
    from . import _lib
    dev = cams[0][0].device
    if dev.type != "cuda":
        raise RuntimeError("adgs_sh_grad_expand needs HIP tensors; there is no CPU path")
    ptr = lambda t: None if (t is None or t.numel() == 0) else t.data_ptr()
    key = tuple((ptr(rgb), ptr(tail), tuple(float(x) for x in campos)) for rgb, tail, campos in cams)
    arr = _cache.get(key) if _cache is not None else None
    if arr is None:
        arr = (_ExpandCam * len(cams))()
        for c, (rgb, tail, campos) in enumerate(cams):
            if not (rgb.is_contiguous() and rgb.dtype == torch.float32 and (tail is None or (tail.is_contiguous() and tail.dtype == torch.float32))):
                raise RuntimeError("adgs_sh_grad_expand: factors and means must be contiguous float32")
            arr[c].rgb, arr[c].xyz_tail = key[c][0], key[c][1]
            for i in range(3):
                arr[c].campos[i] = key[c][2][i]
        if _cache is not None:
            _cache.clear(); _cache[key] = arr
    g = _expand_grads()
    for n, t in zip(("scene_dc", "obj_dc", "scene_rest", "obj_rest", "scene_deform", "obj_deform"), outs):
        setattr(g, n, ptr(t))
    with torch.cuda.device(dev):
        _lib.check(_lib.lib().adgs_sh_grad_expand(len(cams), arr, ptr(W), int(C), int(P), int(Ns), int(row0), ptr(xyz_head), int(D), int(M),
                                                  ctypes.byref(g), _lib.stream_ptr(dev)),
                   "adgs_sh_grad_expand")


ARENA_QUANTUM = 64 * 840


def reduce_flat(flat, group=None, shard_cache=None):
    """Sum ONE flat fp32 buffer over the ranks, asynchronously; returns the list of work handles to wait on.
    ADGS_DP_COLLECTIVE=all_reduce (default): one RCCL all-reduce.  ADGS_DP_COLLECTIVE=rs_ag: reduce-scatter + all-gather, each rank
    owning 1/world of the buffer -- on a fully connected xGMI node every rank then exchanges S/n with every peer directly in
    both phases whatever algorithm RCCL picks for all-reduce; which of the two is faster is for the first
    multi-GPU run to measure (bench.py prints exchange_ms for either).  gloo (host-side dry runs) has no reduce-scatter: there the
    rs_ag form is all-reduce + an all-gather of the owned shard, i.e. the same sequence of collectives on the same buffers."""
    import os
    mode = os.environ.get("ADGS_DP_COLLECTIVE", "all_reduce")
    world = dist.get_world_size(group)
    if mode != "rs_ag" or flat.numel() % world != 0 or world == 1:
        return [dist.all_reduce(flat, op=dist.ReduceOp.SUM, group=group, async_op=True)]
    n = flat.numel() // world
    rank = dist.get_rank(group)
    shard = None if shard_cache is None else shard_cache.get("shard")
    if shard is None or shard.numel() != n or shard.device != flat.device:
        shard = torch.empty(n, dtype=flat.dtype, device=flat.device)
        if shard_cache is not None:
            shard_cache["shard"] = shard
    if dist.get_backend(group) == "gloo":
        w = dist.all_reduce(flat, op=dist.ReduceOp.SUM, group=group, async_op=True)
        w.wait()
        shard.copy_(flat[rank * n:(rank + 1) * n])
        return [dist.all_gather_into_tensor(flat, shard, group=group, async_op=True)]
    w1 = dist.reduce_scatter_tensor(shard, flat, op=dist.ReduceOp.SUM, group=group, async_op=True)
    w2 = dist.all_gather_into_tensor(flat, shard, group=group, async_op=True)      # a communicator runs its collectives in issue order
    return [w1, w2]


class GradArena:
    """ONE persistent flat buffer for the dense parameter gradients of a model.  The deformation backward (adgs.deform) asks
    `take(name, like)` for the tensor it writes a parameter's gradient into and gets a slice of the buffer; autograd then
    installs that slice as `.grad` without a copy, and the data-parallel reduction is a single in-place all-reduce of the buffer
    (no `cat`, no copy back).  Each slice is handed out once per iteration (`reset()`): a second backward of the same iteration
    allocates normally and autograd accumulates into the slice."""

    def __init__(self, named_params):
        self.names = [n for n, p in named_params if p is not None and p.numel() > 0]
        self.shapes = {n: tuple(p.shape) for n, p in named_params if p is not None and p.numel() > 0}
        self.offsets, off = {}, 0
        for n in self.names:
            numel = 1
            for d in self.shapes[n]:
                numel *= d
            self.offsets[n] = (off, numel)
            off += (numel + 63) // 64 * 64                       # 256-byte aligned slices
        ref = next(p for n, p in named_params if p is not None and p.numel() > 0)
        # sized to a multiple of 64 * lcm(1..8) floats: every world size up to 8 splits it into equal, 256-byte aligned shards
        # (the reduce-scatter + all-gather form of the exchange, ADGS_DP_COLLECTIVE=rs_ag)
        self.flat = torch.zeros((max(off, 1) + ARENA_QUANTUM - 1) // ARENA_QUANTUM * ARENA_QUANTUM, dtype=torch.float32, device=ref.device)
        self.handed = set()
        self.xyz_sink = None                 # set by FactoredSHExchange: receives the upstream position gradients of the object range

    def matches(self, named_params):
        return all(self.shapes.get(n) == tuple(p.shape) for n, p in named_params if p is not None and p.numel() > 0) and \
            len(self.shapes) == sum(1 for n, p in named_params if p is not None and p.numel() > 0)

    def reset(self):
        self.handed.clear()

    def take(self, name, like):
        # A parameter that still has a gradient installed (accumulation over several backward passes, or gradients zero-filled in
        # place: optimizer.zero_grad(set_to_none=False) / FusedAdam.step(zero_grad="zeros")) must not get its slice: that `.grad`
        # may BE the slice, the kernels overwrite their destination, and autograd's AccumulateGrad would then add the slice to
        # itself (2 g).  The backward writes a fresh tensor instead and autograd accumulates it into the installed gradient.
        if name not in self.offsets or name in self.handed or tuple(like.shape) != self.shapes[name] or like.device != self.flat.device \
                or like.grad is not None:
            return None
        self.handed.add(name)
        off, numel = self.offsets[name]
        return self.flat[off:off + numel].view(self.shapes[name])

    def holds(self, name, grad):
        """True when `grad` is this arena's slice for `name` (autograd installed it without a copy)."""
        if grad is None or name not in self.offsets:
            return False
        off, numel = self.offsets[name]
        return grad.data_ptr() == self.flat.data_ptr() + 4 * off and grad.numel() == numel and grad.is_contiguous()


def hip_lin_grad_expand(terms, W, C, count, scale, out):
    """adgs_lin_grad_expand (include/adgs_exchange.h): out[m, d, j] = scale * sum_e W[e][j] * terms[e][m, d]; HIP only."""
    from . import _lib
    if not out.is_cuda:
        raise RuntimeError("adgs_lin_grad_expand needs HIP tensors; there is no CPU path")
    arr = (ctypes.c_void_p * len(terms))(*[t.data_ptr() for t in terms])
    for t in terms:
        if not (t.is_contiguous() and t.dtype == torch.float32 and t.numel() == 3 * count):
            raise RuntimeError("adgs_lin_grad_expand: factors must be contiguous float32 [count,3]")
    with torch.cuda.device(out.device):
        _lib.check(_lib.lib().adgs_lin_grad_expand(len(terms), arr, W.data_ptr(), int(C), int(count), float(scale), out.data_ptr(),
                                                   _lib.stream_ptr(out.device)), "adgs_lin_grad_expand")


class _FactorSink(list):
    """What forward_rawsh(factor_sink=...) receives: the backward asks `next_target(P)` for the [P,3] destination of its colour-gradient
    factor (a slice of the exchange's send buffer: no copy afterwards) and appends the tensor it wrote."""

    def __init__(self, owner):
        super().__init__()
        self.owner = owner

    def next_target(self, P):
        return self.owner._target(len(self), P)

    def append(self, factor):
        super().append(factor)
        self.owner._factor_added()


class FactoredSHExchange:
    """Gradient exchange of one camera-parallel iteration with the SH gradients in factored form.

    At C3/C4 the parameter gradients are 444 MB per rank, of which 336 B per Gaussian -- shs_dc, shs_rest and
    shs_deform_param -- are multiples of ONE 3-vector per Gaussian and camera (include/adgs_exchange.h).  Instead of
    materialising and all-reducing those rows, every rank
      1. renders its cameras with `forward_rawsh(..., factor_sink=ex.sink_for(pkg["xyz"]))`: the backward leaves the
         [P,3] factor in the exchange's send buffer and writes no SH gradient (336 MB less HBM traffic per camera at
         1 M Gaussians),
      2. calls `ex.reduce(cam_times, cam_positions)`: one all-gather of the factors (+ the time-dependent means), one
         all-reduce of the remaining (dense) gradients, then adgs_sh_grad_expand sums the SH gradients of ALL cameras
         into `.grad` of the six SH tensors -- in global camera order, so every rank gets the same bits.
    xGMI traffic per rank at 8 GPUs, 1 M Gaussians: 2*(7/8)*108 MB + 7*14.4 MB = 290 MB instead of 777 MB.
    Cameras are dealt round-robin (shard_cameras): local camera j of rank r is global camera j*world + r.
    The SH gradients are written, not accumulated into an existing `.grad`.  With world == 1 the same class serves
    single-GPU multi-camera gradient accumulation.  Buffers, basis weights and the ctypes camera table persist
    across iterations: the per-iteration host work is a handful of launches.
    """

    def __init__(self, model, group=None, expand=None, factor_xyz=False):
        self.model, self.group = model, group
        self.expand = expand
        self.want_factor_xyz = bool(factor_xyz)
        self.n_xyz = 0
        self.sink = _FactorSink(self)
        self.force_collectives = False           # tools/rccl_selftest.py: run the collectives in a one-rank group too
        self.n_means = 0
        self.send = self.recv = None
        self._work = self._expect = None
        self._w_cache, self._cam_cache = {}, {}
        self._shard_cache = {}
        self.timing = None                       # a list: reduce() appends (t0, t_gathered, t_expanded, t_reduced) HIP events per call (bench.py)
        self.arena = None
        self._arena_setup()

    def _dense_named(self):
        from . import deform
        sh = set(_SH_PARAMS)
        return [(field, getattr(self.model, attr, None)) for field, attr in deform._MODEL_ATTRS.items()
                if attr not in sh and field in deform._GRADS and field != "background_deform_param"
                and not (field == "xyz_deform_param" and self.factor_xyz())]

    def factor_xyz(self):
        """xyz_deform_param's gradient also travels in factored form (two [No,3] upstream gradients per camera instead of the
        [No,3,Cx] rows: 43 MB of the 108 MB dense remainder at C3) -- opt-in (`factor_xyz=True`), on the HIP path with object
        Gaussians only, and it needs ONE deformation backward per camera: positions and flow points from the same
        get_deformed_pkg(t, flow_time=...) call (what gaussian_renderer.render() does for adgs.model), flow_times given to reduce()."""
        p = getattr(self.model, "xyz_deform_param", None)
        return self.want_factor_xyz and self.expand is None and p is not None and p.numel() > 0 and p.is_cuda

    def _arena_setup(self):
        """(Re)build the gradient arena when the model's dense parameters changed shape (densification)."""
        named = self._dense_named()
        if not any(p is not None and p.numel() > 0 and p.is_cuda for _, p in named):
            self.arena = None                      # CPU tensors (the gloo tests): plain path
        elif self.arena is None or not self.arena.matches(named):
            self.arena = GradArena(named)
        if self.arena is not None:
            self.arena.xyz_sink = self._xyz_sink if self.factor_xyz() else None
        self.model.grad_arena = self.arena

    def begin(self, n_cameras=None):
        """Start an iteration.  With the iteration's total camera count the all-gather of the factors is issued as soon as this
        rank's last backward has produced its factor -- from inside the autograd backward, so it overlaps with the rest of
        the backward (the deformation kernels) instead of starting in reduce()."""
        del self.sink[:]
        self.n_means = 0
        self.n_xyz = 0
        self._work = None
        self._expect = None
        self._arena_setup()
        if self.arena is not None:
            self.arena.reset()
        if n_cameras is not None:
            world, rank = self._world()
            k_max = (n_cameras + world - 1) // world
            n_local = len(range(rank, n_cameras, world))
            self._expect = (int(n_cameras), k_max, n_local)
            send, _, _ = self._buffers(k_max)
            for j in range(n_local, k_max):
                send[j].zero_()                      # this rank has no j-th camera: an all-zero factor contributes nothing

    def _collectives(self, world):
        return world > 1 or (self.force_collectives and dist.is_available() and dist.is_initialized())

    def _start_gather(self, k_max):
        world, _ = self._world()
        send = self.send
        if self.recv is None or self.recv.shape != (world, k_max, send.shape[1]):
            self.recv = torch.empty(world, k_max, send.shape[1], dtype=torch.float32, device=send.device)
        self._work = dist.all_gather_into_tensor(self.recv.view(-1), send[:k_max].reshape(-1), group=self.group, async_op=True)

    def _factor_added(self):
        if getattr(self, "_expect", None) is None or getattr(self, "_work", None) is not None:
            return
        n_total, k_max, n_local = self._expect
        world, _ = self._world()
        if len(self.sink) != n_local or self.n_means != n_local or not self._collectives(world) or (self.factor_xyz() and self.n_xyz != n_local):
            return
        send, P, _ = self._buffers(k_max)
        for j, f in enumerate(self.sink):
            if f.data_ptr() != send[j].data_ptr():
                send[j, :3 * P].copy_(f.reshape(-1))
                self.sink[j] = send[j, :3 * P].view(P, 3)
        self._start_gather(k_max)

    def _world(self):
        world = dist.get_world_size(self.group) if (dist.is_available() and dist.is_initialized()) else 1
        return world, (dist.get_rank(self.group) if world > 1 else 0)

    def _row0(self):
        bg = self.model.order_args.get("background", [0] * 6)
        return self.model.get_scene_pts_num if all(int(a) == 0 for a in bg) else 0

    def _buffers(self, k_need):
        """send [k, blob] / recv [world, k, blob], blob = [factor P*3 | means (P-row0)*3]; grown on demand, kept across iterations."""
        m = self.model
        P, row0 = m.get_pts_num, self._row0()
        blob = 3 * P + 3 * (P - row0) + (6 * m.get_obj_pts_num if self.factor_xyz() else 0)     # [factor | means | g_xyz, g_flow of the objects]
        ref = m._scene_xyz if m._scene_xyz.numel() else m._obj_xyz
        if self.send is None or self.send.shape[1] != blob or self.send.shape[0] < k_need or self.send.device != ref.device:
            k = max(k_need, 1 if self.send is None else self.send.shape[0])
            old = self.send
            self.send = torch.zeros(k, blob, dtype=torch.float32, device=ref.device)
            if old is not None and old.shape[1] == blob and old.device == ref.device:
                self.send[:old.shape[0]].copy_(old)          # keep what this iteration's earlier cameras already wrote
            self.recv = None
        return self.send, P, row0

    def _target(self, j, P):
        send, P_, _ = self._buffers(j + 1)
        if P != P_:
            raise RuntimeError("FactoredSHExchange: the rasterizer saw %d Gaussians, the model has %d" % (P, P_))
        return send[j, :3 * P].view(P, 3)

    def sink_for(self, xyz):
        """Register the next local camera (its deformed means3D) and return the sink for forward_rawsh."""
        j = self.n_means
        send, P, row0 = self._buffers(j + 1)
        if row0 < P:
            send[j, 3 * P:3 * P + 3 * (P - row0)].copy_(xyz.detach()[row0:].reshape(-1))
        self.n_means = j + 1
        return self.sink

    def _xyz_sink(self, g_xyz, g_flow):
        """Called by the deformation backward (through the arena) with the upstream gradients of the object positions at the camera
        time and at the flow time; stores them in the send buffer.  The all-gather of an announced iteration starts here, when
        the last local camera has delivered both its colour factor and these."""
        m = self.model
        j = self.n_xyz
        send, P, row0 = self._buffers(j + 1)
        No = m.get_obj_pts_num
        base = 3 * P + 3 * (P - row0)
        for k, g in enumerate((g_xyz, g_flow)):
            dst = send[j, base + 3 * No * k: base + 3 * No * (k + 1)]
            if g is None:
                dst.zero_()
            else:
                dst.copy_(g.reshape(-1))
        self.n_xyz = j + 1
        self._factor_added()
        return True

    def reduce(self, cam_times, cam_positions, dense_params=None, flow_times=None):
        """flow_times: per camera, the time stamp its flow points were evaluated at (get_deformed_pkg(t, flow_time=...)) or None;
        needed to expand xyz_deform_param's gradient when it travels in factored form."""
        m = self.model
        world, rank = self._world()
        n_total = len(cam_times)
        assert len(cam_positions) == n_total and n_total >= 1
        k_max = (n_total + world - 1) // world
        n_local = len(range(rank, n_total, world))
        if len(self.sink) != n_local or self.n_means != n_local:
            raise RuntimeError("FactoredSHExchange: rank %d rendered %d cameras (%d backward passes), the deal gives it %d"
                               % (rank, self.n_means, len(self.sink), n_local))
        send, P, row0 = self._buffers(k_max)
        Ns = m.get_scene_pts_num
        fx = self.factor_xyz()
        if fx and self.n_xyz != n_local:
            raise RuntimeError("FactoredSHExchange: %d of this rank's %d backward passes delivered the object position gradients (the "
                               "deformation must go through adgs.deform with this model's grad_arena)" % (self.n_xyz, n_local))
        if self._expect is not None and self._expect != (n_total, k_max, n_local):
            raise RuntimeError("FactoredSHExchange: begin() announced %d cameras, reduce() got %d" % (self._expect[0], n_total))
        coll = self._collectives(world)
        tm = self.timing
        stamp = (lambda: None) if tm is None else self._stamp
        ev = [stamp()]
        work = self._work                            # already in flight when begin(n_cameras) was used
        if work is None:
            for j, f in enumerate(self.sink):        # a backward that could not write in place (foreign sink use): copy now
                if f.data_ptr() != send[j].data_ptr():
                    send[j, :3 * P].copy_(f.reshape(-1))
            for j in range(n_local, k_max):
                send[j].zero_()                      # this rank has no j-th camera: an all-zero factor contributes nothing
            if coll:
                self._start_gather(k_max)
                work = self._work
        recv = self.recv if coll else send[:k_max].unsqueeze(0)
        if coll and work is not None and dist.get_backend(self.group) == "gloo":
            # gloo (the host-side dry runs; RCCL executes a communicator's collectives in issue order) hangs with four or more ranks when
            # an all-gather and all-reduces of device tensors are in flight together (reproduced without any of this code): finish the
            # gather before the reductions start
            work.wait()
        # the dense remainder: every parameter except the six SH tensors
        sh = [getattr(m, n, None) for n in _SH_PARAMS]
        if coll:
            if dense_params is None:
                skip = sh + ([m.xyz_deform_param] if fx else [])
                dense_params = [p for p in m.parameters() if not any(p is s for s in skip)]
            # one flat bucket for the whole dense remainder (108 MB at C3): one collective instead of one per large tensor --
            # xGMI is point-to-point and every extra collective costs a launch + synchronisation round
            named = self._dense_named() if self.arena is not None else []
            in_arena = []
            if self.arena is not None:
                # Which collectives a rank issues must not depend on what ITS cameras produced (a rank that got no camera in this
                # iteration, a parameter without gradient, a gradient autograd allocated elsewhere): every rank reduces the arena,
                # after moving into its slice whatever is not there yet (zeros for a missing gradient).
                for n, p in named:
                    if p is None or p.numel() == 0:
                        continue
                    if not self.arena.holds(n, p.grad):
                        off, numel = self.arena.offsets[n]
                        view = self.arena.flat[off:off + numel].view(self.arena.shapes[n])
                        if p.grad is None:
                            view.zero_()
                        else:
                            view.copy_(p.grad)
                        p.grad = view
                    in_arena.append(p)
            if self.arena is not None:
                # every dense gradient of the deformation backward sits in the arena: ONE in-place all-reduce, nothing to copy
                rest = [p for p in dense_params if not any(p is q for q in in_arena)]
                ws = reduce_flat(self.arena.flat, self.group, self._shard_cache)
                dense = allreduce_gradients_start(rest, group=self.group, force=self.force_collectives, in_place_bytes=1 << 40, bucket_bytes=1 << 40) \
                    if rest else dict(works=[], flats=[], big=[], average=False, world=world)
                dense["works"].extend(ws)
            else:
                dense = allreduce_gradients_start(dense_params, group=self.group, force=self.force_collectives, in_place_bytes=1 << 40, bucket_bytes=1 << 40)
        else:
            dense = None
        if work is not None:
            work.wait()
        ev.append(stamp())
        cams = []
        for g in range(n_total):                     # global camera order: identical summation order on every rank
            r, j = g % world, g // world
            cams.append((recv[r, j, :3 * P].view(P, 3), recv[r, j, 3 * P:3 * P + 3 * (P - row0)].view(P - row0, 3) if row0 < P else None, cam_positions[g]))
        C = int(m.shs_deform_param_scene.shape[-1]) if getattr(m, "shs_deform_param_scene", None) is not None else 0
        M = 1 + int(m._scene_shs_rest.shape[1])
        W = None
        if C > 0:
            wkey = (tuple(float(t) for t in cam_times), tuple(m.order_args["shs"]), C, str(send.device))
            W = self._w_cache.get(wkey)
            if W is None:
                if len(self._w_cache) > 256:
                    self._w_cache.clear()
                W = self._w_cache[wkey] = _dense_basis_weights(wkey[0], m.order_args["shs"], C, send.device)
        outs = []
        for s_ in sh:
            if s_ is None or s_.numel() == 0 or not s_.requires_grad:
                outs.append(None)
            else:
                if s_.grad is None or s_.grad.shape != s_.shape:
                    s_.grad = torch.empty_like(s_)
                outs.append(s_.grad)
        head = m._scene_xyz.detach() if row0 > 0 else None
        if self.expand is not None:
            self.expand(cams, W, C, P, Ns, row0, head, int(m.active_sh_degree), M, outs)
        else:
            hip_sh_grad_expand(cams, W, C, P, Ns, row0, head, int(m.active_sh_degree), M, outs, _cache=self._cam_cache)
        if fx:
            xp = m.xyz_deform_param
            No, Cx = xp.shape[0], int(xp.shape[-1])
            base = 3 * P + 3 * (P - row0)
            ft = [None] * n_total if flow_times is None else list(flow_times)
            key = ("xyz", tuple(float(t) for t in cam_times), tuple(None if t is None else float(t) for t in ft), tuple(m.order_args["xyz"]), Cx, str(send.device))
            Wx = self._w_cache.get(key)
            if Wx is None:
                rows = []
                for g in range(n_total):
                    for t in (cam_times[g], ft[g]):
                        if t is not None:
                            rows.append(_dense_basis_weights((float(t),), m.order_args["xyz"], Cx, "cpu")[0])
                Wx = self._w_cache[key] = torch.stack(rows).contiguous().to(send.device)
            terms = []
            for g in range(n_total):
                r, j = g % world, g // world
                terms.append(recv[r, j, base:base + 3 * No])
                if ft[g] is not None:
                    terms.append(recv[r, j, base + 3 * No:base + 6 * No])
            if xp.grad is None or xp.grad.shape != xp.shape:
                xp.grad = torch.empty_like(xp)
            hip_lin_grad_expand(terms, Wx, Cx, No, 1.0, xp.grad)
        ev.append(stamp())
        allreduce_gradients_finish(dense)            # the expansions above ran while the dense all-reduce was on the links
        ev.append(stamp())
        if tm is not None:
            tm.append(tuple(ev))
        self.begin()

    @staticmethod
    def _stamp():
        e = torch.cuda.Event(enable_timing=True)
        e.record()
        return e
