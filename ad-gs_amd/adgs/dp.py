"""Camera-parallel data parallelism for AD-GS training on one MI355X node.

The reference is single-GPU, one camera per iteration (train.py:55-61,74); there is no
collective anywhere in it.  Renders of different cameras share read-only Gaussian
parameters and are independent forward and backward, so the natural sharding is: one
process per GPU, full parameter replica per rank, the iteration's camera batch dealt
round-robin to the ranks, and ONE exchange step -- the sum of the per-camera parameter
gradients (plus the densification statistics) -- as an all-reduce over RCCL/xGMI
(SURVEY.md section 8(e)).  `torch.distributed` backend "nccl" is RCCL on ROCm; the CPU
tests use "gloo".
"""
import torch
import torch.distributed as dist


def shard_cameras(cameras, rank, world_size):
    """Round-robin deal of this iteration's camera batch (C4: 3 cameras over 4 ranks -> rank 3 idles)."""
    return [c for i, c in enumerate(cameras) if i % world_size == rank]


def _flatten_bucket(tensors, max_bytes):
    """Greedy bucketing of tensors into groups of at most max_bytes (one collective per group)."""
    buckets, cur, size = [], [], 0
    for t in tensors:
        nbytes = t.numel() * t.element_size()
        if cur and size + nbytes > max_bytes:
            buckets.append(cur); cur, size = [], 0
        cur.append(t); size += nbytes
    if cur:
        buckets.append(cur)
    return buckets


def allreduce_gradients(params, group=None, average=False, bucket_bytes=256 << 20, in_place_bytes=32 << 20):
    """Sum `.grad` of every parameter over all ranks (missing grads count as zero).

    Tensors of at least `in_place_bytes` are reduced in place, one async collective each (no flatten
    copy: at 1M Gaussians the SH gradient alone is 180 MB); the others are coalesced into flat buckets
    (the copies are ~80 MB in total at C3, 0.03 ms).  xGMI is point-to-point, so few large collectives
    are preferred over many small ones: C3 ends up with 6 in-place reductions and one bucket.
    """
    if not dist.is_available() or not dist.is_initialized() or dist.get_world_size(group) == 1:
        return
    world = dist.get_world_size(group)
    grads = []
    for p in params:
        if p.grad is None:
            p.grad = torch.zeros_like(p)
        grads.append(p.grad)
    big = [g for g in grads if g.numel() * g.element_size() >= in_place_bytes]
    small = [g for g in grads if g.numel() * g.element_size() < in_place_bytes]
    works = [dist.all_reduce(g, op=dist.ReduceOp.SUM, group=group, async_op=True) for g in big]
    flats = []
    for bucket in _flatten_bucket(small, bucket_bytes):
        flat = torch.cat([g.reshape(-1) for g in bucket])
        works.append(dist.all_reduce(flat, op=dist.ReduceOp.SUM, group=group, async_op=True))
        flats.append((flat, bucket))
    for w in works:
        w.wait()
    for flat, bucket in flats:
        off = 0
        for g in bucket:
            g.copy_(flat[off:off + g.numel()].view_as(g)); off += g.numel()
    if average:
        for g in grads:
            g.div_(world)


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
