"""Multi-GPU data parallelism over training views (new in this build; the reference is single
process).

One process per GPU.  Every rank holds a full replica of the Gaussians, renders a different view
(rank r takes views r, r+G, r+2G, ... of a permutation that is seeded identically everywhere), runs
forward + backward locally, then the per-Gaussian gradients are summed across ranks with ONE
exchange step: an all-reduce (torch.distributed backend "nccl" = RCCL over xGMI on ROCm; "gloo" in
the CPU tests).  Views are independent, so there is no other collective on the data path.
"""
from __future__ import annotations

import torch


def shard_views(num_views: int, rank: int, world: int, epoch: int = 0, seed: int = 0) -> list:
    """Indices of the views rank `rank` renders in `epoch`: a seeded permutation dealt round-robin.
    Every rank computes the same permutation, so the shards are disjoint and cover all views."""
    g = torch.Generator()
    g.manual_seed(seed * 1_000_003 + epoch)
    perm = torch.randperm(num_views, generator=g).tolist()
    return perm[rank::world]


def coalesce_shared_storage(grads, max_waste: float = 0.25):
    """Gradients that are views of one buffer (the rasterizer's backward writes the six parameter
    gradients into a single flat allocation, _C.rasterize_gaussians_backward) are exchanged as ONE
    tensor spanning them: xGMI rings are per-link bound, so one 300 MB collective beats six smaller
    ones.  Tensors that do not share storage -- or whose span would be mostly foreign bytes -- are
    returned unchanged."""
    groups = {}
    for g in grads:
        key = (g.untyped_storage().data_ptr(), g.dtype, g.device) if g.is_contiguous() else id(g)
        groups.setdefault(key, []).append(g)
    out = []
    for key, gs in groups.items():
        if len(gs) == 1 or not isinstance(key, tuple):
            out.extend(gs)
            continue
        lo = min(g.storage_offset() for g in gs)
        hi = max(g.storage_offset() + g.numel() for g in gs)
        if sum(g.numel() for g in gs) < (1.0 - max_waste) * (hi - lo):
            out.extend(gs)
            continue
        out.append(torch.as_strided(gs[0], (hi - lo,), (1,), lo))
    return out


def allreduce_gradients(params, dist, bucket_bytes: int = 0):
    """Sum-all-reduce `.grad` of every parameter in place.

    bucket_bytes == 0: one asynchronous collective per gradient tensor, all in flight together
    (RCCL pipelines them; no flattening copy).  bucket_bytes > 0: gradients are packed into flat
    buckets of about that size first (fewer, larger collectives -- the better shape for xGMI's
    per-link-bound rings when there are many small tensors)."""
    grads = [p.grad for p in params if p.grad is not None]
    if not grads:
        return
    if bucket_bytes <= 0:
        works = [dist.all_reduce(g, op=dist.ReduceOp.SUM, async_op=True) for g in coalesce_shared_storage(grads)]
        for w in works:
            w.wait()
        return
    bucket, size = [], 0
    pending = []

    def flush():
        nonlocal bucket, size
        if not bucket:
            return
        flat = torch.cat([g.reshape(-1) for g in bucket])
        pending.append((dist.all_reduce(flat, op=dist.ReduceOp.SUM, async_op=True), flat, bucket))
        bucket, size = [], 0

    for g in grads:
        bucket.append(g)
        size += g.numel() * g.element_size()
        if size >= bucket_bytes:
            flush()
    flush()
    for work, flat, members in pending:
        work.wait()
        off = 0
        for g in members:
            n = g.numel()
            g.copy_(flat[off:off + n].view_as(g))
            off += n
