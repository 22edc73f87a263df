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


def shard_views(num_views: int, rank: int, world: int, epoch: int = 0, seed: int = 0, even: str = "pad") -> list:
    """Indices of the views rank `rank` renders in `epoch`: a seeded permutation dealt round-robin.
    Every rank computes the same permutation, so the shards cover all views.

    A training loop issues one gradient exchange per local view, so every rank must hold the SAME number of views or
    the ranks with fewer never enter the last collective.  even="pad" (default): when num_views is not a multiple of
    `world` the permutation is extended by wrapping around to its start, every rank gets ceil(num_views / world) views
    and a few views are rendered twice in that epoch; even="drop": the tail is dropped, floor(num_views / world) views
    per rank; even="none": the plain deal (a partition; shard lengths may differ by one -- only for loops that do not
    exchange per view)."""
    if even not in ("pad", "drop", "none"):
        raise ValueError("even must be 'pad', 'drop' or 'none'")
    g = torch.Generator()
    g.manual_seed(seed * 1_000_003 + epoch)
    perm = torch.randperm(num_views, generator=g).tolist()
    rem = num_views % world
    if rem and num_views:
        if even == "pad":
            need = -(-num_views // world) * world  # wrap around as often as it takes (num_views may be < world)
            perm = (perm * (need // num_views + 1))[:need]
        elif even == "drop":
            perm = perm[:num_views - rem]
    return perm[rank::world]


def coalesce_shared_storage(grads, max_waste: float = 0.25):
    """Gradients that are views of one buffer (the rasterizer's backward writes the six parameter
    gradients into a single flat allocation, _C.rasterize_gaussians_backward) are exchanged as ONE
    tensor spanning them: xGMI rings are per-link bound, so one 300 MB collective beats six smaller
    ones.  Tensors that do not share storage -- or whose span would be mostly foreign bytes -- are
    returned unchanged."""
    groups = {}
    for g in grads:
        # (the storage's base address from the view's own pointer and offset: g.untyped_storage() would create a Python storage
        # object that keeps the StorageImpl referenced for good -- measured on torch 2.10: the binding's gradient pool, which
        # reuses a buffer only when nobody else holds its storage, then never saw an exchanged buffer again)
        key = ((g.data_ptr() - g.storage_offset() * g.element_size(), g.dtype, g.device) if g.is_contiguous() else id(g))
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


def _touch(t):
    """Marks `t` as written in place.  c10d collectives write their output tensors WITHOUT bumping the autograd version
    counter (checked on torch 2.10: all_reduce on an as_strided view leaves `_version` unchanged, blocking or async), and the
    compiled binding's gradient-buffer pool (torch_binding.cpp, backward_ex) hands a buffer out again as "rows of invisible
    Gaussians still hold zeros" exactly when that counter has not moved.  A buffer another rank's rows were summed into is
    NOT such a buffer: every helper here therefore bumps the counter of every tensor it passes to an in-place collective
    (views share their base's counter), so the pool rewrites the buffer in full.  No kernel, no sync."""
    torch.autograd.graph.increment_version(t)
    return t


def allreduce_gradients(params, dist, bucket_bytes: int = 0):
    """Sum-all-reduce `.grad` of every parameter in place.

    SUM, not mean: the result is the gradient of the sum of the ranks' per-view losses, i.e. what the reference's
    single process would accumulate over the same views (all-reduced == sum of the single-view
    gradients).  A caller that wants the data-parallel mean divides by dist.get_world_size() (or scales the lr).

    bucket_bytes == 0: one asynchronous collective per gradient tensor, all in flight together
    (RCCL pipelines them; no flattening copy).  bucket_bytes > 0: gradients are packed into flat
    buckets of about that size first (fewer, larger collectives -- the better shape for xGMI's
    per-link-bound rings when there are many small tensors)."""
    grads = [p.grad for p in params if p.grad is not None]
    if not grads:
        return
    if bucket_bytes <= 0:
        works = [dist.all_reduce(_touch(g), op=dist.ReduceOp.SUM, async_op=True) for g in coalesce_shared_storage(grads)]
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


def allreduce_gradients_visible(params, visible, dist, dense_above: float = 0.9, per_gaussian=None, check_zero_rows=False,
                                direct: bool = False):
    """Sum-all-reduce `.grad` of every parameter, sending only the rows of Gaussians that were VISIBLE in at least one
    rank's view(s) of this step.

    A Gaussian culled in a view (radius 0) gets an exactly-zero gradient row in every tensor of that view's backward, so
    a row that no rank saw is zero everywhere and its sum needs no wire.  The ranks OR their visibility masks (one
    all-reduce(MAX) of P bytes: 1 MB at 1 M Gaussians), gather the rows of the union from every per-Gaussian gradient
    into one [n_union, 75] buffer, all-reduce that, and scatter it back; rows outside the union keep their local zeros,
    which IS the sum.  xGMI rings are per-link bound, so volume is what counts: at the headline scene a
    view sees 51 % of the Gaussians; BASELINE config 5 (6 M Gaussians, a 512 x 512 close-up per rank) sees far fewer.

    PRECONDITION (hard): every gradient of a per-Gaussian tensor comes from the rasterizer's backward ALONE -- a row the
    rasterizer did not see is exactly zero.  The reference's train.py has only render losses, so this holds there; a weight
    decay or a regulariser on opacities / scales would leave non-zero rows outside the union and the ranks would diverge
    silently.  check_zero_rows=True verifies the premise on every call (one more reduction + host read: debugging aid) and
    raises if a row outside the union is non-zero.

    visible: bool / uint8 [P] of this rank -- `radii > 0` of its view, OR-ed over the views it accumulated locally.
    per_gaussian: the parameters whose gradient rows are per Gaussian (sent by row); every other parameter (the decoder,
    the code book) is all-reduced whole.  None: every parameter whose gradient's leading dimension is P -- only safe when
    no other tensor happens to have P rows (a [P, ...] tensor that is NOT per Gaussian would only be row-reduced), so
    callers with a decoder or a code book should pass the list.  If the union covers more than `dense_above` of the
    Gaussians the plain all-reduce is used (the gather would only add copies).
    Costs one host synchronisation (the union's size); use it where the exchange is waited for anyway.
    Returns the number of rows sent (P when it fell back to the dense exchange)."""
    P = int(visible.shape[0])
    vis = visible.to(torch.uint8).clone()
    dist.all_reduce(vis, op=dist.ReduceOp.MAX)
    idx = torch.nonzero(vis, as_tuple=True)[0]
    n = int(idx.numel())
    with_grad = [p for p in params if p.grad is not None]
    if per_gaussian is None:
        rows = [p for p in with_grad if p.grad.dim() >= 1 and p.grad.shape[0] == P]
    else:
        chosen = {id(p) for p in per_gaussian}
        rows = [p for p in with_grad if id(p) in chosen]
        for p in rows:
            if p.grad.dim() < 1 or p.grad.shape[0] != P:
                raise ValueError(f"per_gaussian parameter with gradient shape {tuple(p.grad.shape)}: leading dimension is not P={P}")
    row_ids = {id(p) for p in rows}
    rest = [p for p in with_grad if id(p) not in row_ids]
    if n >= dense_above * P or not rows:
        (allreduce_gradients_direct if direct else allreduce_gradients)(with_grad, dist)
        return P
    works = [dist.all_reduce(_touch(g), op=dist.ReduceOp.SUM, async_op=True)
             for g in coalesce_shared_storage([p.grad for p in rest])] if rest else []
    views = []
    for p in rows:
        # .view, never .reshape: a reshape of a non-contiguous gradient is a COPY, and the index_copy_ below would write the
        # reduced rows into that copy and lose them
        if not p.grad.is_contiguous():
            raise ValueError("allreduce_gradients_visible needs contiguous per-Gaussian gradients (got strides "
                             f"{tuple(p.grad.stride())} for shape {tuple(p.grad.shape)})")
        views.append(p.grad.view(P, -1))
    widths = [int(v.shape[1]) for v in views]
    if check_zero_rows:
        unseen = (vis == 0)
        worst = max((float(v[unseen].abs().max()) if bool(unseen.any()) else 0.0) for v in views)
        if worst != 0.0:
            raise RuntimeError(f"allreduce_gradients_visible: a gradient row of a Gaussian no rank saw is non-zero (|g| up to "
                               f"{worst:.3e}): some loss reaches the per-Gaussian tensors without going through the rasterizer "
                               "(weight decay, a regulariser); use allreduce_gradients for such steps")
    pack = torch.cat([v.index_select(0, idx) for v in views], dim=1)
    if direct:
        allreduce_gradients_direct([_G(pack)], dist)
    else:
        dist.all_reduce(pack, op=dist.ReduceOp.SUM)
    off = 0
    for v, w in zip(views, widths):
        v.index_copy_(0, idx, pack[:, off:off + w])
        off += w
    for w in works:
        w.wait()
    return n


# ---- direct exchange: reduce-scatter + all-gather over all links -----------------------------------------------------------
def _direct_spans(grads, world):
    """(span, padded) pairs for the direct exchange: the gradients coalesced as for the all-reduce; a span whose length is
    not a multiple of `world` travels through a zero-padded copy (the rasterizer's flat gradient buffer is laid out in
    64-float sections, so its span divides by 2, 4 and 8 as it stands: no copy on the data path)."""
    out = []
    for span in coalesce_shared_storage(grads):
        flat = span if span.dim() == 1 and span.is_contiguous() else None
        if flat is None and span.is_contiguous():
            flat = span.view(-1)
        n = span.numel()
        if flat is not None and n % world == 0:
            out.append((span, flat, False))
        else:
            padded = torch.zeros(-(-n // world) * world, dtype=span.dtype, device=span.device)
            padded[:n].copy_(span.reshape(-1))
            out.append((span, padded, True))
    return out


def _issue_sum(grads, dist, direct: bool):
    """Issues the sum of `grads` over the ranks without waiting: (works, finish, keep).  direct=False: one asynchronous
    all_reduce per coalesced span; direct=True: reduce-scatter + all-gather per span (allreduce_gradients_direct)."""
    if not grads:
        return [], None, None
    if not direct:
        spans = coalesce_shared_storage(grads)
        return [dist.all_reduce(_touch(g), op=dist.ReduceOp.SUM, async_op=True) for g in spans], None, spans
    h = allreduce_gradients_direct([_G(g) for g in grads], dist, async_op=True)
    return h.works, h._finish, h.keep


class _G:
    """(a bare gradient tensor dressed as a parameter for the functions that take parameters)"""
    __slots__ = ("grad",)

    def __init__(self, g):
        self.grad = g


def allreduce_gradients_direct(params, dist, async_op: bool = False):
    """The same sum as allreduce_gradients, spelled as the two collectives SURVEY.md 8(e) prefers on xGMI: ONE
    reduce-scatter (rank r receives and sums shard r of every rank's flat gradient span: every rank talks to all G-1 peers
    at once, 1/G of the bytes per link) followed by ONE all-gather of the reduced shards.  A ring all-reduce is bound by a
    single 153 GB/s link (300 MB -> 3.4 ms at 8 GPUs); the direct form spreads the same bytes over all seven (0.49 ms in
    the link model, exchange_model_ms).  Which algorithm RCCL picks for a plain all_reduce is RCCL's business; this form
    fixes the communication PATTERN, whatever the library does inside each collective.

    In place: the shard a rank reduces is a slice of its own span (reduce_scatter_tensor with output = input[rank]), and
    the all-gather writes every shard back where it came from -- no staging copy for the rasterizer's flat gradient buffer.
    Every rank ends up with the same bits (each shard is summed on one rank and copied to the others).
    async_op: return a PendingExchange (wait() before reading the gradients) instead of waiting here."""
    grads = [p.grad for p in params if p.grad is not None]
    world, rank = dist.get_world_size(), dist.get_rank()
    if not grads:
        return PendingExchange([], None) if async_op else None
    spans = _direct_spans(grads, world)
    # RCCL runs the collectives of one communicator in issue order on its stream: both can be queued at once.  Other
    # backends (gloo in the CPU tests: a pool of worker threads) give no such order between asynchronous collectives, so
    # there the all-gathers are issued once the reduce-scatters have completed (in wait()).
    ordered = "nccl" in str(dist.get_backend()).lower() and all(f.is_cuda for _s, f, _p in spans)  # (composite strings: 'cuda:nccl,cpu:gloo')
    shards, works, copies = [], [], []
    for span, flat, padded in spans:
        shard = flat.numel() // world
        mine = flat[rank * shard:(rank + 1) * shard]
        shards.append((flat, mine))
        works.append(dist.reduce_scatter_tensor(mine, _touch(flat), op=dist.ReduceOp.SUM, async_op=True))
        if ordered:
            works.append(dist.all_gather_into_tensor(flat, mine, async_op=True))
        if padded:
            copies.append((span, flat))

    def finish():
        if not ordered:
            for w in [dist.all_gather_into_tensor(flat, mine, async_op=True) for flat, mine in shards]:
                w.wait()
        for span, flat in copies:
            span.copy_(flat[:span.numel()].view_as(span))
    h = PendingExchange(works, (grads, spans), finish if (copies or not ordered) else None)
    if async_op:
        return h
    h.wait()
    return None


def pick_exchange(nbytes: float, world: int) -> str:
    """"direct" or "ring" (= a plain all_reduce, whatever RCCL makes of it): whichever SURVEY.md 8(e)'s link model prices
    lower for `nbytes` over `world` GPUs.  With more than two fully connected GPUs that is the direct form for every size the
    model distinguishes; with two the forms move the same bytes over the same single link and the plain all-reduce (one
    collective instead of two) is kept."""
    m = exchange_model_ms(nbytes, world)
    return "direct" if world > 2 and m["direct"] < m["ring"] else "ring"


def exchange_model_ms(nbytes: float, world: int, link_GBps: float = 153.0, links: int = 7) -> dict:
    """SURVEY.md 8(e)'s xGMI cost model for a sum all-reduce of `nbytes` over `world` fully connected GPUs: a ring is
    bound by ONE link (2 (G-1)/G bytes / 153 GB/s); a direct reduce-scatter + all-gather spreads 2 bytes/G over each of
    the G-1 links it uses at once.  Milliseconds; what RCCL actually achieves lies in between and is measured."""
    if world <= 1 or nbytes <= 0:
        return {"ring": 0.0, "direct": 0.0}
    ring = 2.0 * (world - 1) / world * nbytes / (link_GBps * 1e9) * 1e3
    direct = 2.0 * (nbytes / world) * (world - 1) / (min(links, world - 1) * link_GBps * 1e9) * 1e3
    return {"ring": ring, "direct": direct}


def allreduce_gradients_sh_factored(params, sh_leaves, means3D, factor, dist, reconstruct=None, direct: bool = False):
    """Data-parallel gradient exchange with the SH gradient sent as its FACTORS.

    One view per rank.  dL/dSH is 48 of the 75 gradient floats of a Gaussian, but for one view it is the outer
    product  basis_k(direction camera -> Gaussian) x gcol  of 16 numbers every rank can compute itself and the
    clamp-masked colour gradient gcol (3 floats).  So instead of all-reducing 192 bytes per Gaussian, the ranks
    all-gather gcol (12 bytes per Gaussian and view) and their camera centres, and each rebuilds the SUM over all
    views with one kernel (goi_raster_sh_grad_from_views; views added in rank order: every rank gets the same bits).
    The other gradients (`params`: every leaf except the SH tensors) are all-reduced as usual; that collective is
    on the wire while the SH part is rebuilt.  xGMI rings are per-link bound: at 1 M
    Gaussians, degree 3, 8 ranks this moves 108 MB (all-reduce) + 96 MB (all-gather) instead of 300 MB.

    factor: rasterizer.take_sh_factor() of this rank's backward (run with set_backward_mode(sh_factored=True)).
    sh_leaves: the SH parameter(s) whose concatenation along dim 1 is the [P,M,3] tensor the rasterizer was given --
    (features_dc [P,1,3], features_rest [P,M-1,3]) for the reference's GaussianModel, one [P,M,3] tensor for
    render.GaussianSet; they receive .grad here.
    reconstruct(means3D, campos[V,3], gcol[V,P,3], degree, M) -> [P,M,3]; default: the HIP kernel."""
    if factor is None:
        raise RuntimeError("no SH factor: run the backward with rasterizer.set_backward_mode(sh_factored=True)")
    if reconstruct is None:
        from . import _C
        reconstruct = _C.sh_grad_from_views
    world = dist.get_world_size()
    gcol = factor["gcol"].detach().contiguous()
    campos = factor["campos"].detach().reshape(1, 3).to(gcol.dtype).contiguous()
    all_g = torch.empty((world,) + tuple(gcol.shape), dtype=gcol.dtype, device=gcol.device)
    all_c = torch.empty((world, 3), dtype=gcol.dtype, device=gcol.device)
    try:
        dist.all_gather_into_tensor(all_g, gcol)
        dist.all_gather_into_tensor(all_c, campos)
    except (RuntimeError, AttributeError, NotImplementedError):  # a backend without the flat form
        dist.all_gather(list(all_g.unbind(0)), gcol)
        dist.all_gather(list(all_c.unbind(0)), campos.reshape(3))
    # one communicator runs its collectives in order: the (small) gathers go first, then the all-reduce of the other
    # gradients is queued and the reconstruction kernel runs on the compute stream while it is on the wire
    grads = [p.grad for p in params if p.grad is not None]
    works, fin, _keep = _issue_sum(grads, dist, direct)  # (direct: reduce-scatter + all-gather instead of RCCL's all_reduce)
    dsh = reconstruct(means3D.detach(), all_c, all_g, int(factor["degree"]), int(factor["M"]))
    k = 0
    for leaf in sh_leaves:
        n = int(leaf.shape[1])
        part = dsh[:, k:k + n, :].contiguous() if (k or n != dsh.shape[1]) else dsh
        if leaf.grad is None:  # autograd semantics: a gradient that is already there is accumulated into
            leaf.grad = part
        else:
            leaf.grad.add_(part)
        k += n
    if k != dsh.shape[1]:
        raise ValueError(f"sh_leaves hold {k} coefficients, the rasterizer was given {dsh.shape[1]}")
    for w in works:
        w.wait()
    if fin is not None:
        fin()


# ---- exchange in flight: overlap the collective with the next view's render ---------------------------------------------
class PendingExchange:
    """Handle of a gradient exchange that is on the wire.  It owns the gradient tensors being reduced (so the caller
    may drop or replace `.grad` at once) and finishes the exchange in wait(): the collectives are waited for on the
    current stream and, for the factored form, dL/dSH is rebuilt and handed to the SH leaves."""

    def __init__(self, works, keep, finish=None):
        self.works, self.keep, self._finish, self.done = list(works), keep, finish, False

    def wait(self):
        if self.done:
            return
        for w in self.works:
            w.wait()
        if self._finish is not None:
            self._finish()
        self.done, self.works, self._finish = True, [], None


def allreduce_gradients_async(params, dist, direct: bool = False) -> PendingExchange:
    """allreduce_gradients without the wait: the collectives are issued (RCCL runs them on its own stream, ordered
    after the work already queued on the current one) and the handle is returned at once.  The next view's forward and
    backward can be enqueued while the reduction is on the wire -- the backward writes its gradients into a fresh flat
    buffer every call, so the buffer in flight is never touched ("double buffering" by allocation).  Legal wherever
    the reduced gradient is consumed later than the next render starts: several views accumulated per optimizer step,
    or one-step-delayed updates.  wait() before reading the reduced `.grad` tensors (they are the ones the parameters
    held when this was called)."""
    grads = [p.grad for p in params if p.grad is not None]
    works, fin, keep = _issue_sum(grads, dist, direct)
    return PendingExchange(works, (grads, keep), fin)


def allreduce_gradients_sh_factored_async(params, sh_leaves, means3D, factor, dist, reconstruct=None,
                                          direct: bool = False) -> PendingExchange:
    """allreduce_gradients_sh_factored with the collectives left in flight; wait() rebuilds dL/dSH from the gathered
    factors and hands each SH leaf its part exactly as the blocking variant does: assigned to `leaf.grad`, or added to a
    gradient that is already there (in sh_factored mode autograd gives the SH leaves none of its own).  The same tensors
    are also left in `handle.sh_grads[id(leaf)]` for callers whose leaves have moved on to the next step by then."""
    if factor is None:
        raise RuntimeError("no SH factor: run the backward with rasterizer.set_backward_mode(sh_factored=True)")
    if reconstruct is None:
        from . import _C
        reconstruct = _C.sh_grad_from_views
    world = dist.get_world_size()
    gcol = factor["gcol"].detach().contiguous()
    campos = factor["campos"].detach().reshape(1, 3).to(gcol.dtype).contiguous()
    all_g = torch.empty((world,) + tuple(gcol.shape), dtype=gcol.dtype, device=gcol.device)
    all_c = torch.empty((world, 3), dtype=gcol.dtype, device=gcol.device)
    try:
        works = [dist.all_gather_into_tensor(all_g, gcol, async_op=True),
                 dist.all_gather_into_tensor(all_c, campos, async_op=True)]
    except (RuntimeError, AttributeError, NotImplementedError):  # a backend without the flat form
        works = [dist.all_gather(list(all_g.unbind(0)), gcol, async_op=True),
                 dist.all_gather(list(all_c.unbind(0)), campos.reshape(3), async_op=True)]
    grads = [p.grad for p in params if p.grad is not None]
    sum_works, sum_fin, spans = _issue_sum(grads, dist, direct)
    works += sum_works
    means = means3D.detach()
    degree, M = int(factor["degree"]), int(factor["M"])
    result = {}

    def finish():
        if sum_fin is not None:
            sum_fin()
        dsh = reconstruct(means, all_c, all_g, degree, M)
        k = 0
        for leaf in sh_leaves:
            n = int(leaf.shape[1])
            part = dsh[:, k:k + n, :].contiguous() if (k or n != dsh.shape[1]) else dsh
            result[id(leaf)] = part
            if leaf.grad is None:  # (as the blocking variant: a gradient that is already there is accumulated into)
                leaf.grad = part
            else:
                leaf.grad.add_(part)
            k += n
        if k != dsh.shape[1]:
            raise ValueError(f"sh_leaves hold {k} coefficients, the rasterizer was given {dsh.shape[1]}")

    h = PendingExchange(works, (grads, spans, gcol, campos, all_g, all_c), finish)
    h.sh_grads = result  # id(leaf) -> its reduced dL/dSH part, filled by wait() (the leaves' .grad may have moved on)
    return h


# ---- a multi-view batch on ONE rank: the K views of a rank's share of a batch, in flight on several HIP streams -----------------
_VIEW_STREAMS = {}  # (device index, n) -> list of side streams (created once: a stream owns its allocator pool and its scratch)


def _view_streams(dev: torch.device, n: int):
    key = (dev.index if dev.index is not None else torch.cuda.current_device(), n)
    if key not in _VIEW_STREAMS:
        _VIEW_STREAMS[key] = [torch.cuda.Stream(device=dev) for _ in range(n)]
    return _VIEW_STREAMS[key]


def backward_views(views, render_view, upstream, params, streams: int = 1, accumulate: bool = False, on_device: bool = True):
    """Forward + backward of the K views of a multi-view batch (BASELINE config 4's "8-view batch" on fewer than 8 GPUs, or
    --views-per-exchange K before one gradient exchange) with the views ALTERNATING over `streams` HIP streams of this GPU,
    and the SUM of their gradients left in `p.grad` of every parameter -- what K serial `loss.backward()` calls leave there.
    train.py's own loop (one view, optimiser step, next view: train.py:96-198) cannot batch; a batch of views between two
    optimiser steps can.

      views        sequence of K view descriptors (cameras)
      render_view  view -> dict of outputs (e.g. lambda cam: render(cam, pc, pipe, bg))
      upstream     (outputs, k) -> (tensors, grad_tensors): what is differentiated for view k
      params       the leaves whose gradients are wanted
      streams      HIP streams the views alternate over (1 = one after the other on the current stream)
      accumulate   add to the gradients already in p.grad instead of replacing them
      on_device    sum the rasterizer's gradients on the device (below); False: K dense sums of leaf gradients

    MEASURED on one MI355X, headline workload (bench.py `views_in_flight_batch`, round 6): single-view steps 759 views/s; a batch
    through this function 717-728 (streams 1: a batch's gradients are the caller's until the next batch, so the gradient-buffer
    pool cannot skip the rows that already hold zeros) and 684-704 with streams 2 -- two views whose gradients need NOT be summed
    overlap to 824-839 views/s (`two_views_in_flight`), but with one sum the backwards have to take turns and what is left to
    overlap (a forward beside the other view's backward) does not pay for the joins at the batch's ends.  Hence streams = 1 by
    default; the parameter stays for GPUs / sizes where the balance differs.

    Two things can make a batch cheaper than K steps.  (1) Independent views overlap: one view's latency-bound kernels (sorts, scans,
    kernel tails) run beside the other's blends.  (2) on_device: the K views share their operands (the activations of the same
    parameters), so the rasterizer's backward of view 2, 3, ... ADDS its per-Gaussian gradients to view 1's tensors inside the
    per-Gaussian kernel (rasterizer.accumulate_gradients -> goi_raster_backward3, GOI_BACKWARD_ACCUMULATE: a view reads and rewrites
    the rows of the Gaussians it sees -- half of the scene -- instead of a dense [P, 75 + S] addition per view, and no row of an
    invisible Gaussian is zero-filled or touched), one sum per stream; the per-stream sums are added once and back-propagated
    through the activations ONCE instead of K times.  The result equals the serial sum up to the association of fp32 additions
    (tests/test_gpu_dist.py).  Gradients that do not come through the rasterizer, and backward modes that do not accumulate
    (semantics-only, factored, ctypes binding), are summed as leaf gradients, per stream in view order, then in stream order.
    Returns the list of per-view outputs (detached)."""
    import contextlib

    from . import rasterizer
    params = [p for p in params if p.requires_grad]
    if not views:
        return []
    dev = params[0].device
    n = max(1, min(int(streams), len(views)))
    cur = torch.cuda.current_stream(dev)
    side = [cur] if n == 1 else _view_streams(dev, n)
    sums = [None] * n
    # on_device: ONE sum for all streams.  The backward of view k waits (an event) for the backward of view k - 1, whichever stream
    # that ran on, so the accumulating kernels never meet -- what overlaps is view k + 1's forward (front end, forward blend) with
    # view k's backward; no per-stream sums to add at the end, one zero-fill per batch.
    shared = dict(grads=None, inputs=None, views=0)
    states = [shared if on_device else dict(grads=None, inputs=None, views=0) for _ in range(n)]
    bwd_done = None
    outs = [None] * len(views)
    for s_ in side:
        if s_ is not cur:
            s_.wait_stream(cur)  # the parameters' latest values (an optimiser step on the current stream) are visible
    for k, view in enumerate(views):
        si = k % n
        with torch.cuda.stream(side[si]):
            ctx = rasterizer.accumulate_gradients(states[si]) if on_device else contextlib.nullcontext()
            with ctx:
                out = render_view(view)
                tensors, grads = upstream(out, k)
                if on_device and n > 1 and bwd_done is not None:
                    side[si].wait_event(bwd_done)
                g = torch.autograd.grad(tensors, params, grads, allow_unused=True)
                if on_device and n > 1:
                    bwd_done = torch.cuda.Event()
                    bwd_done.record(side[si])
                    if shared["grads"] is not None:
                        for t in shared["grads"]:
                            if t is not None:
                                t.record_stream(side[si])
            if sums[si] is None:
                sums[si] = list(g)
            else:
                for i, t in enumerate(g):
                    if t is None:
                        continue
                    # (out of place: a gradient may be a view of the rasterizer's pooled buffer)
                    sums[si][i] = t if sums[si][i] is None else sums[si][i] + t
            outs[k] = {key: (v.detach() if torch.is_tensor(v) else v) for key, v in out.items()}
            for t in g:  # the caching allocator must not hand these blocks to another stream before this one is done with them
                if t is not None and side[si] is not cur:
                    t.record_stream(side[si])
    for s_ in side:
        if s_ is not cur:
            cur.wait_stream(s_)
    # ---- the rasterizer's gradients, summed on the device per stream: add the streams' sums, then through the activations once
    op_total, op_inputs = None, None
    for si in range(n):
        st = states[si]
        if st["grads"] is None or (on_device and si > 0):  # (on_device: the one shared sum)
            continue
        if side[si] is not cur:
            for t in st["grads"]:
                if t is not None:
                    t.record_stream(cur)
        if op_total is None:
            op_total, op_inputs = list(st["grads"]), st["inputs"]
        else:
            for i, t in enumerate(st["grads"]):
                if t is not None and op_total[i] is not None:
                    op_total[i].add_(t)
    from_op = [None] * len(params)
    if op_total is not None:
        # operands (means3D, sh, colors_precomp, semantics, opacities, scales, rotations, cov3Ds_precomp) <- positions of their
        # gradients in the op's result (dL_dmeans2D, dL_dcolors, dL_dsemantics, dL_dopacity, dL_dmeans3D, dL_dcov3D, dL_dsh, ...)
        pairs = [(inp, op_total[j]) for inp, j in zip(op_inputs, (4, 6, 1, 2, 3, 7, 8, 5))
                 if torch.is_tensor(inp) and inp.requires_grad and op_total[j] is not None and op_total[j].numel() == inp.numel()]
        if pairs:
            from_op = torch.autograd.grad([a_ for a_, _b in pairs], params, [b_.view_as(a_) for a_, b_ in pairs], allow_unused=True)
    for i, p in enumerate(params):
        tot = from_op[i]
        for si in range(n):
            t = None if sums[si] is None else sums[si][i]
            if t is None:
                continue
            if side[si] is not cur:
                t.record_stream(cur)
            tot = t if tot is None else tot + t
        if tot is None:
            continue
        if accumulate and p.grad is not None:
            p.grad = p.grad + tot
        else:
            p.grad = tot
    return outs
