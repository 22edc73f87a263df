"""world_size-2 gloo tests (CPU) of the multi-GPU path: view sharding and the gradient all-reduce
(all-reduced gradient == sum of the single-view gradients)."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from goi_hyperplane_amd.dist import allreduce_gradients, shard_views


def test_shard_views_is_a_partition():
    for world in (1, 2, 3, 8):
        for epoch in (0, 1):
            shards = [shard_views(37, r, world, epoch, even="none") for r in range(world)]
            flat = sorted(i for s in shards for i in s)
            assert flat == list(range(37))
    assert shard_views(10, 0, 2, epoch=0) != shard_views(10, 0, 2, epoch=1)


def test_shard_views_gives_every_rank_the_same_number_of_views():
    """One gradient exchange per local view: unequal shards would leave the longer ranks alone in the last
    collective.  Default: pad by wrapping around; "drop": drop the tail."""
    for n in (1, 5, 8, 37, 200):
        for world in (1, 2, 3, 8):
            pad = [shard_views(n, r, world) for r in range(world)]
            assert {len(s) for s in pad} == {-(-n // world)}
            assert {i for s in pad for i in s} == set(range(n))  # every view is still rendered
            drop = [shard_views(n, r, world, even="drop") for r in range(world)]
            assert {len(s) for s in drop} == {n // world}
            flat = [i for s in drop for i in s]
            assert len(flat) == len(set(flat))


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, bucket_bytes, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    # per-view "gradients" from the CPU oracle of a tiny scene, one view per rank
    from goi_hyperplane_amd.scene import make_camera, make_scene
    from oracle import oracle
    sc = make_scene(300, S=10, seed=2, log_scale_mean=-2.4)
    views = shard_views(world, rank, world)
    # (the visible-rows exchange wants views that cull: a narrow field of view, well apart)
    cam = make_camera(64, 48, yaw=0.1 * views[0]) if bucket_bytes != -2 else make_camera(64, 48, fovx=0.35, yaw=0.25 * views[0])
    o = oracle.from_scene(sc, cam, threads=1)
    fwd = o.forward()
    HW = 64 * 48
    g = o.backward(np.full((3, 48, 64), 1 / HW, np.float32), np.full((10, 48, 64), 1 / HW, np.float32))
    names = ["means3D", "sh", "semantics", "opacity", "scales", "rotations"]
    params = []
    flat = torch.empty(sum((g[n].size + 63) // 64 * 64 for n in names))  # the layout of _C.rasterize_gaussians_backward
    off = 0
    for n in names:
        p = torch.nn.Parameter(torch.zeros(g[n].shape))
        v = flat[off:off + g[n].size].view(g[n].shape)
        v.copy_(torch.tensor(g[n]))
        p.grad = v
        off += (g[n].size + 63) // 64 * 64
        params.append(p)
    local = [p.grad.clone() for p in params]
    version_before = flat._version  # (the binding's gradient pool reuses a buffer whose counter has not moved)
    if bucket_bytes == -1:  # the exchange left in flight: the handle owns the gradients, the parameters move on
        from goi_hyperplane_amd.dist import allreduce_gradients_async
        held = [p.grad for p in params]
        h = allreduce_gradients_async(params, dist)
        for p in params:
            p.grad = None  # what a training loop does before the next view
        h.wait()
        for p, g_ in zip(params, held):
            p.grad = g_
    elif bucket_bytes == -2:  # only the rows of Gaussians some rank saw travel
        from goi_hyperplane_amd.dist import allreduce_gradients_visible
        extra = torch.nn.Parameter(torch.zeros(5, 3))  # a tensor that is not per-Gaussian (decoder, code book)
        extra.grad = torch.full((5, 3), float(rank + 1))
        sent = allreduce_gradients_visible(params + [extra], torch.tensor(fwd.radii > 0), dist, dense_above=2.0)
        assert 0 < sent < 300, sent  # a real subset: the test would be vacuous otherwise
        assert float(extra.grad[0, 0]) == sum(range(1, world + 1))
    elif bucket_bytes == -3:  # reduce-scatter + all-gather over the flat span (SURVEY.md 8(e)'s direct exchange)
        from goi_hyperplane_amd.dist import allreduce_gradients_direct
        lone = torch.nn.Parameter(torch.zeros(7, 3))  # not part of the flat buffer, 21 elements: travels zero-padded
        lone.grad = torch.arange(21, dtype=torch.float32).view(7, 3) * (rank + 1)
        allreduce_gradients_direct(params + [lone], dist)
        assert torch.equal(lone.grad, torch.arange(21, dtype=torch.float32).view(7, 3) * sum(range(1, world + 1)))
    elif bucket_bytes == -4:  # the same, left in flight
        from goi_hyperplane_amd.dist import allreduce_gradients_direct
        held = [p.grad for p in params]
        h = allreduce_gradients_direct(params, dist, async_op=True)
        for p in params:
            p.grad = None
        h.wait()
        for p, g_ in zip(params, held):
            p.grad = g_
    else:
        allreduce_gradients(params, dist, bucket_bytes=bucket_bytes)
    # every exchange that summed other ranks' rows into the flat gradient buffer must have marked it as written: c10d
    # collectives do not bump the autograd version counter themselves, and the compiled binding's gradient-buffer pool would
    # otherwise hand the buffer out again as "rows of invisible Gaussians still hold zeros" (torch_binding.cpp, backward_ex)
    assert flat._version > version_before, (bucket_bytes, flat._version, version_before)
    q.put((rank, views[0], [x.numpy() for x in local], [p.grad.numpy() for p in params]))
    dist.barrier()
    dist.destroy_process_group()


def test_views_of_one_buffer_are_exchanged_as_one_tensor():
    from goi_hyperplane_amd.dist import coalesce_shared_storage
    flat = torch.arange(64 * 5, dtype=torch.float32)
    a, b, c = flat[0:30].view(10, 3), flat[64:64 + 40].view(10, 4), flat[128:128 + 160].view(10, 16)
    lone = torch.zeros(7)
    out = coalesce_shared_storage([a, lone, b, c])
    assert len(out) == 2 and any(o is lone for o in out)
    span = next(o for o in out if o is not lone)
    assert span.numel() == 128 + 160 and span.data_ptr() == flat.data_ptr()
    span.mul_(2)  # what an in-place all-reduce does: the views see it
    assert float(b[0, 0]) == 128.0 and float(c[9, 15]) == 2 * (128 + 159)
    # a mostly-foreign span is not coalesced
    big = torch.zeros(10000)
    assert len(coalesce_shared_storage([big[0:10], big[9000:9010]])) == 2


def _run(bucket_bytes, world=2):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, bucket_bytes, q)) for r in range(world)]
    for p in procs:
        p.start()
    got = [q.get(timeout=120) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    got.sort(key=lambda t: t[0])
    assert sorted(v for _, v, _, _ in got) == list(range(world))  # distinct views
    # float64 sum of the single-view oracle gradients; the ring adds in its own order: fp32 rounding only
    summed = [sum(np.asarray(g[2][i], np.float64) for g in got) for i in range(len(got[0][2]))]
    for r in range(world):
        for s, red in zip(summed, got[r][3]):
            scale = float(np.abs(s).max()) + 1e-30
            assert float(np.abs(red - s).max()) <= 1e-5 * scale  # (north_star asks for 1e-3)
    for r in range(1, world):  # every rank ends up with the same bits
        for a, b in zip(got[0][3], got[r][3]):
            np.testing.assert_array_equal(a, b)


def test_allreduce_equals_sum_of_single_view_gradients_per_tensor():
    _run(0)


def test_allreduce_equals_sum_of_single_view_gradients_bucketed():
    _run(1 << 12)


def test_exchange_in_flight_equals_sum_of_single_view_gradients():
    """allreduce_gradients_async: same sums as the blocking form, with the parameters' .grad released in between."""
    _run(-1)


def test_visible_rows_exchange_equals_sum_of_single_view_gradients():
    """allreduce_gradients_visible: only the rows of Gaussians visible to some rank are sent; the result is the plain sum."""
    _run(-2)
    _run(-2, world=3)


@pytest.mark.parametrize("world", [2, 3, 8])
def test_direct_exchange_equals_sum_of_single_view_gradients(world):
    """allreduce_gradients_direct (reduce-scatter + all-gather over the flat gradient span, in place): the plain sum, the
    same bits on every rank -- 2, 3 (the span does not divide: padded copy for the lone tensor) and 8 ranks (config 4)."""
    _run(-3, world=world)


def test_direct_exchange_in_flight_equals_sum_of_single_view_gradients():
    _run(-4, world=3)


def test_exchange_is_picked_by_the_link_model():
    from goi_hyperplane_amd.dist import pick_exchange
    assert pick_exchange(300e6, 8) == "direct" and pick_exchange(64e6, 4) == "direct"
    assert pick_exchange(300e6, 2) == "ring" and pick_exchange(300e6, 1) == "ring"


def test_visible_exchange_rejects_what_would_silently_lose_rows():
    """a non-contiguous per-Gaussian gradient (its reshape would be a copy) and a per_gaussian entry whose
    leading dimension is not P raise instead of returning a wrong sum (single process: the checks run before any collective
    that matters -- a 1-rank gloo group is enough)."""
    port = _free_port()
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    dist.init_process_group("gloo", rank=0, world_size=1)
    try:
        from goi_hyperplane_amd.dist import allreduce_gradients_visible
        P = 10
        vis = torch.zeros(P, dtype=torch.bool)
        vis[:4] = True
        a = torch.nn.Parameter(torch.zeros(P, 3))
        a.grad = torch.zeros(3, P).t()  # non-contiguous
        with pytest.raises(ValueError, match="contiguous"):
            allreduce_gradients_visible([a], vis, dist, dense_above=2.0)
        b = torch.nn.Parameter(torch.zeros(5, 3))
        b.grad = torch.ones(5, 3)
        with pytest.raises(ValueError, match="leading dimension"):
            allreduce_gradients_visible([b], vis, dist, dense_above=2.0, per_gaussian=[b])
        # a [P, ...] tensor that is NOT per Gaussian is reduced whole when the caller names the per-Gaussian ones
        c = torch.nn.Parameter(torch.zeros(P, 2))
        c.grad = torch.ones(P, 2)
        d = torch.nn.Parameter(torch.zeros(P, 3))
        d.grad = torch.zeros(P, 3)
        d.grad[:4] = 1.0
        assert allreduce_gradients_visible([c, d], vis, dist, dense_above=2.0, per_gaussian=[d], check_zero_rows=True) == 4
        d.grad[7] = 0.5  # a row no rank saw carries a gradient: the premise is violated
        with pytest.raises(RuntimeError, match="no rank saw"):
            allreduce_gradients_visible([d], vis, dist, dense_above=2.0, check_zero_rows=True)
    finally:
        dist.destroy_process_group()


def test_exchange_cost_model_matches_the_survey_figures():
    """ring 64 MB -> 0.73 ms, 300 MB -> 3.4 ms; direct 64 MB -> 0.10 ms, 300 MB -> 0.49 ms at 8 GPUs."""
    from goi_hyperplane_amd.dist import exchange_model_ms
    m64, m300 = exchange_model_ms(64.3e6, 8), exchange_model_ms(300e6, 8)
    assert abs(m64["ring"] - 0.73) < 0.02 and abs(m300["ring"] - 3.43) < 0.05
    assert abs(m64["direct"] - 0.105) < 0.01 and abs(m300["direct"] - 0.49) < 0.01
    assert exchange_model_ms(1e9, 1) == {"ring": 0.0, "direct": 0.0}


def test_eight_ranks_allreduce_equals_sum_of_eight_single_view_gradients():
    """BASELINE config 4's exchange step at its real width (8-view batch over 8 ranks): the
    all-reduced gradient on every rank == the sum of the 8 single-view oracle gradients."""
    _run(0, world=8)


def _worker_factored(rank, world, port, q, in_flight=False, direct=False):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from goi_hyperplane_amd.dist import allreduce_gradients_sh_factored
    from goi_hyperplane_amd.scene import make_camera, make_scene
    from oracle import oracle
    from tests.sh_basis_ref import C0, sh_grad_from_views
    sc = make_scene(300, S=10, sh_degree=3, seed=2, log_scale_mean=-2.4)
    cam = make_camera(64, 48, yaw=0.35 * rank, pitch=0.1 * rank)
    o = oracle.from_scene(sc, cam, threads=1)
    o.forward()
    HW = 64 * 48
    g = o.backward(np.full((3, 48, 64), 1 / HW, np.float32), np.full((10, 48, 64), 1 / HW, np.float32))
    names = ["means3D", "semantics", "opacity", "scales", "rotations"]  # every leaf but the SH tensors
    params = []
    for n in names:
        p = torch.nn.Parameter(torch.zeros(g[n].shape))
        p.grad = torch.tensor(g[n]).clone()
        params.append(p)
    dsh = torch.tensor(g["sh"])  # [P,16,3] of THIS view (what the factored mode does not form)
    # the factor the HIP backward leaves in sh_factored mode: the clamp-masked colour gradient = dL/dSH[0] / C0
    factor = dict(gcol=dsh[:, 0, :] / C0, campos=torch.tensor(np.asarray(cam.camera_center, np.float32)), degree=3, M=16)
    dc = torch.nn.Parameter(torch.zeros(dsh.shape[0], 1, 3))
    rest = torch.nn.Parameter(torch.zeros(dsh.shape[0], 15, 3))
    local = [p.grad.clone().numpy() for p in params]
    means = torch.tensor(np.asarray(sc.means3D, np.float32))
    versions = [p.grad._version for p in params]
    if in_flight:
        from goi_hyperplane_amd.dist import allreduce_gradients_sh_factored_async
        h = allreduce_gradients_sh_factored_async(params, (dc, rest), means, factor, dist, reconstruct=sh_grad_from_views,
                                                  direct=direct)
        h.wait()
        dc.grad, rest.grad = h.sh_grads[id(dc)], h.sh_grads[id(rest)]
    else:
        allreduce_gradients_sh_factored(params, (dc, rest), means, factor, dist, reconstruct=sh_grad_from_views, direct=direct)
    assert all(p.grad._version > v for p, v in zip(params, versions))  # (marked as written in place: see _worker)
    q.put((rank, local, dsh.numpy(), [p.grad.numpy() for p in params],
           torch.cat([dc.grad, rest.grad], dim=1).numpy()))
    dist.barrier()
    dist.destroy_process_group()


import pytest  # noqa: E402


@pytest.mark.parametrize("in_flight,direct", [(False, False), (True, False), (False, True), (True, True)])
def test_sh_factored_exchange_equals_sum_of_single_view_gradients(in_flight, direct):
    """SURVEY.md 8(e) with the SH gradient exchanged as factors (all-gather of the masked colour gradients + local
    reconstruction): same sums as the plain all-reduce, on every rank; blocking and in-flight forms."""
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker_factored, args=(r, world, port, q, in_flight, direct)) for r in range(world)]
    for p in procs:
        p.start()
    got = [q.get(timeout=120) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    got.sort(key=lambda t: t[0])
    summed = [a + b for a, b in zip(got[0][1], got[1][1])]
    dsh_sum = got[0][2] + got[1][2]
    assert float(np.abs(dsh_sum).max()) > 0
    for r in range(world):
        for s, red in zip(summed, got[r][3]):
            np.testing.assert_allclose(red, s, rtol=1e-6, atol=1e-7)
        np.testing.assert_allclose(got[r][4], dsh_sum, rtol=2e-5, atol=1e-6 * float(np.abs(dsh_sum).max()))
    np.testing.assert_array_equal(got[0][4], got[1][4])  # every rank holds the same bits
