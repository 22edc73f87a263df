"""world_size-2 gloo tests (CPU) of the multi-GPU path: view sharding and the gradient all-reduce
(all-reduced gradient == sum of the single-view gradients)."""
import os
import socket

import numpy as np
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from goi_hyperplane_amd.dist import allreduce_gradients, shard_views


def test_shard_views_is_a_partition():
    for world in (1, 2, 3, 8):
        for epoch in (0, 1):
            shards = [shard_views(37, r, world, epoch) for r in range(world)]
            flat = sorted(i for s in shards for i in s)
            assert flat == list(range(37))
    assert shard_views(10, 0, 2, epoch=0) != shard_views(10, 0, 2, epoch=1)


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
    cam = make_camera(64, 48, yaw=0.1 * views[0])
    o = oracle.from_scene(sc, cam, threads=1)
    o.forward()
    HW = 64 * 48
    g = o.backward(np.full((3, 48, 64), 1 / HW, np.float32), np.full((10, 48, 64), 1 / HW, np.float32))
    names = ["means3D", "sh", "semantics", "opacity", "scales", "rotations"]
    params = []
    for n in names:
        p = torch.nn.Parameter(torch.zeros(g[n].shape))
        p.grad = torch.tensor(g[n])
        params.append(p)
    local = [p.grad.clone() for p in params]
    allreduce_gradients(params, dist, bucket_bytes=bucket_bytes)
    q.put((rank, views[0], [x.numpy() for x in local], [p.grad.numpy() for p in params]))
    dist.barrier()
    dist.destroy_process_group()


def _run(bucket_bytes):
    world = 2
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
    got.sort()
    assert sorted(v for _, v, _, _ in got) == [0, 1]  # distinct views
    summed = [a + b for a, b in zip(got[0][2], got[1][2])]
    for r in range(world):
        for s, red in zip(summed, got[r][3]):
            np.testing.assert_allclose(red, s, rtol=1e-6, atol=1e-7)


def test_allreduce_equals_sum_of_single_view_gradients_per_tensor():
    _run(0)


def test_allreduce_equals_sum_of_single_view_gradients_bucketed():
    _run(1 << 12)
