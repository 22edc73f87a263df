"""Opt-in geometry cache (goi_raster_forward_reblend): with only the semantic features trainable, a camera that
was rendered before must render from the cached geometry state with the blend alone -- outputs and dL/dsemantics bit for bit
what the full forward gives for the current semantics -- and anything the key can see (an in-place update of the positions, a new
camera, an option switch) must miss."""
import pytest
import torch

from goi_hyperplane_amd import _C, _lib, rasterizer
from goi_hyperplane_amd.render import GaussianSet, PipelineParams, TorchCamera, render
from goi_hyperplane_amd.scene import make_camera, make_scene

pytestmark = pytest.mark.gpu


@pytest.fixture()
def cache():
    rasterizer.set_geometry_cache(4 << 30)
    base = rasterizer.geometry_cache_stats()
    yield base
    rasterizer.set_geometry_cache(0)
    assert rasterizer.geometry_cache_stats()["entries"] == 0


def _model(P=30000, S=16, seed=2):
    dev = torch.device("cuda", 0)
    sc = make_scene(P, S=S, sh_degree=3, seed=seed, extent=(2.0, 1.5, 1.0), log_scale_mean=-3.2)
    pc = GaussianSet.from_scene(sc, dev)
    for p in pc.parameters():  # the reference's semantic stage: arguments/__init__.py:85-90
        p.requires_grad_(False)
    pc._semantics.requires_grad_(True)
    return dev, pc


def _frame(cam, pc, g_sem, g_col):
    pc._semantics.grad = None
    out = render(cam, pc, PipelineParams(), torch.zeros(3, device=g_sem.device))
    ((out["semantics"] * g_sem).sum() + (out["render"] * g_col).sum()).backward()
    return {k: out[k].detach().clone() for k in ("render", "semantics", "depth", "alpha", "radii")}, pc._semantics.grad.clone()


@pytest.mark.parametrize("speculative", [True, False])
def test_cached_frames_are_bit_identical_and_hit(cache, speculative):
    dev, pc = _model()
    W, H = 400, 304
    cams = [TorchCamera(make_camera(W, H, fovx=1.0, yaw=0.04 * i), dev) for i in range(3)]
    gen = torch.Generator(device=dev).manual_seed(9)
    g_sem = torch.randn((16, H, W), device=dev, generator=gen)
    g_col = torch.randn((3, H, W), device=dev, generator=gen)
    rasterizer.set_forward_mode(speculative=speculative)
    try:
        s0 = rasterizer.geometry_cache_stats()
        first = [_frame(c, pc, g_sem, g_col) for c in cams]  # misses: fill the cache
        torch.cuda.synchronize()
        _C.poll_counts(dev, wait=True)
        s1 = rasterizer.geometry_cache_stats()
        assert s1["misses"] - s0["misses"] == 3 and s1["entries"] == 3
        with torch.no_grad():  # "training": the semantic features move, nothing else does
            pc._semantics.add_(0.05 * torch.randn(pc._semantics.shape, device=dev, generator=gen))
        cached = [_frame(c, pc, g_sem, g_col) for c in cams]
        s2 = rasterizer.geometry_cache_stats()
        assert s2["hits"] - s1["hits"] == 3, s2
        assert rasterizer.last_backward_kernel() == "semantics"
        rasterizer.set_geometry_cache(0)  # the same frames by the full forward
        full = [_frame(c, pc, g_sem, g_col) for c in cams]
        for (oc, gc), (of, gf), (o1, _g1) in zip(cached, full, first):
            for k in oc:
                assert torch.equal(oc[k], of[k]), k
            assert torch.equal(gc, gf)
            assert not torch.equal(oc["semantics"], o1["semantics"])  # the features did change
    finally:
        rasterizer.set_forward_mode(speculative=True)


def test_what_the_key_sees_misses(cache):
    dev, pc = _model(P=8000)
    W, H = 240, 160
    cam = TorchCamera(make_camera(W, H, fovx=1.0, yaw=0.1), dev)
    g_sem = torch.ones((16, H, W), device=dev)
    g_col = torch.ones((3, H, W), device=dev)
    _frame(cam, pc, g_sem, g_col)
    _C.poll_counts(dev, wait=True)
    base = rasterizer.geometry_cache_stats()
    _frame(cam, pc, g_sem, g_col)
    assert rasterizer.geometry_cache_stats()["hits"] == base["hits"] + 1
    with torch.no_grad():
        pc._xyz.add_(0.01)  # an in-place update bumps the version of the positions
    want = rasterizer.geometry_cache_stats()["misses"] + 1
    moved, _ = _frame(cam, pc, g_sem, g_col)
    assert rasterizer.geometry_cache_stats()["misses"] == want
    rasterizer.set_geometry_cache(0)
    ref, _ = _frame(cam, pc, g_sem, g_col)
    assert torch.equal(moved["render"], ref["render"])  # and the frame is the moved scene's
    rasterizer.set_geometry_cache(4 << 30)
    _frame(cam, pc, g_sem, g_col)
    _C.poll_counts(dev, wait=True)
    cam2 = TorchCamera(make_camera(W, H, fovx=1.0, yaw=0.1), dev)  # same pose, new tensors: a miss, by design
    m = rasterizer.geometry_cache_stats()["misses"]
    _frame(cam2, pc, g_sem, g_col)
    assert rasterizer.geometry_cache_stats()["misses"] == m + 1
    try:
        _lib.set_option("cull_variant", 0)
        m = rasterizer.geometry_cache_stats()["misses"]
        _frame(cam, pc, g_sem, g_col)
        assert rasterizer.geometry_cache_stats()["misses"] == m + 1
    finally:
        _lib.set_option("cull_variant", 2)


def test_not_eligible_when_geometry_is_trainable_and_bounded_by_bytes(cache):
    dev, pc = _model(P=8000)
    W, H = 240, 160
    cams = [TorchCamera(make_camera(W, H, fovx=1.0, yaw=0.05 * i), dev) for i in range(4)]
    g_sem = torch.ones((16, H, W), device=dev)
    g_col = torch.ones((3, H, W), device=dev)
    pc._opacity.requires_grad_(True)
    before = rasterizer.geometry_cache_stats()
    _frame(cams[0], pc, g_sem, g_col)
    _frame(cams[0], pc, g_sem, g_col)
    after = rasterizer.geometry_cache_stats()
    assert after["hits"] == before["hits"] and after["misses"] == before["misses"] and after["entries"] == 0
    pc._opacity.requires_grad_(False)
    pc._opacity.grad = None
    _frame(cams[0], pc, g_sem, g_col)
    one = rasterizer.geometry_cache_stats()["bytes"]
    assert one > 0
    rasterizer.set_geometry_cache(int(2.5 * one))  # room for two cameras
    for c in cams:
        _frame(c, pc, g_sem, g_col)
    st = rasterizer.geometry_cache_stats()
    assert st["entries"] == 2 and st["bytes"] <= int(2.5 * one) and st["evictions"] >= 2


def test_hits_from_other_streams_than_the_one_that_filled_the_cache(cache):
    """render_views puts view i on stream i % streams: with three cameras on two streams, the second pass renders camera 1 on the
    stream that rendered it before but cameras 0 and 2 swap streams relative to a one-stream fill -- a hit must wait for the
    stream that filled the workspaces."""
    from goi_hyperplane_amd.render import render_views
    dev, pc = _model(P=20000)
    W, H = 320, 208
    cams = [TorchCamera(make_camera(W, H, fovx=1.0, yaw=0.05 * i), dev) for i in range(3)]
    bg = torch.zeros(3, device=dev)
    gen = torch.Generator(device=dev).manual_seed(3)
    g_sem = torch.randn((16, H, W), device=dev, generator=gen)
    loss = lambda i, o: (o["semantics"] * g_sem).sum()  # noqa: E731
    for c in cams:  # fill on the caller's stream
        render(c, pc, PipelineParams(), bg)
    _C.poll_counts(dev, wait=True)
    with torch.no_grad():
        pc._semantics.mul_(1.01)
    h0 = rasterizer.geometry_cache_stats()["hits"]
    pc._semantics.grad = None
    outs = render_views(cams, pc, PipelineParams(), bg, loss_fn=loss, streams=2)
    torch.cuda.synchronize()
    assert rasterizer.geometry_cache_stats()["hits"] == h0 + 3
    g_cached = pc._semantics.grad.clone()
    rasterizer.set_geometry_cache(0)
    pc._semantics.grad = None
    ref = []
    for i, c in enumerate(cams):
        o = render(c, pc, PipelineParams(), bg)
        loss(i, o).backward()
        ref.append(o["semantics"].detach().clone())
    for r, o in zip(ref, outs):
        assert torch.equal(r, o["semantics"].detach())
    scale = float(pc._semantics.grad.abs().max())
    assert float((pc._semantics.grad - g_cached).abs().max()) <= 2e-6 * scale  # three views: the order of the additions may differ


def test_a_truncated_cached_frame_is_never_reblended(cache):
    """a speculative frame that overflowed and whose overflow was only found lazily (nobody read the
    count) holds a cut-off tile list in its binning workspace.  The cache must not serve it: the next frame of that camera
    drops the entry and renders in full, bit-identical to the uncached frame."""
    import warnings
    dev, pc = _model(P=8000)
    W, H = 240, 160
    cam = TorchCamera(make_camera(W, H, fovx=1.0, yaw=0.07), dev)
    g_sem = torch.ones((16, H, W), device=dev)
    g_col = torch.ones((3, H, W), device=dev)
    rasterizer.set_geometry_cache(0)
    ref, gref = _frame(cam, pc, g_sem, g_col)
    rasterizer.set_geometry_cache(4 << 30)
    try:
        with warnings.catch_warnings():
            warnings.simplefilter("ignore", _C.RasterOverflowWarning)
            rasterizer.set_forward_mode(speculative=True, capacity=3000)  # far below the ~60 k instances of this view
            bad, gbad = _frame(cam, pc, g_sem, g_col)                     # truncated, never read, cached as pending
            torch.cuda.synchronize()
            assert not torch.equal(bad["render"], ref["render"]) and float(gbad.abs().max()) == 0.0
            rasterizer.set_forward_mode(capacity=None)
            s0 = rasterizer.geometry_cache_stats()
            again, gagain = _frame(cam, pc, g_sem, g_col)  # must NOT be a hit on the truncated lists
            s1 = rasterizer.geometry_cache_stats()
        assert s1["hits"] == s0["hits"] and s1["misses"] == s0["misses"] + 1
        for k in ref:
            assert torch.equal(again[k], ref[k]), k
        assert torch.equal(gagain, gref)
        _C.poll_counts(dev, wait=True)
        hit, ghit = _frame(cam, pc, g_sem, g_col)  # and the repaired entry is a hit with the right bits
        assert rasterizer.geometry_cache_stats()["hits"] == s1["hits"] + 1
        assert torch.equal(hit["render"], ref["render"]) and torch.equal(ghit, gref)
    finally:
        rasterizer.set_forward_mode(speculative=True, capacity=None)


def test_recycled_addresses_of_freed_camera_tensors_do_not_alias(cache):
    """the key identifies tensors by (address, version, shape).  A viewer builds its camera tensors per
    frame (gui/gs_renderer.py MiniCam) and frees them after the call; the caching allocator then hands the next camera's
    tensors the SAME addresses with version 0.  The entry keeps the keyed tensors alive, so a different camera can never
    inherit a stale entry's identity."""
    dev, pc = _model(P=8000)
    W, H = 240, 160
    g_sem = torch.ones((16, H, W), device=dev)
    g_col = torch.ones((3, H, W), device=dev)
    outs = []
    for i in range(6):  # every camera is a temporary: dropped right after its frame
        cam = TorchCamera(make_camera(W, H, fovx=1.0, yaw=0.15 * i - 0.4), dev)
        o, _g = _frame(cam, pc, g_sem, g_col)
        outs.append(o["render"])
        del cam
    st = rasterizer.geometry_cache_stats()
    assert st["hits"] == cache["hits"], "a new camera hit a stale entry"
    rasterizer.set_geometry_cache(0)
    for i in range(6):
        cam = TorchCamera(make_camera(W, H, fovx=1.0, yaw=0.15 * i - 0.4), dev)
        o, _g = _frame(cam, pc, g_sem, g_col)
        assert torch.equal(o["render"], outs[i]), i
