"""The forward without the host round trip (goi_raster_forward_async, SURVEY.md section 7 "no host sync"; the
reference blocks at cuda_rasterizer/rasterizer_impl.cu:285).  A speculative frame is sized from a capacity guess and
returns `num_rendered` lazily; these tests pin down that it changes no result:

  * capacity >= count (any capacity): outputs, sorted lists, ranges and every gradient are BIT-identical to the exact,
    synchronous path;
  * capacity < count (forced): reading the count before the outputs redoes the frame in place -> bit-identical again,
    forward and backward; not reading it leaves a truncated IMAGE but a backward that writes ZERO gradients on the device
    (a skipped view; FusedAdam.step(skip_if=...) leaves the parameters untouched), reported at the next forward;
  * the default policy: the first three frames of a scene exact, later frames speculative, nothing waits."""
import warnings

import numpy as np
import pytest
import torch

from goi_hyperplane_amd.scene import make_camera, make_scene

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def dev():
    assert torch.cuda.is_available(), "GPU tests need a ROCm device"
    from goi_hyperplane_amd import _lib
    _lib.load()
    return torch.device("cuda:0")


@pytest.fixture(autouse=True)
def _restore_mode():
    from goi_hyperplane_amd import _C
    yield
    _C.poll_counts(wait=True)
    _C.set_forward_mode(speculative=True, headroom=2.0, capacity=None, on_overflow="warn", max_ahead=64,
                        inference_speculative=False, min_history=3)


def _args(sc, cam, dev, bg):
    from goi_hyperplane_amd.render import TorchCamera
    tc = TorchCamera(cam, dev)
    t = lambda a: torch.tensor(a, device=dev)  # noqa: E731
    return (torch.tensor(bg, device=dev), t(sc.means3D), torch.Tensor([]), t(sc.semantics), t(sc.opacities), t(sc.scales),
            t(sc.rotations), 1.0, torch.Tensor([]), tc.world_view_transform, tc.full_proj_transform, cam.tanfovx,
            cam.tanfovy, cam.image_height, cam.image_width, t(sc.shs), sc.sh_degree, tc.camera_center, False, False)


def _raw(dev, sc, cam, bg, **mode):
    """raw op forward (+ workspace views) under a forward mode; returns (n, outputs, views)"""
    from goi_hyperplane_amd import _C
    _C.set_forward_mode(**mode)
    n, color, sem, depth, alpha, radii, geom, binning, img = _C.rasterize_gaussians(*_args(sc, cam, dev, bg))
    return n, (color, sem, depth, alpha, radii), (geom, binning, img)


@pytest.mark.parametrize("P,S,W,H,mu", [(3000, 16, 160, 120, -2.8), (20000, 16, 400, 300, -3.5), (800, 10, 123, 77, -1.6),
                                         (1500, 4, 3840, 2160, -2.0)])
def test_speculative_frames_are_bit_identical_to_exact_ones(dev, P, S, W, H, mu):
    from goi_hyperplane_amd import _C
    sc = make_scene(P, S=S, seed=3, log_scale_mean=mu)
    cam = make_camera(W, H, yaw=0.2, pitch=-0.1)
    bg = np.array([0.1, 0.3, 0.6], np.float32)
    n0, o0, w0 = _raw(dev, sc, cam, bg, speculative=False)
    assert isinstance(n0, int) and n0 > 0
    v0 = _C.debug_views(P, W, H, n0, *w0)
    for cap in (n0, n0 + 1, 2 * n0 + 12345, 7 * n0):  # exactly full, barely, the default headroom, far too large
        n1, o1, w1 = _raw(dev, sc, cam, bg, speculative=True, capacity=cap)
        assert isinstance(n1, _C.LazyCount) and not n1.resolved
        for a, b in zip(o0, o1):
            assert torch.equal(a, b)
        assert int(n1) == n0 and not n1.overflowed and n1.layout == cap
        v1 = _C.debug_views(P, W, H, n1, *w1)
        for k in v0:
            assert torch.equal(v0[k], v1[k]), (cap, k)


@pytest.mark.parametrize("frac", [0.05, 0.5, 0.999])
def test_overflow_is_redone_bit_identically_when_the_count_is_read_first(dev, frac):
    from goi_hyperplane_amd import _C
    P, S, W, H = 4000, 16, 208, 160
    sc = make_scene(P, S=S, seed=8, log_scale_mean=-2.7)
    cam = make_camera(W, H, yaw=-0.1)
    bg = np.array([0.2, 0.1, 0.0], np.float32)
    n0, o0, w0 = _raw(dev, sc, cam, bg, speculative=False)
    v0 = _C.debug_views(P, W, H, n0, *w0)
    before = dict(_C.SPECULATION_STATS)
    cap = max(1, int(frac * n0))
    n1, o1, w1 = _raw(dev, sc, cam, bg, speculative=True, capacity=cap)
    torch.cuda.synchronize()
    # the truncated frame differs (that is what an overflow is) but is finite and was memory-safe
    assert not torch.equal(o0[0], o1[0]) and all(torch.isfinite(x.float()).all() for x in o1)
    assert int(n1) == n0 and n1.overflowed and n1.redone and n1.layout == n0  # reading the count repaired the frame
    for a, b in zip(o0, o1):
        assert torch.equal(a, b)
    v1 = _C.debug_views(P, W, H, n1, w1[0], n1.binning, w1[2])
    for k in v0:
        assert torch.equal(v0[k], v1[k]), k
    after = _C.SPECULATION_STATS
    assert after["overflows"] == before["overflows"] + 1 and after["redone"] == before["redone"] + 1


@pytest.mark.parametrize("frac", [0.3, 0.9, 2.0])
def test_big_rectangles_through_the_queue_survive_overflow_and_redo(dev, frac):
    """Small tile grid (<= 2048 tiles) and frame-filling Gaussians: emit_k hands every rectangle above 128 tiles to emit_big_k
    through its device-side queue (binning.hip, round 5).  Speculative frames with enough capacity equal the exact frame bit for
    bit, lists included; a truncated one is repaired by the redo (which emits AGAIN from the same workspace: the queue counter
    must have been put back to zero by the frame's own tile_ranges_hist_k)."""
    from goi_hyperplane_amd import _C
    P, S, W, H = 300, 16, 400, 300
    sc = make_scene(P, S=S, seed=21, log_scale_mean=-0.6)  # most Gaussians cover most of the 475 tiles
    cam = make_camera(W, H, yaw=0.1)
    bg = np.array([0.0, 0.2, 0.1], np.float32)
    n0, o0, w0 = _raw(dev, sc, cam, bg, speculative=False)
    assert n0 > 128 * 100  # (rectangles of several hundred tiles: the queue is what wrote most of the list)
    v0 = _C.debug_views(P, W, H, n0, *w0)
    cap = max(1, int(frac * n0))
    n1, o1, w1 = _raw(dev, sc, cam, bg, speculative=True, capacity=cap)
    torch.cuda.synchronize()
    assert int(n1) == n0 and n1.overflowed == (cap < n0)
    for a, b in zip(o0, o1):
        assert torch.equal(a, b)
    v1 = _C.debug_views(P, W, H, n1, w1[0], n1.binning, w1[2])
    for k in v0:
        assert torch.equal(v0[k], v1[k]), k


def test_adaptive_sort_tiles_agree_with_the_three_kernel_sort_above_two_million_gaussians(dev):
    """Above 2 M keys of CAPACITY the onesweep passes choose their tile shape from the device-side count (scan_sort.hip, round 5:
    512 x 2 .. 16 keys).  2.3 M Gaussians of which a 320 x 240 view lists a small fraction: the exact frame (count known), a
    speculative frame (count on the device, generous capacity) and the histogram / scan / scatter sort (sort_variant 0, no
    look-back at all) must produce the same sorted lists bit for bit; a view that sees NOTHING must run (count 0 everywhere)."""
    from goi_hyperplane_amd import _C, _lib
    P, S, W, H = 2_300_000, 4, 320, 240
    sc = make_scene(P, S=S, sh_degree=0, seed=31, log_scale_mean=-4.2)
    bg = np.zeros(3, np.float32)
    cam = make_camera(W, H, yaw=0.1)
    n0, o0, w0 = _raw(dev, sc, cam, bg, speculative=False)
    assert isinstance(n0, int) and 0 < n0
    v0 = _C.debug_views(P, W, H, n0, *w0)
    n1, o1, w1 = _raw(dev, sc, cam, bg, speculative=True, capacity=3 * n0 + 100_000)
    for a, b in zip(o0, o1):
        assert torch.equal(a, b)
    v1 = _C.debug_views(P, W, H, n1, *w1)
    _lib.set_option("sort_variant", 0)
    try:
        n2, o2, w2 = _raw(dev, sc, cam, bg, speculative=False)
        v2 = _C.debug_views(P, W, H, n2, *w2)
    finally:
        _lib.set_option("sort_variant", 1)
    _lib.set_option("sort_lookback", 0)  # the chained decoupled look-back instead of the grouped one
    try:
        n4, o4, w4 = _raw(dev, sc, cam, bg, speculative=False)
        v4 = _C.debug_views(P, W, H, n4, *w4)
    finally:
        _lib.set_option("sort_lookback", 1)
    assert n2 == n0 and n4 == n0
    for k in ("point_list", "ranges", "n_contrib"):
        assert torch.equal(v0[k], v1[k]), k
        assert torch.equal(v0[k], v2[k]), k
        assert torch.equal(v0[k], v4[k]), k
    # ... and a view that lists NOTHING: the scene reflected through the camera centre lies behind the near plane
    from goi_hyperplane_amd.render import TorchCamera
    c = TorchCamera(cam, dev).camera_center.detach().cpu().numpy().reshape(1, 3)
    sc.means3D[:] = (2.0 * c - sc.means3D).astype(np.float32)
    n3, o3, _w3 = _raw(dev, sc, cam, bg, speculative=True, capacity=100_000)
    torch.cuda.synchronize()
    assert int(n3) == 0 and not n3.overflowed and float(o3[3].abs().max()) == 0.0  # (alpha: nothing was composited)
    _C.set_forward_mode(speculative=True, capacity=None)


def test_overflow_through_autograd_read_before_use_gives_exact_gradients(dev):
    """The autograd path: an overflowed frame whose count is read (LazyCount.resolve, here through
    rasterizer.last_num_rendered) before the loss is formed has the exact path's outputs AND gradients."""
    from goi_hyperplane_amd import _C, rasterizer
    from goi_hyperplane_amd.render import GaussianSet, PipelineParams, TorchCamera, render
    sc = make_scene(5000, S=16, seed=12, log_scale_mean=-2.9)
    cam = TorchCamera(make_camera(208, 160, yaw=0.05), dev)
    pc = GaussianSet.from_scene(sc, dev)
    bg = torch.zeros(3, device=dev)
    g = torch.Generator(device=dev).manual_seed(4)
    up = [torch.randn(s, device=dev, generator=g) for s in ((3, 160, 208), (16, 160, 208))]

    def step(read_count, **mode):
        _C.set_forward_mode(**mode)
        for p in pc.parameters():
            p.grad = None
        out = render(cam, pc, PipelineParams(), bg)
        n = rasterizer.last_num_rendered()
        if read_count:
            n = int(n)
        torch.autograd.backward((out["render"], out["semantics"]), up)
        return n, {k: out[k].detach().clone() for k in ("render", "semantics", "depth", "alpha")}, \
            {k: p.grad.clone() for k, p in pc.named_parameters()}, out["viewspace_points"].grad.clone()

    n0, o0, g0, v0 = step(True, speculative=False)
    n1, o1, g1, v1 = step(True, speculative=True, capacity=n0 // 3)  # overflow, repaired before use
    assert n1 == n0
    for k in o0:
        assert torch.equal(o0[k], o1[k]), k
    for k in g0:
        assert torch.equal(g0[k], g1[k]), k
    assert torch.equal(v0, v1)
    n2, o2, g2, v2 = step(False, speculative=True, capacity=3 * n0)  # fits: nothing is ever read back in time
    for k in o0:
        assert torch.equal(o0[k], o2[k]), k
    for k in g0:
        assert torch.equal(g0[k], g2[k]), k
    assert torch.equal(v0, v2)


@pytest.mark.parametrize("path", ["rows", "rows_fp32", "atomic", "semantics_only"])
def test_unread_overflow_never_trains_anything(dev, path):
    """VERDICT r02 item 2 / ADVICE r02: a truncated frame must never reach the optimiser unnoticed.  Overflow is forced
    and the count is NEVER read: every gradient the backward produces is exactly zero (decided on the device: emit's
    COUNTER_OVF word), the fused Adam step guarded by rasterizer.truncated_flag() leaves parameters and moments bit for
    bit where they were, and an unguarded torch.optim.Adam sees a zero gradient.  The same frame with room to spare
    trains normally.  All three backward paths (atomic-free rows with either flush, float-atomic per tile,
    feature-gradient-only)."""
    from goi_hyperplane_amd import _C, _lib, rasterizer
    from goi_hyperplane_amd.optim import FusedAdam
    from goi_hyperplane_amd.render import GaussianSet, PipelineParams, TorchCamera, render
    sc = make_scene(4000, S=16, seed=5, log_scale_mean=-2.7)
    cam = TorchCamera(make_camera(160, 128), dev)
    pc = GaussianSet.from_scene(sc, dev)
    bg = torch.zeros(3, device=dev)
    sem_only = path == "semantics_only"
    _lib.set_option("bwd_variant", {"rows": 0, "rows_fp32": 2, "atomic": 1, "semantics_only": 0}[path])
    rasterizer.set_backward_mode(semantics_only=sem_only)
    if sem_only:
        for p in pc.parameters():
            p.requires_grad_(False)
        pc._semantics.requires_grad_(True)
    trainable = [p for p in pc.parameters() if p.requires_grad]
    opt = FusedAdam([{"params": trainable, "lr": 1e-2}], lr=1e-2)
    try:
        def step(capacity, guarded=True):
            _C.set_forward_mode(speculative=True, capacity=capacity)
            opt.zero_grad(set_to_none=True)
            out = render(cam, pc, PipelineParams(), bg)
            flag = rasterizer.truncated_flag()
            loss = out["semantics"].sum() if sem_only else out["render"].sum() + out["semantics"].sum() + out["depth"].sum()
            loss.backward()
            grads = [p.grad.clone() for p in trainable]
            opt.step(skip_if=flag if guarded else None)
            return flag, grads, out["viewspace_points"].grad

        with warnings.catch_warnings():
            warnings.simplefilter("ignore", _C.RasterOverflowWarning)
            flag, grads, _ = step(capacity=1 << 20)  # fits: a normal training step (also creates the Adam state)
            torch.cuda.synchronize()
            assert int(flag.item()) == 0 and any(float(g.abs().max()) > 0 for g in grads)
            before = [p.detach().clone() for p in trainable]
            moments = [(opt.state[p]["exp_avg"].clone(), opt.state[p]["exp_avg_sq"].clone()) for p in trainable]
            flag, grads, vs = step(capacity=2000)  # TRUNCATED, never read
            torch.cuda.synchronize()
            assert int(flag.item()) == 1
            for g in grads:
                assert float(g.abs().max()) == 0.0, "a truncated frame produced a non-zero gradient"
            assert vs is None or float(vs.abs().max()) == 0.0
            for p, b, (m, v) in zip(trainable, before, moments):
                assert torch.equal(p.detach(), b), "the guarded optimiser step moved a parameter"
                assert torch.equal(opt.state[p]["exp_avg"], m) and torch.equal(opt.state[p]["exp_avg_sq"], v)
            # the same view with room: trains again, and the flag is back to zero
            flag, grads, _ = step(capacity=1 << 20)
            torch.cuda.synchronize()
            assert int(flag.item()) == 0 and any(float(g.abs().max()) > 0 for g in grads)
            assert any(not torch.equal(p.detach(), b) for p, b in zip(trainable, before))
        assert _C.SPECULATION_STATS["skipped_views"] >= 1
    finally:
        _lib.set_option("bwd_variant", 0)
        rasterizer.set_backward_mode(semantics_only="auto")


def test_unread_overflow_is_reported_at_a_later_forward_as_a_skipped_view(dev):
    from goi_hyperplane_amd import _C
    from goi_hyperplane_amd.render import GaussianSet, PipelineParams, TorchCamera, render
    sc = make_scene(4000, S=16, seed=5, log_scale_mean=-2.7)
    cam = TorchCamera(make_camera(160, 128), dev)
    pc = GaussianSet.from_scene(sc, dev)
    bg = torch.zeros(3, device=dev)
    _C.set_forward_mode(speculative=True, capacity=2000)
    with pytest.warns(_C.RasterOverflowWarning, match="ZERO gradients"):
        out = render(cam, pc, PipelineParams(), bg)
        (out["render"].sum() + out["semantics"].sum()).backward()  # backward of the truncated frame: zeros
        torch.cuda.synchronize()
        assert all(float(p.grad.abs().max()) == 0.0 for p in pc.parameters())
        _C.set_forward_mode(capacity=None)
        # found by whichever looks first without waiting: the backward's free look, or the poll of the next forward
        render(cam, pc, PipelineParams(), bg)
    # ... and the policy has learned: the next frames are speculative with room to spare, no warning, no wait
    with warnings.catch_warnings():
        warnings.simplefilter("error")
        waits = _C.SPECULATION_STATS["waits"]
        outs = [render(cam, pc, PipelineParams(), bg) for _ in range(5)]
        assert _C.SPECULATION_STATS["waits"] == waits
    torch.cuda.synchronize()
    _C.set_forward_mode(speculative=False)
    ref = render(cam, pc, PipelineParams(), bg)
    assert all(torch.equal(o["render"], ref["render"]) for o in outs)
    # raise instead of warn
    _C.set_forward_mode(speculative=True, capacity=2000, on_overflow="raise")
    render(cam, pc, PipelineParams(), bg)
    torch.cuda.synchronize()
    _C.set_forward_mode(capacity=None)
    with pytest.raises(_C.RasterOverflowError):
        render(cam, pc, PipelineParams(), bg)


def test_default_policy_first_frames_exact_then_nothing_waits(dev):
    from goi_hyperplane_amd import _C
    from goi_hyperplane_amd.render import GaussianSet, PipelineParams, TorchCamera, render
    _C._SPEC.clear()  # a fresh process
    sc = make_scene(30000, S=16, seed=2, log_scale_mean=-3.4)
    pc = GaussianSet.from_scene(sc, dev)
    cams = [TorchCamera(make_camera(400, 304, yaw=0.03 * i), dev) for i in range(8)]
    bg = torch.zeros(3, device=dev)
    s0 = dict(_C.SPECULATION_STATS)
    for i in range(24):
        for p in pc.parameters():
            p.grad = None
        out = render(cams[i % 8], pc, PipelineParams(), bg)
        (out["render"].mean() + out["semantics"].mean()).backward()
    torch.cuda.synchronize()
    _C.poll_counts(wait=True)
    s1 = _C.SPECULATION_STATS
    assert s1["exact_frames"] - s0["exact_frames"] == 3  # (min_history: three counts teach the capacity policy)
    assert s1["speculative_frames"] - s0["speculative_frames"] == 21
    assert s1["overflows"] == s0["overflows"]
    assert len(_C._SPEC[dev.index]["pending"]) == 0


@pytest.mark.parametrize("binding", ["ctypes", "compiled"])
def test_two_host_threads_on_two_streams(dev, binding):
    """concurrent calls on different streams are legal for a drop-in.  Two Python threads, each with its
    own stream and scene, render + backpropagate concurrently (speculative forwards: tickets, the capacity policy and the
    stage-profile state are shared process-wide behind locks); every result equals the single-threaded one bit for bit."""
    import threading
    from goi_hyperplane_amd import _C
    from goi_hyperplane_amd.render import GaussianSet, PipelineParams, TorchCamera, render
    _C.set_binding(binding)
    try:
        jobs = []
        for seed, (P, W, H) in enumerate([(20000, 320, 208), (12000, 256, 192)]):
            sc = make_scene(P, S=16, seed=40 + seed, log_scale_mean=-3.2)
            jobs.append((GaussianSet.from_scene(sc, dev), [TorchCamera(make_camera(W, H, yaw=0.05 * i), dev) for i in range(4)]))
        bg = torch.zeros(3, device=dev)

        def run(job, stream, out, n=12):
            pc, cams = job
            res = []
            with torch.cuda.stream(stream):
                for i in range(n):
                    for p in pc.parameters():
                        p.grad = None
                    o = render(cams[i % 4], pc, PipelineParams(), bg)
                    (o["render"].sum() + o["semantics"].sum()).backward()
                    res.append((o["render"].detach().clone(), pc._xyz.grad.clone(), pc._semantics.grad.clone()))
            stream.synchronize()
            out.append(res)

        serial = []
        for job in jobs:
            run(job, torch.cuda.current_stream(dev), serial)
        torch.cuda.synchronize()
        streams = [torch.cuda.Stream(device=dev) for _ in jobs]
        for s in streams:
            s.wait_stream(torch.cuda.current_stream(dev))
        outs = [[] for _ in jobs]
        errors = []

        def guarded(*a):
            try:
                run(*a)
            except Exception as ex:  # noqa: BLE001
                errors.append(ex)
        threads = [threading.Thread(target=guarded, args=(job, s, o)) for job, s, o in zip(jobs, streams, outs)]
        for t in threads:
            t.start()
        for t in threads:
            t.join()
        assert not errors, errors
        for want, got in zip(serial, outs):
            for a, b in zip(want, got[0]):
                for x, y in zip(a, b):
                    assert torch.equal(x, y)
    finally:
        _C.poll_counts(wait=True)
        _C.set_binding("compiled")


def _fuzz_cases(n=36, seed=77):
    rng = np.random.default_rng(seed)
    out = []
    for k in range(n):
        P = int(rng.choice([1, 7, 64, 65, 300, 1500, 4000, 20000]))
        W, H = int(rng.integers(17, 260)), int(rng.integers(17, 200))
        S = int(rng.choice([1, 3, 4, 10, 16, 17, 32]))
        mu = float(rng.uniform(-4.0, -1.0))
        frac = float(rng.choice([0.02, 0.3, 0.9, 1.0, 1.0001, 1.5, 4.0]))  # capacity as a fraction of the true count
        out.append((k, P, S, W, H, mu, frac))
    return out


@pytest.mark.parametrize("k,P,S,W,H,mu,frac", _fuzz_cases())
def test_random_capacities_forward_and_backward_are_bit_identical_to_exact(dev, k, P, S, W, H, mu, frac):
    """Seeded sweep over shapes and over capacities below, at and above the true count: a speculative frame whose count is
    read before its outputs are used (redo on overflow) equals the exact frame bit for bit, outputs and gradients; so does
    one that fits and is never read."""
    from goi_hyperplane_amd import _C, rasterizer
    from goi_hyperplane_amd.render import GaussianSet, PipelineParams, TorchCamera, render
    sc = make_scene(P, S=S, sh_degree=int(k % 4), seed=300 + k, log_scale_mean=mu)
    cam = TorchCamera(make_camera(W, H, yaw=0.02 * k - 0.3, pitch=0.01 * (k % 7)), dev)
    pc = GaussianSet.from_scene(sc, dev)
    bg = torch.tensor([0.1, 0.2, 0.3], device=dev)
    g = torch.Generator(device=dev).manual_seed(k)
    ups = [torch.randn(s, device=dev, generator=g) for s in ((3, H, W), (S, H, W), (1, H, W), (1, H, W))]

    def step(read, **mode):
        _C.set_forward_mode(**mode)
        for p in pc.parameters():
            p.grad = None
        out = render(cam, pc, PipelineParams(), bg)
        n = rasterizer.last_num_rendered()
        if read:
            n = int(n)
        torch.autograd.backward((out["render"], out["semantics"], out["depth"], out["alpha"]), ups)
        return n, [out[x].detach().clone() for x in ("render", "semantics", "depth", "alpha", "radii")], \
            [p.grad.clone() for p in pc.parameters()] + [out["viewspace_points"].grad.clone()]

    n0, o0, g0 = step(True, speculative=False)
    cap = max(1, int(np.ceil(frac * max(n0, 1))))
    n1, o1, g1 = step(True, speculative=True, capacity=cap)
    assert n1 == n0
    for a, b in zip(o0 + g0, o1 + g1):
        assert torch.equal(a, b), (k, cap, n0)
    if cap >= n0:  # fits: identical even when nobody ever reads the count
        n2, o2, g2 = step(False, speculative=True, capacity=cap)
        for a, b in zip(o0 + g0, o2 + g2):
            assert torch.equal(a, b), (k, cap, n0)
        assert int(n2) == n0


def test_overflow_under_no_grad_is_redone_when_the_count_is_read_right_away(dev):
    """Inference (torch.no_grad): nothing but the LazyCount keeps the frame's workspaces once the operator has returned;
    the newest pending frames hold on to them, so reading the count right after the render still repairs an overflow."""
    from goi_hyperplane_amd import _C, rasterizer
    from goi_hyperplane_amd.render import GaussianSet, PipelineParams, TorchCamera, render
    sc = make_scene(6000, S=16, seed=15, log_scale_mean=-2.9)
    cam = TorchCamera(make_camera(224, 160, yaw=0.1), dev)
    pc = GaussianSet.from_scene(sc, dev)
    bg = torch.zeros(3, device=dev)
    with torch.no_grad():
        _C.set_forward_mode(speculative=False)
        ref = render(cam, pc, PipelineParams(), bg, scaling_modifier=2.5)
        n_ref = int(rasterizer.last_num_rendered())
        _C.set_forward_mode(speculative=True, capacity=n_ref // 4)
        out = render(cam, pc, PipelineParams(), bg, scaling_modifier=2.5)
        n = rasterizer.last_num_rendered()
        assert int(n) == n_ref and n.overflowed and n.redone
        for k in ("render", "semantics", "depth", "alpha"):
            assert torch.equal(out[k], ref[k]), k
        # outputs dropped before the count is read: nothing to repair, no error
        render(cam, pc, PipelineParams(), bg, scaling_modifier=2.5)
        n2 = rasterizer.last_num_rendered()
        assert int(n2) == n_ref and n2.overflowed and not n2.redone


def test_image_only_frames_are_exact_by_default(dev):
    """A frame rendered without autograd (torch.no_grad(), or nothing requires a gradient) is rendered for its image: by
    default it takes the exact forward (int count, never truncated); with autograd recording the same call is speculative;
    set_forward_mode(inference_speculative=True) / GOI_FORWARD_INFERENCE=speculative lifts the restriction."""
    from goi_hyperplane_amd import _C, rasterizer
    from goi_hyperplane_amd.render import GaussianSet, PipelineParams, TorchCamera, render
    sc = make_scene(5000, S=16, seed=21, log_scale_mean=-2.9)
    cam = TorchCamera(make_camera(200, 152), dev)
    pc = GaussianSet.from_scene(sc, dev)
    bg = torch.zeros(3, device=dev)
    render(cam, pc, PipelineParams(), bg)  # teaches the policy
    a = render(cam, pc, PipelineParams(), bg)
    assert isinstance(rasterizer.last_num_rendered(), _C.LazyCount)  # training frame: speculative
    with torch.no_grad():
        b = render(cam, pc, PipelineParams(), bg)
        assert isinstance(rasterizer.last_num_rendered(), int)  # image-only frame: exact
        _C.set_forward_mode(inference_speculative=True)
        c = render(cam, pc, PipelineParams(), bg)
        assert isinstance(rasterizer.last_num_rendered(), _C.LazyCount)
    for p in pc.parameters():
        p.requires_grad_(False)
    _C.set_forward_mode(inference_speculative=False)
    d = render(cam, pc, PipelineParams(), bg)  # grad mode on, but nothing to differentiate: image-only as well
    assert isinstance(rasterizer.last_num_rendered(), int)
    for x in (b, c, d):
        assert torch.equal(a["render"].detach(), x["render"]) and torch.equal(a["semantics"].detach(), x["semantics"])


def test_a_much_larger_image_starts_the_capacity_policy_over(dev):
    """The policy sizes a speculative frame from the counts it has seen.  num_rendered grows with the image: after frames of
    a small image, the first frame of one with more than twice the tiles must be EXACT again (found by a fuzz soak: a frame
    sized from a much smaller one overflowed and showed a truncated image)."""
    from goi_hyperplane_amd import _C
    sc = make_scene(3000, S=4, seed=5, log_scale_mean=-1.6)
    bg = np.zeros(3, np.float32)
    _C.poll_counts(wait=True)
    _C.set_forward_mode(speculative=True, headroom=2.0, capacity=None, min_history=3)
    small, large = make_camera(64, 48), make_camera(320, 240)
    for _ in range(5):
        n, *_ = _C.rasterize_gaussians(*_args(sc, small, dev, bg))
    assert isinstance(n, _C.LazyCount), "the small image should be speculative by now"
    n_small = int(n)
    n, color, *_ = _C.rasterize_gaussians(*_args(sc, large, dev, bg))
    assert not isinstance(n, _C.LazyCount), "first frame of a 25x larger image: exact"
    assert int(n) > 4 * n_small  # (it would not have fitted 2 x the small image's count)
    _C.set_forward_mode(speculative=False)
    n_ref, color_ref, *_ = _C.rasterize_gaussians(*_args(sc, large, dev, bg))
    assert int(n_ref) == int(n) and torch.equal(color_ref, color)


@pytest.mark.parametrize("P,S,W,H,mu", [(20000, 16, 400, 300, -3.3), (3000, 10, 160, 120, -2.8)])
def test_row_scratch_is_sized_by_the_count_once_it_has_arrived(dev, P, S, W, H, mu):
    """goi_raster_backward3: a speculative frame's workspaces are laid out for its CAPACITY, but the backward's row scratch (129
    bytes per instance and quadrant: the largest workspace of a step) only has to hold the COUNT -- and the count has usually
    reached the host by the time the backward is enqueued.  The binding polls (no wait) and lays the scratch out for whichever
    is known: the gradients are bit-identical either way, and identical to those of an exact frame."""
    from goi_hyperplane_amd import _C, rasterizer
    from goi_hyperplane_amd.render import GaussianSet, PipelineParams, TorchCamera, render
    sc = make_scene(P, S=S, seed=4, log_scale_mean=mu)
    cam = TorchCamera(make_camera(W, H, yaw=0.1), dev)
    pc = GaussianSet.from_scene(sc, dev)
    bg = torch.zeros(3, device=dev)
    gen = torch.Generator(device=dev).manual_seed(5)
    gc = torch.randn((3, H, W), device=dev, generator=gen) / (W * H)
    gs = torch.randn((S, H, W), device=dev, generator=gen) / (W * H)

    def step(wait):
        for p in pc.parameters():
            p.grad = None
        out = render(cam, pc, PipelineParams(), bg)
        if wait:
            torch.cuda.synchronize()  # the frame has run: its count is in the pinned words (nobody has looked yet)
        torch.autograd.backward((out["render"], out["semantics"]), (gc, gs))
        torch.cuda.synchronize()
        return [p.grad.clone() for p in pc.parameters()]
    _C.set_forward_mode(speculative=False)
    want = step(False)
    n_exact = int(rasterizer.last_num_rendered())
    _C.set_forward_mode(speculative=True, capacity=3 * n_exact + 999)
    s0 = dict(_C.SCRATCH_STATS)
    got_late = step(True)    # count known at the backward: scratch for the count
    s1 = dict(_C.SCRATCH_STATS)
    assert s1["sized_by_count"] == s0["sized_by_count"] + 1 and s1["sized_by_capacity"] == s0["sized_by_capacity"]
    for a, b in zip(want, got_late):
        assert torch.equal(a, b)
    lib = __import__("goi_hyperplane_amd._lib", fromlist=["load"]).load()
    assert lib.goi_raster_backward_scratch_bytes(n_exact, S) * 2 < lib.goi_raster_backward_scratch_bytes(3 * n_exact + 999, S)
    got_any = step(False)    # whatever is known (usually nothing yet): same gradients
    s2 = dict(_C.SCRATCH_STATS)
    assert s2["sized_by_count"] + s2["sized_by_capacity"] == s1["sized_by_count"] + s1["sized_by_capacity"] + 1
    for a, b in zip(want, got_any):
        assert torch.equal(a, b)
