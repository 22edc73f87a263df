"""Speculative depth cut-off of the tile lists (include/goi_raster.h: goi_raster_forward_async_cut; _C._depth_cut_for).

A speculative training frame of a camera that was rendered before lists, per tile, only Gaussians up to the depth that camera's
previous frame found worth listing.  What must hold:
  * while the learnt cut holds, a cut frame's OUTPUTS are bit-identical to the uncut frame's and its gradients equal up to the
    order of one fp32 sum (every (quadrant, Gaussian) partial row is identical; the row reduction adds a Gaussian's rows in
    chunks of 16 listed tiles, and the cut changes which tiles are listed) -- with far fewer instances emitted and sorted;
  * a cut that is too tight never trains anything: the device raises the frame's flag, the backward writes zeros, reading the
    count before using the frame renders it again exactly, not reading it counts a skipped view; the camera forgets its cut;
  * a scene that moves between the visits of a camera is either rendered exactly or skipped, never wrongly."""
import warnings

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def dev():
    assert torch.cuda.is_available(), "GPU tests need a ROCm device"
    from goi_hyperplane_amd import _lib
    _lib.load()
    return torch.device("cuda:0")


def _setup(dev, P=300_000, W=800, H=528, mu=-3.8, S=16, seed=21):
    from goi_hyperplane_amd.render import GaussianSet, TorchCamera
    from goi_hyperplane_amd.scene import HEADLINE, make_camera, make_scene
    sc = make_scene(P, S=S, sh_degree=3, seed=seed, extent=HEADLINE["extent"], log_scale_mean=mu)
    cam = TorchCamera(make_camera(W, H, yaw=0.06, pitch=-0.02), dev)
    pc = GaussianSet.from_scene(sc, dev)
    gen = torch.Generator(device=dev).manual_seed(9)
    ups = [torch.randn(shape, device=dev, generator=gen) / (W * H) for shape in ((3, H, W), (S, H, W), (1, H, W), (1, H, W))]
    return sc, cam, pc, ups


def _same_gradients(got, ref):
    """equal up to the summation order of the row reduction (tests/test_gpu_parity.py::_check_culled has the same criterion)"""
    names = ["xyz", "scaling", "rotation", "opacity", "features", "semantics", "means2D"]
    for name, a, b in zip(names, got, ref):
        scale = float(b.abs().max()) + 1e-30
        tol = 1e-5 if name in ("semantics", "opacity", "features") else 1e-3
        assert float((a - b).abs().max()) <= tol * scale, (name, float((a - b).abs().max()) / scale)


def _step(cam, pc, ups, read_count=False):
    from goi_hyperplane_amd import rasterizer
    from goi_hyperplane_amd.render import PipelineParams, render
    for p in pc.parameters():
        p.grad = None
    out = render(cam, pc, PipelineParams(), torch.zeros(3, device=cam.camera_center.device))
    n = rasterizer.last_num_rendered()
    if read_count:
        int(n)
    torch.autograd.backward((out["render"], out["semantics"], out["depth"], out["alpha"]), ups)
    grads = [p.grad.clone() for p in pc.parameters()] + [out["viewspace_points"].grad.clone()]
    return out, n, grads


def test_cut_frames_are_bit_identical_to_uncut_frames_with_far_fewer_instances(dev):
    from goi_hyperplane_amd import _C, rasterizer
    sc, cam, pc, ups = _setup(dev)
    _C._SPEC.clear()
    _C.forget_depth_cuts()
    rasterizer.set_forward_mode(speculative=True, depth_cut=False)
    try:
        for _ in range(4):
            ref_out, ref_n, ref_g = _step(cam, pc, ups)
        n_uncut = int(ref_n)
        rasterizer.set_forward_mode(depth_cut=True)
        s0 = rasterizer.speculation_stats()
        _step(cam, pc, ups)              # learns (speculative, uncut)
        out1, n1, g1 = _step(cam, pc, ups)  # first cut frame
        out2, n2, g2 = _step(cam, pc, ups)  # a cut frame whose cut was learnt by a cut frame
        s1 = rasterizer.speculation_stats()
        assert isinstance(n1, _C.LazyCount) and n1.cut_key is not None and n2.cut_key is not None
        assert s1["cut_frames"] - s0["cut_frames"] == 2 and s1["cut_failures"] == s0["cut_failures"]
        for out, g in ((out1, g1), (out2, g2)):
            for k in ("render", "semantics", "depth", "alpha", "radii"):
                assert torch.equal(out[k], ref_out[k]), k
            _same_gradients(g, ref_g)
        assert not n1.cut_failed and not n2.cut_failed
        assert int(n1) < 0.7 * n_uncut and int(n2) < 0.7 * n_uncut, (int(n1), int(n2), n_uncut)
        assert int(n2) <= 1.05 * int(n1) + 4096  # (re-learning from a cut frame does not let the cut creep outwards)
    finally:
        rasterizer.set_forward_mode(depth_cut=False)
        _C.forget_depth_cuts()


def test_a_cut_that_is_too_tight_never_trains_and_is_repaired_when_the_count_is_read(dev):
    from goi_hyperplane_amd import _C, rasterizer
    sc, cam, pc, ups = _setup(dev, seed=22)
    _C._SPEC.clear()
    _C.forget_depth_cuts()
    rasterizer.set_forward_mode(speculative=True, depth_cut=False)
    try:
        for _ in range(4):
            ref_out, ref_n, ref_g = _step(cam, pc, ups)
        rasterizer.set_forward_mode(depth_cut=True)
        _step(cam, pc, ups)  # learns

        def sabotage():  # pull every learnt cut far in front of where the lists saturate
            (entry,) = _C._DEPTH_CUTS["entries"].values()
            entry["z"].mul_(0.5)

        # (1) nobody reads the count: zero gradients, the flag is set, a skipped view is counted at the next poll
        sabotage()
        s0 = rasterizer.speculation_stats()
        out, n, g = _step(cam, pc, ups)
        flag = rasterizer.truncated_flag()
        torch.cuda.synchronize()
        assert flag is not None and int(flag.item()) & 4
        assert all(float(t.abs().max()) == 0.0 for t in g)
        with warnings.catch_warnings(record=True):
            warnings.simplefilter("always")
            _C.poll_counts(dev, wait=True)
        s1 = rasterizer.speculation_stats()
        assert n.cut_failed and s1["cut_failures"] == s0["cut_failures"] + 1 and s1["skipped_views"] == s0["skipped_views"] + 1
        assert len(_C._DEPTH_CUTS["entries"]) == 0  # the camera forgot its cut
        # (2) the count is read before the frame is used: rendered again, exact
        _step(cam, pc, ups)  # learns again (no cut to apply)
        sabotage()
        out, n, g = _step(cam, pc, ups, read_count=True)
        assert n.cut_failed and n.redone and int(n) == int(ref_n)
        for k in ("render", "semantics", "depth", "alpha", "radii"):
            assert torch.equal(out[k], ref_out[k]), k
        for a, b in zip(g, ref_g):
            assert torch.equal(a, b)
    finally:
        rasterizer.set_forward_mode(depth_cut=False)
        _C.forget_depth_cuts()


def test_a_moving_scene_is_rendered_exactly_or_skipped_never_wrongly(dev):
    """Between two visits of a camera the Gaussians move (an optimiser at work): the cut learnt on the first visit either still
    holds -- the frame equals the uncut render of the MOVED scene bit for bit -- or the frame is flagged and its gradients are
    zero."""
    from goi_hyperplane_amd import _C, rasterizer
    sc, cam, pc, ups = _setup(dev, P=200_000, seed=23)
    _C._SPEC.clear()
    _C.forget_depth_cuts()
    rasterizer.set_forward_mode(speculative=True, depth_cut=True)
    gen = torch.Generator(device=dev).manual_seed(4)
    held = failed = 0
    try:
        for _ in range(4):
            _step(cam, pc, ups)
        for visit, amount in enumerate((0.002, 0.01, 0.05, 0.3)):  # fractions of the scene's extent
            with torch.no_grad():
                pc._xyz.add_(torch.randn(pc._xyz.shape, device=dev, generator=gen) * amount)
                pc._opacity.mul_(0.97)
            out, n, g = _step(cam, pc, ups)
            torch.cuda.synchronize()
            flagged = int(rasterizer.truncated_flag().item()) != 0
            _C._FWD["depth_cut"] = False  # (the uncut reference frame; what the camera has learnt is left alone)
            try:
                ref_out, ref_n, ref_g = _step(cam, pc, ups, read_count=True)
            finally:
                _C._FWD["depth_cut"] = True
            if flagged:
                failed += 1
                assert all(float(t.abs().max()) == 0.0 for t in g)
            else:
                held += 1
                for k in ("render", "semantics", "depth", "alpha", "radii"):
                    assert torch.equal(out[k], ref_out[k]), (visit, k)
                _same_gradients(g, ref_g)
            with warnings.catch_warnings(record=True):
                warnings.simplefilter("always")
                _C.poll_counts(dev, wait=True)
            _step(cam, pc, ups)  # (a frame that re-learns where the last one failed)
        assert held >= 1  # small motions must be absorbed by the margin
    finally:
        rasterizer.set_forward_mode(depth_cut=False)
        _C.forget_depth_cuts()


def test_oversized_gaussians_behind_the_cut_never_hide_dropped_ones(dev):
    """the cut lives in the ellipse tile masks, and a rectangle of more than 64 tiles has none -- such a Gaussian
    stays listed at every depth, so a cut list is NOT a prefix of the uncut one.  Scene: an opaque foreground (small Gaussians)
    in front of a mid layer (small) in front of frame-sized opaque blobs (hundreds of tiles each).  The camera learns a cut just
    behind the foreground; then the foreground turns transparent: every pixel passes the cut unsaturated and would stop on a
    blob -- having skipped the mid layer the cut dropped.  The frame must be flagged (zero gradients / rendered again exactly
    when the count is read), never returned as if it were exact."""
    from goi_hyperplane_amd import _C, rasterizer
    from goi_hyperplane_amd.render import GaussianSet, TorchCamera
    from goi_hyperplane_amd.scene import make_camera, make_scene
    W, H, S = 640, 480, 16
    sc = make_scene(60_000, S=S, sh_degree=1, seed=31, extent=(2.0, 1.5, 1.0), log_scale_mean=-3.0)
    P = sc.means3D.shape[0]
    rng = np.random.default_rng(5)
    layer = rng.integers(0, 3, P)                    # 0: foreground, 1: mid layer, 2: far blobs
    layer[rng.choice(P, 40, replace=False)] = 2
    layer[(layer == 2) & (np.arange(P) % 50 != 0)] = 1  # only a few dozen blobs
    sc.means3D[:, 2] = np.where(layer == 0, -0.6, np.where(layer == 1, 0.0, 0.6)) + 0.05 * rng.standard_normal(P)
    sc.scales[layer == 2] = 0.9                      # frame-sized: rectangles of several hundred tiles
    sc.opacities[:] = np.where(layer == 2, 0.99, 0.95)[:, None].astype(np.float32)  # opaque everywhere
    cam = TorchCamera(make_camera(W, H, yaw=0.0, pitch=0.0), dev)
    pc = GaussianSet.from_scene(sc, dev)
    gen = torch.Generator(device=dev).manual_seed(9)
    ups = [torch.randn(shape, device=dev, generator=gen) / (W * H) for shape in ((3, H, W), (S, H, W), (1, H, W), (1, H, W))]
    fg = torch.tensor(layer == 0, device=dev)
    _C._SPEC.clear()
    _C.forget_depth_cuts()
    rasterizer.set_forward_mode(speculative=True, depth_cut=True)
    try:
        for _ in range(4):
            _step(cam, pc, ups)  # exact frames, then speculative ones that learn the cut behind the foreground
        with torch.no_grad():
            pc._opacity[fg] = 0.02  # the foreground turns translucent: the cut no longer holds anywhere
        # the reference: the same scene without a cut
        _C._FWD["depth_cut"] = False
        try:
            ref_out, ref_n, ref_g = _step(cam, pc, ups, read_count=True)
        finally:
            _C._FWD["depth_cut"] = True
        # (1) nobody reads the count: flagged, zero gradients
        out, n, g = _step(cam, pc, ups)
        assert isinstance(n, _C.LazyCount) and n.cut_key is not None
        torch.cuda.synchronize()
        assert int(rasterizer.truncated_flag().item()) & 4, "a cut frame that looked beyond its cut was returned as exact"
        assert all(float(t.abs().max()) == 0.0 for t in g)
        with warnings.catch_warnings(record=True):
            warnings.simplefilter("always")
            _C.poll_counts(dev, wait=True)
        # (2) learn again on the old geometry, break it again, read the count: rendered again, exact
        with torch.no_grad():
            pc._opacity[fg] = 0.95
        for _ in range(2):
            _step(cam, pc, ups)
        with torch.no_grad():
            pc._opacity[fg] = 0.02
        out, n, g = _step(cam, pc, ups, read_count=True)
        assert n.cut_failed and n.redone
        for k in ("render", "semantics", "depth", "alpha", "radii"):
            assert torch.equal(out[k], ref_out[k]), k
    finally:
        rasterizer.set_forward_mode(depth_cut=False)
        _C.forget_depth_cuts()
