"""Gradient-buffer pool of the compiled binding (csrc/torch_binding.cpp; goi_raster_backward2 in include/goi_raster.h).

The reference hands autograd eleven zero-filled [P, ..] tensors per backward (rasterize_points.cu:252-262).  The binding keeps the
one allocation all outputs of a backward are views of and reuses it once nobody holds it and nothing wrote to it in place; the
kernel then skips the rows of Gaussians that were invisible in the frame that last wrote the buffer and are invisible now.  What
must hold: every gradient is bit-identical to the pool-less backward whatever the sequence of cameras, a buffer somebody still
holds is never handed out, and a buffer that was modified in place is treated as fresh."""
import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def dev():
    assert torch.cuda.is_available(), "GPU tests need a ROCm device"
    from goi_hyperplane_amd import _lib
    _lib.load()
    return torch.device("cuda:0")


def _ext():
    from goi_hyperplane_amd import _C
    ext = _C._ext()
    if ext is None:
        pytest.skip("the compiled binding is not built (the pool lives there)")
    return ext


def _setup(dev, P=60_000, W=400, H=300, S=16):
    from goi_hyperplane_amd.render import GaussianSet, TorchCamera
    from goi_hyperplane_amd.scene import make_camera, make_scene
    sc = make_scene(P, S=S, sh_degree=3, seed=31, extent=(6.0, 4.0, 1.0), log_scale_mean=-3.4)  # wider than the frustum:
    pc = GaussianSet.from_scene(sc, dev)                                                         # half the Gaussians are culled
    cams = [TorchCamera(make_camera(W, H, yaw=y, pitch=p, fovx=0.7), dev) for y, p in
            ((0.0, 0.0), (0.5, 0.1), (-0.6, -0.1), (0.05, 0.0), (0.9, 0.2), (0.0, 0.0))]
    gen = torch.Generator(device=dev).manual_seed(2)
    ups = [torch.randn(shape, device=dev, generator=gen) / (W * H) for shape in ((3, H, W), (S, H, W), (1, H, W), (1, H, W))]
    return pc, cams, ups


def _grads(cam, pc, ups, hold=None):
    from goi_hyperplane_amd.render import PipelineParams, render
    for p in pc.parameters():
        p.grad = None
    out = render(cam, pc, PipelineParams(), torch.zeros(3, device=cam.camera_center.device))
    torch.autograd.backward((out["render"], out["semantics"], out["depth"], out["alpha"]), ups)
    g = [p.grad for p in pc.parameters()] + [out["viewspace_points"].grad]
    if hold is not None:
        hold.append(g)
    vis = int((out["radii"] > 0).sum())
    return [t.clone() for t in g], vis


def test_pooled_backward_is_bit_identical_over_a_sequence_of_cameras(dev):
    ext = _ext()
    pc, cams, ups = _setup(dev)
    ext.set_grad_pool(False)
    ref = [_grads(c, pc, ups) for c in cams]
    assert any(v < 0.8 * pc._xyz.shape[0] for _, v in ref), "the test needs views that cull a good part of the scene"
    ext.set_grad_pool(True)
    try:
        h0 = ext.grad_pool_stats()
        got = [_grads(c, pc, ups) for c in cams + cams]  # the buffers of one view are reused by the next
        h1 = ext.grad_pool_stats()
        assert h1[0] - h0[0] >= len(cams), (h0, h1)  # hits: reused with rows skipped
        for (g, _), (r, _) in zip(got, ref + ref):
            for a, b in zip(g, r):
                assert torch.equal(a, b)
    finally:
        ext.set_grad_pool(True)


def test_a_buffer_somebody_holds_is_not_reused_and_a_modified_one_counts_as_fresh(dev):
    ext = _ext()
    pc, cams, ups = _setup(dev)
    ext.set_grad_pool(False)
    ref = [_grads(c, pc, ups)[0] for c in cams[:4]]
    ext.set_grad_pool(True)
    held = []
    g0, _ = _grads(cams[0], pc, ups, hold=held)          # the caller keeps the gradients of view 0 ...
    kept = [t.clone() for t in held[0]]
    g1, _ = _grads(cams[1], pc, ups)                      # ... while view 1 runs: another buffer
    for a, b in zip(held[0], kept):
        assert torch.equal(a, b)                          # view 0's gradients were not overwritten
    for a, b in zip(g1, ref[1]):
        assert torch.equal(a, b)
    # in-place modification (a gradient clip, an in-place all-reduce): the version counter says so, every row is rewritten
    held.clear()                                          # view 0's buffer is free again, untouched
    s0 = ext.grad_pool_stats()
    for p in pc.parameters():
        p.grad.add_(1.0)                                  # view 1's buffer: rows of invisible Gaussians are no longer zero
        p.grad = None
    g2, _ = _grads(cams[2], pc, ups, hold=held)           # takes one of the two free buffers and keeps it ...
    g3, _ = _grads(cams[3], pc, ups)                      # ... so this one takes the other
    s1 = ext.grad_pool_stats()
    assert s1[1] > s0[1]                                  # the modified one was counted as dirty
    for g, r in ((g2, ref[2]), (g3, ref[3])):
        for a, b in zip(g, r):
            assert torch.equal(a, b)
