"""The compiled reference-side binding (goi_hyperplane_amd/csrc/torch_binding.cpp -> lib/_goi_C.so: what a maintainer of
the reference would build in place of rasterize_points.cu + ext.cpp) against the ctypes binding of the same C ABI:
identical bits from both, the reference's four pybind signatures callable as the reference calls them
(rasterize_points.cu:35-123, :125-211, :213-306, :308-327), and the host time each binding spends per call."""
import json
import os
import time

import numpy as np
import pytest
import torch

from goi_hyperplane_amd.scene import make_camera, make_scene

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def dev():
    assert torch.cuda.is_available(), "GPU tests need a ROCm device"
    from goi_hyperplane_amd import _C
    _C.set_binding("compiled")  # fails loudly if lib/_goi_C.so has not been built
    return torch.device("cuda:0")


@pytest.fixture(autouse=True)
def _restore():
    from goi_hyperplane_amd import _C
    yield
    _C.poll_counts(wait=True)
    _C.set_binding("compiled")
    _C.set_forward_mode(speculative=True, capacity=None)


def _step(dev, sc, cam, bg, ups, **kw):
    from goi_hyperplane_amd.render import GaussianSet, PipelineParams, TorchCamera, render
    pc = GaussianSet.from_scene(sc, dev)
    out = render(TorchCamera(cam, dev), pc, PipelineParams(**kw), bg)
    torch.autograd.backward((out["render"], out["semantics"], out["depth"], out["alpha"]), ups)
    return ({k: out[k].detach().clone() for k in ("render", "semantics", "depth", "alpha", "radii")},
            {k: p.grad.clone() for k, p in pc.named_parameters()}, out["viewspace_points"].grad.clone())


@pytest.mark.parametrize("P,S,W,H,mu,kw", [(3000, 16, 160, 120, -2.8, {}), (1500, 10, 123, 77, -2.4, {}),
                                            (1200, 3, 96, 80, -2.2, dict(convert_SHs_python=True, compute_cov3D_python=True))])
def test_both_bindings_produce_identical_bits(dev, P, S, W, H, mu, kw):
    from goi_hyperplane_amd import _C
    sc = make_scene(P, S=S, seed=6, log_scale_mean=mu)
    cam = make_camera(W, H, yaw=0.1, pitch=0.05)
    bg = torch.tensor([0.2, 0.3, 0.1], device=dev)
    g = torch.Generator(device=dev).manual_seed(0)
    ups = [torch.randn(s, device=dev, generator=g) for s in ((3, H, W), (S, H, W), (1, H, W), (1, H, W))]
    got = {}
    for name in ("compiled", "ctypes"):
        _C.set_binding(name)
        assert _C.binding() == name
        for spec in (False, True):
            _C.set_forward_mode(speculative=spec)
            got[(name, spec)] = _step(dev, sc, cam, bg, ups, **kw)
    ref = got[("ctypes", False)]
    for key, val in got.items():
        for k in ref[0]:
            assert torch.equal(ref[0][k], val[0][k]), (key, k)
        for k in ref[1]:
            assert torch.equal(ref[1][k], val[1][k]), (key, k)
        assert torch.equal(ref[2], val[2]), key


def test_reference_signatures_called_the_way_the_reference_calls_them(dev):
    """Positional calls with the reference's argument lists, empty tensors for absent inputs, every upstream gradient a
    tensor -- exactly what diff_gaussian_rasterization/__init__.py:106-172 does with _C."""
    from goi_hyperplane_amd import _C
    from goi_hyperplane_amd.render import TorchCamera
    ext = _C._ext()
    sc = make_scene(2000, S=10, seed=9, log_scale_mean=-2.6)
    cam = make_camera(128, 96, yaw=-0.1)
    tc = TorchCamera(cam, dev)
    t = lambda a: torch.tensor(a, device=dev)  # noqa: E731
    E = torch.Tensor([])
    bg = torch.zeros(3, device=dev)
    fwd_args = (bg, t(sc.means3D), E, t(sc.semantics), t(sc.opacities), t(sc.scales), t(sc.rotations), 1.0, E,
                tc.world_view_transform, tc.full_proj_transform, cam.tanfovx, cam.tanfovy, 96, 128, t(sc.shs), 3,
                tc.camera_center, False, False)
    n, color, sem, depth, alpha, radii, geom, binning, img = ext.rasterize_gaussians(*fwd_args)
    assert isinstance(n, int) and n > 0 and color.shape == (3, 96, 128) and sem.shape == (10, 96, 128)
    _C.set_binding("ctypes")
    _C.set_forward_mode(speculative=False)
    n2, color2, sem2, depth2, alpha2, radii2, *_ = _C.rasterize_gaussians(*fwd_args)
    assert n2 == n and torch.equal(color, color2) and torch.equal(sem, sem2) and torch.equal(radii, radii2)
    g = torch.Generator(device=dev).manual_seed(1)
    ups = [torch.randn(s, device=dev, generator=g) for s in ((3, 96, 128), (10, 96, 128), (1, 96, 128), (1, 96, 128))]
    bwd_args = (bg, fwd_args[1], radii, E, fwd_args[3], fwd_args[5], fwd_args[6], 1.0, E, fwd_args[9], fwd_args[10],
                cam.tanfovx, cam.tanfovy, ups[0], ups[1], ups[2], ups[3], fwd_args[15], 3, fwd_args[17], geom, n, binning,
                img, alpha, False)
    a = ext.rasterize_gaussians_backward(*bwd_args)
    b = _C.rasterize_gaussians_backward(*bwd_args)
    assert len(a) == len(b) == 9
    for x, y in zip(a, b):
        assert torch.equal(x, y)
    # trace and markVisible
    img_sem = torch.randn((10, 96, 128), device=dev, generator=g)
    tr_args = fwd_args[:3] + (img_sem,) + fwd_args[4:]
    ta = ext.rasterize_gaussians_trace(*tr_args)
    tb = _C.rasterize_gaussians_trace(*tr_args)
    assert ta[0] == tb[0] and torch.equal(ta[1], tb[1]) and torch.equal(ta[3], tb[3])
    assert (ta[2] - tb[2]).abs().max() <= 1e-4 * tb[2].abs().max()  # (float atomics: order noise)
    assert torch.equal(ext.mark_visible(fwd_args[1], fwd_args[9], fwd_args[10]),
                       _C.mark_visible(fwd_args[1], fwd_args[9], fwd_args[10]))
    # the error behaviour survives the binding: wrong dtype / shape / missing semantics
    with pytest.raises(TypeError):
        ext.rasterize_gaussians(*((bg.double(),) + fwd_args[1:]))
    with pytest.raises(RuntimeError, match="means3D must have dimensions"):
        ext.rasterize_gaussians(*((bg, fwd_args[1][:, :2]) + fwd_args[2:]))
    with pytest.raises(RuntimeError, match="semantics"):
        ext.rasterize_gaussians(*(fwd_args[:3] + (E,) + fwd_args[4:]))


def test_host_time_per_call(dev):
    """Host cost of one forward + backward through each binding on the headline workload (the GPU is not waited for:
    speculative forward, time per enqueue).  Reported to gpurun_out/binding_host_time.json; the compiled binding must
    not be slower."""
    from goi_hyperplane_amd import _C
    from goi_hyperplane_amd.render import GaussianSet, PipelineParams, TorchCamera, render
    from goi_hyperplane_amd.scene import HEADLINE, make_headline_scene
    sc = make_headline_scene()
    pc = GaussianSet.from_scene(sc, dev)
    cam = TorchCamera(make_camera(HEADLINE["W"], HEADLINE["H"], fovx=HEADLINE["fovx"]), dev)
    bg = torch.zeros(3, device=dev)
    g = torch.Generator(device=dev).manual_seed(2)
    ups = (torch.randn((3, HEADLINE["H"], HEADLINE["W"]), device=dev, generator=g),
           torch.randn((HEADLINE["S"], HEADLINE["H"], HEADLINE["W"]), device=dev, generator=g))
    res = {}
    for name in ("ctypes", "compiled", "ctypes", "compiled"):
        _C.set_binding(name)
        _C.set_forward_mode(speculative=True)
        for i in range(3):
            out = render(cam, pc, PipelineParams(), bg)
            torch.autograd.backward((out["render"], out["semantics"]), ups)
        torch.cuda.synchronize()
        _C.poll_counts(wait=True)
        n = 24
        t0 = time.perf_counter()
        tf = 0.0
        for i in range(n):
            for p in pc.parameters():
                p.grad = None
            a = time.perf_counter()
            out = render(cam, pc, PipelineParams(), bg)
            tf += time.perf_counter() - a
            torch.autograd.backward((out["render"], out["semantics"]), ups)
        host = (time.perf_counter() - t0) / n * 1e3
        torch.cuda.synchronize()
        res.setdefault(name, []).append({"host_ms_per_step": host, "host_ms_forward": tf / n * 1e3})
    best = {k: min(v, key=lambda d: d["host_ms_per_step"]) for k, v in res.items()}
    os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
    with open(os.path.join(ROOT, "gpurun_out", "binding_host_time.json"), "w") as fh:
        json.dump({"what": "host ms per training step (render + backward enqueue, speculative forward, headline "
                           "workload, 24 steps, best of 2)", "runs": res, "best": best}, fh, indent=1)
    assert best["compiled"]["host_ms_per_step"] <= 1.05 * best["ctypes"]["host_ms_per_step"], best
