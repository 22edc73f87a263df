"""GPU parity of the operand paths of the raw operator that the first suite only touched through the Python
harness: precomputed colours + precomputed 3D covariance (rasterize_points.cu:213-306 returns dL/dcolors_precomp and
dL/dcov3D_precomp), scale_modifier != 1 (CR/forward.cu:122-124, CR/backward.cu:286-290), prefiltered=True
(CR/auxiliary.h:154-161 traps) and debug=True (DGR/diff_gaussian_rasterization/__init__.py:112-119,165-172 dump
the arguments when the call raises).  Same tolerances as tests/test_gpu_parity.py: forward 1e-4 outside the oracle's
fragile pixels, gradients 1e-3 of each tensor's scale, plus the element-wise statistics check_backward returns."""
import os

import numpy as np
import pytest
import torch

from goi_hyperplane_amd.scene import make_camera, make_scene
from tests.golden.make_golden import ORACLE_CASES, upstream_grads
from tests.test_gpu_parity import BWD_TOL, check_backward, check_forward, dev, run_hip  # noqa: F401  (dev: fixture)

pytestmark = pytest.mark.gpu


def _cov3d(scales, rotations, mod=1.0):
    """pc.get_covariance (scene/gaussian_model.py:33-37): L = R(q/|q|) diag(mod s), Sigma = L L^T, packed upper triangle."""
    q = rotations / np.linalg.norm(rotations, axis=1, keepdims=True)
    r, x, y, z = q[:, 0], q[:, 1], q[:, 2], q[:, 3]
    R = np.stack([1 - 2 * (y * y + z * z), 2 * (x * y - r * z), 2 * (x * z + r * y),
                  2 * (x * y + r * z), 1 - 2 * (x * x + z * z), 2 * (y * z - r * x),
                  2 * (x * z - r * y), 2 * (y * z + r * x), 1 - 2 * (x * x + y * y)], -1).reshape(-1, 3, 3)
    L = R * (mod * scales)[:, None, :]
    S = L @ L.transpose(0, 2, 1)
    return np.stack([S[:, 0, 0], S[:, 0, 1], S[:, 0, 2], S[:, 1, 1], S[:, 1, 2], S[:, 2, 2]], -1).astype(np.float32)


def _raster(dev, cam, bg, scale_modifier=1.0, sh_degree=3, prefiltered=False, debug=False):
    from goi_hyperplane_amd.rasterizer import GaussianRasterizationSettings, GaussianRasterizer
    from goi_hyperplane_amd.render import TorchCamera
    tc = TorchCamera(cam, dev)
    rs = GaussianRasterizationSettings(cam.image_height, cam.image_width, cam.tanfovx, cam.tanfovy,
                                       torch.tensor(bg, device=dev), scale_modifier, tc.world_view_transform,
                                       tc.full_proj_transform, sh_degree, tc.camera_center, prefiltered, debug)
    return GaussianRasterizer(rs)


def _leaf(a, dev):
    return torch.tensor(np.asarray(a, np.float32), device=dev, requires_grad=True)


@pytest.mark.parametrize("P,S,W,H,mu", [(2000, 16, 160, 120, -2.8), (1200, 10, 123, 77, -2.4), (600, 3, 64, 48, -1.6)])
def test_precomputed_colour_and_covariance_operands(oracle_mod, dev, P, S, W, H, mu):  # noqa: F811
    sc = make_scene(P, S=S, seed=21, log_scale_mean=mu)
    cam = make_camera(W, H, yaw=0.12, pitch=-0.07)
    bg = np.array([0.15, 0.25, 0.05], np.float32)
    colors = np.random.default_rng(3).random((P, 3)).astype(np.float32)
    cov = _cov3d(sc.scales, sc.rotations)
    grads = upstream_grads(S, H, W, seed=8)
    o = oracle_mod.from_scene(sc, cam, bg=bg, shs=None, scales=None, rotations=None, colors_precomp=colors,
                              cov3D_precomp=cov)
    f = o.forward()
    means3D, opac, sem = _leaf(sc.means3D, dev), _leaf(sc.opacities, dev), _leaf(sc.semantics, dev)
    col, cv = _leaf(colors, dev), _leaf(cov, dev)
    m2d = torch.zeros(P, 3, device=dev, requires_grad=True)
    color, semant, radii, depth, alpha = _raster(dev, cam, bg)(
        means3D=means3D, means2D=m2d, opacities=opac, colors_precomp=col, semantics=sem, cov3D_precomp=cv)
    res = dict(render=color.detach().cpu().numpy(), semantics=semant.detach().cpu().numpy(),
               depth=depth.detach().cpu().numpy(), alpha=alpha.detach().cpu().numpy(), radii=radii.cpu().numpy())
    tag = f"precomp_P{P}_S{S}_{W}x{H}"
    check_forward(res, f, tag)  # 1e-4, radii exact
    gc, gs, gd, ga = (torch.tensor(g, device=dev) for g in grads)
    ((color * gc).sum() + (semant * gs).sum() + (depth * gd).sum() + (alpha * ga).sum()).backward()
    g = o.backward(*grads)
    got = dict(means3D=means3D.grad, opacity=opac.grad, semantics=sem.grad, colors=col.grad, cov3D=cv.grad,
               means2D=m2d.grad)
    got = {k: v.detach().cpu().numpy() for k, v in got.items()}
    assert float(np.abs(got["colors"]).max()) > 0 and float(np.abs(got["cov3D"]).max()) > 0
    stats = check_backward(got, g, tag, names=("means3D", "opacity", "semantics", "colors", "cov3D", "means2D"))
    # element-wise: at most a handful of elements (guard flips) may sit outside 1e-3 of the tensor's scale
    for name, s in stats.items():
        assert s["p9999"] < BWD_TOL, (tag, name, s)


@pytest.mark.parametrize("mod", [0.7, 1.6])
def test_scale_modifier(oracle_mod, dev, mod):  # noqa: F811
    P, S, W, H = 1500, 16, 128, 96
    sc = make_scene(P, S=S, seed=31, log_scale_mean=-2.7)
    cam = make_camera(W, H, yaw=-0.1, pitch=0.05)
    bg = np.array([0.3, 0.1, 0.2], np.float32)
    grads = upstream_grads(S, H, W, seed=4)
    o = oracle_mod.from_scene(sc, cam, bg=bg, scale_modifier=mod)
    f = o.forward()
    leaves = {k: _leaf(getattr(sc, k), dev) for k in ("means3D", "opacities", "semantics", "shs", "scales", "rotations")}
    m2d = torch.zeros(P, 3, device=dev, requires_grad=True)
    color, semant, radii, depth, alpha = _raster(dev, cam, bg, scale_modifier=mod)(
        means3D=leaves["means3D"], means2D=m2d, opacities=leaves["opacities"], shs=leaves["shs"],
        semantics=leaves["semantics"], scales=leaves["scales"], rotations=leaves["rotations"])
    res = dict(render=color.detach().cpu().numpy(), semantics=semant.detach().cpu().numpy(),
               depth=depth.detach().cpu().numpy(), alpha=alpha.detach().cpu().numpy(), radii=radii.cpu().numpy())
    check_forward(res, f, f"mod{mod}")
    # a different modifier really is a different image
    f1 = oracle_mod.from_scene(sc, cam, bg=bg).forward()
    assert np.abs(f1.color - f.color).max() > 1e-2
    gc, gs, gd, ga = (torch.tensor(g, device=dev) for g in grads)
    ((color * gc).sum() + (semant * gs).sum() + (depth * gd).sum() + (alpha * ga).sum()).backward()
    got = dict(means3D=leaves["means3D"].grad, opacity=leaves["opacities"].grad, semantics=leaves["semantics"].grad,
               sh=leaves["shs"].grad, scales=leaves["scales"].grad, rotations=leaves["rotations"].grad, means2D=m2d.grad)
    check_backward({k: v.detach().cpu().numpy() for k, v in got.items()}, o.backward(*grads), f"mod{mod}")


def test_prefiltered_raises_the_reference_message(dev):  # noqa: F811
    """prefiltered=True promises that no Gaussian is behind the near plane; the reference prints
    "Point is filtered although prefiltered is set. This shouldn't happen!" and traps (CR/auxiliary.h:154-161)."""
    sc = make_scene(500, S=10, seed=1)
    sc.means3D[::7, 2] = -20.0  # behind the camera
    cam = make_camera(64, 48)
    t = lambda a: torch.tensor(a, device=dev)  # noqa: E731
    args = dict(means3D=t(sc.means3D), means2D=None, opacities=t(sc.opacities), shs=t(sc.shs), semantics=t(sc.semantics),
                scales=t(sc.scales), rotations=t(sc.rotations))
    with pytest.raises(RuntimeError, match="Point is filtered although prefiltered is set"):
        _raster(dev, cam, np.zeros(3, np.float32), prefiltered=True)(**args)
    # the same scene without the promise renders; and with the promise KEPT (nothing behind) it renders identically
    a = _raster(dev, cam, np.zeros(3, np.float32))(**args)
    keep = sc.means3D[:, 2] > -10
    args2 = {k: (None if v is None else v[torch.tensor(keep, device=dev)]) for k, v in args.items()}
    b = _raster(dev, cam, np.zeros(3, np.float32), prefiltered=True)(**args2)
    assert torch.equal(a[0], b[0]) and torch.equal(a[1], b[1])


def test_debug_mode_runs_and_dumps_on_failure(dev, tmp_path, monkeypatch):  # noqa: F811
    """debug=True: every stage is followed by a stream synchronisation + error check (CR/auxiliary.h:166-174
    CHECK_CUDA) and a failing call leaves snapshot_fw.dump / snapshot_bw.dump (the Python wrapper's behaviour)."""
    monkeypatch.chdir(tmp_path)
    sc = make_scene(800, S=16, seed=2, log_scale_mean=-2.6)
    cam = make_camera(96, 64, yaw=0.1)
    bg = np.array([0.1, 0.2, 0.3], np.float32)

    def run(debug):
        leaves = {k: _leaf(getattr(sc, k), dev) for k in ("means3D", "opacities", "semantics", "shs", "scales", "rotations")}
        out = _raster(dev, cam, bg, debug=debug)(
            means3D=leaves["means3D"], means2D=torch.zeros(sc.P, 3, device=dev, requires_grad=True),
            opacities=leaves["opacities"], shs=leaves["shs"], semantics=leaves["semantics"], scales=leaves["scales"],
            rotations=leaves["rotations"])
        (out[0].sum() + out[1].sum() + out[3].sum()).backward()
        return [o.detach().clone() for o in out], {k: v.grad.clone() for k, v in leaves.items()}

    o0, g0 = run(False)
    o1, g1 = run(True)
    for a, b in zip(o0, o1):
        assert torch.equal(a, b)
    for k in g0:
        assert torch.equal(g0[k], g1[k]), k
    assert not os.path.exists("snapshot_fw.dump")
    # a failing forward in debug mode dumps its arguments
    sc.means3D[::5, 2] = -20.0
    t = lambda a: torch.tensor(a, device=dev)  # noqa: E731
    with pytest.raises(RuntimeError):
        _raster(dev, cam, bg, prefiltered=True, debug=True)(
            means3D=t(sc.means3D), means2D=None, opacities=t(sc.opacities), shs=t(sc.shs), semantics=t(sc.semantics),
            scales=t(sc.scales), rotations=t(sc.rotations))
    assert os.path.exists("snapshot_fw.dump")
    snap = torch.load("snapshot_fw.dump")
    assert isinstance(snap, tuple) and len(snap) == 20 and snap[1].shape == (sc.P, 3) and snap[1].device.type == "cpu"


@pytest.mark.parametrize("name", list(ORACLE_CASES))
def test_committed_cases_against_the_float64_reference(oracle_mod, dev, name):  # noqa: F811
    """Second yardstick for the committed cases: the dense float64 autograd restatement written from the mathematics
    (tests/torch_reference.py).  HIP and the fp32 oracle are both measured against it; the HIP path must be within the
    north_star tolerance of the EXACT gradient too, not only of the reference-order fp32 one."""
    from tests.torch_reference import float64_gradients
    c = ORACLE_CASES[name]
    sc = make_scene(c["P"], S=c["S"], sh_degree=c["deg"], seed=7, log_scale_mean=c["mu"])
    cam = make_camera(c["W"], c["H"], yaw=0.15, pitch=-0.1)
    bg = np.array([0.1, 0.2, 0.3], np.float32)
    grads = upstream_grads(c["S"], c["H"], c["W"])
    res = run_hip(sc, cam, bg, dev, grads=grads)
    g64 = float64_gradients(sc, cam, bg, grads, c["deg"])
    o = oracle_mod.from_scene(sc, cam, bg=bg)
    o.forward()
    g32 = o.backward(*grads)
    s_hip = check_backward(res["grads"], g64, name + "_hip_vs_f64")
    s_orc = check_backward({k: np.asarray(g32[k]) for k in s_hip}, g64, name + "_oracle_vs_f64")
    for k in s_hip:  # the HIP path is not allowed to be an order of magnitude further from exact than the oracle
        assert s_hip[k]["max"] <= max(10 * s_orc[k]["max"], 2e-4), (name, k, s_hip[k], s_orc[k])


def test_gradient_gating_is_automatic(dev):  # noqa: F811
    """SURVEY.md section 7 "skip unneeded gradients": with the process defaults (no env var, no mode call) a model whose
    only trainable tensor is `_semantics` -- the reference's default training configuration, arguments/__init__.py:85-90
    -- takes the feature-gradient-only backward, and its dL/dsemantics is bit-identical to the full kernel's; with only
    the SH coefficients frozen the dL/dSH row is skipped and every other gradient is bit-identical."""
    from goi_hyperplane_amd import rasterizer
    from goi_hyperplane_amd.render import GaussianSet, PipelineParams, TorchCamera, render
    assert rasterizer._BACKWARD_MODE["semantics_only"] == "auto"  # the default
    sc = make_scene(5000, S=16, seed=12, log_scale_mean=-2.9)
    cam = TorchCamera(make_camera(208, 160, yaw=0.05), dev)
    pc = GaussianSet.from_scene(sc, dev)
    bg = torch.zeros(3, device=dev)
    g = torch.Generator(device=dev).manual_seed(4)
    up_c = torch.randn((3, 160, 208), device=dev, generator=g)
    up_s = torch.randn((16, 160, 208), device=dev, generator=g)

    def step(trainable):
        for n, p in pc.named_parameters():
            p.grad = None
            p.requires_grad_(n in trainable)
        out = render(cam, pc, PipelineParams(), bg)
        torch.autograd.backward((out["render"], out["semantics"]), (up_c, up_s))
        kernel = rasterizer.last_backward_kernel()
        grads = {n: (None if p.grad is None else p.grad.clone()) for n, p in pc.named_parameters()}
        return kernel, grads, out["viewspace_points"].grad.clone()

    everything = {"_xyz", "_scaling", "_rotation", "_opacity", "_features", "_semantics"}
    try:
        k_full, g_full, v_full = step(everything)
        assert k_full == "full" and all(v is not None for v in g_full.values())
        k_sem, g_sem, v_sem = step({"_semantics"})
        assert k_sem == "semantics"
        assert torch.equal(g_sem["_semantics"], g_full["_semantics"]) and float(v_sem.abs().max()) == 0.0
        assert all(g_sem[n] is None for n in everything - {"_semantics"})
        k_nosh, g_nosh, v_nosh = step(everything - {"_features"})
        assert k_nosh == "full_no_dsh" and g_nosh["_features"] is None
        for n in everything - {"_features"}:
            assert torch.equal(g_nosh[n], g_full[n]), n
        assert torch.equal(v_nosh, v_full)
        rasterizer.set_backward_mode(semantics_only=False)  # opt out: the reference's behaviour, full kernel always
        k_off, g_off, v_off = step({"_semantics"})
        assert k_off == "full" and torch.equal(g_off["_semantics"], g_full["_semantics"]) and torch.equal(v_off, v_full)
    finally:
        rasterizer.set_backward_mode(semantics_only="auto")
        for p in pc.parameters():
            p.requires_grad_(True)


def test_misaligned_semantics_are_refused(dev):  # noqa: F811
    """S a multiple of 4: the kernels move a Gaussian's semantic row as 16-byte words (by LDS-DMA in the backward); a [P, S]
    view that starts one float into its storage is contiguous, so nothing would copy it -- the C ABI refuses it by name."""
    P, S = 64, 16
    sc = make_scene(P, S=S, seed=5, log_scale_mean=-2.0)
    cam = make_camera(64, 48)
    means3D, opac = _leaf(sc.means3D, dev), _leaf(sc.opacities, dev)
    shs, scales, rots = _leaf(sc.shs, dev), _leaf(sc.scales, dev), _leaf(sc.rotations, dev)
    storage = torch.zeros(P * S + 4, device=dev)
    sem = storage[1:1 + P * S].view(P, S)
    assert sem.is_contiguous() and sem.data_ptr() % 16 != 0
    m2d = torch.zeros(P, 3, device=dev)
    with pytest.raises(Exception, match="16-byte aligned"):
        _raster(dev, cam, np.zeros(3, np.float32))(means3D, m2d, opac, shs=shs, scales=scales, rotations=rots, semantics=sem)
