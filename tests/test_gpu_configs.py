"""BASELINE.json config 5, the part one GPU can run: 6 M Gaussians, hyperplane-masked render (60 % kept) at the
editing resolution 512x512 with the masked-MSE guidance loss of gui/main_edit.py:551-669 (white background, image
clamped to [0,1], dense dL/dcolour inside a dilated image-space mask) and EVERY Gaussian parameter trainable.  The
oracle cannot run this size, so the checks are size-independent properties of the operator:

  * forward and every gradient are bit-reproducible;
  * masked render == render of the kept subset (gui/gs_renderer.py:315-321 index-selects the tensors): outputs
    bit-equal, gradients of kept Gaussians bit-equal to the subset's, gradients of masked-out Gaussians exactly zero;
  * the backward is linear in the upstream gradient: grad(u1 + u2) == grad(u1) + grad(u2) to fp32 rounding;
  * colour checksum: with precomputed colours and upstream dL/dcolour = 1 on one channel, the sum over Gaussians of
    dL/dcolour[g][c] equals the alpha mass of the image (sum_g w[pix,g] = 1 - T[pix]).
"""
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


def test_config5_masked_edit_step_properties(dev):
    from goi_hyperplane_amd.render import GaussianSet, PipelineParams, TorchCamera, render, render_gui
    from goi_hyperplane_amd.scene import HEADLINE, make_camera, make_scene
    P, S, R = 6_000_000, 16, 512
    sc = make_scene(P, S=S, sh_degree=3, seed=5, extent=HEADLINE["extent"], log_scale_mean=HEADLINE["log_scale_mean"],
                    log_scale_std=HEADLINE["log_scale_std"])
    cam = TorchCamera(make_camera(R, R, fovx=HEADLINE["fovx"], yaw=0.08, pitch=-0.03), dev)
    pc = GaussianSet.from_scene(sc, dev)
    del sc
    white = torch.ones(3, device=dev)
    g = torch.Generator(device=dev).manual_seed(0)
    keep = torch.rand(P, device=dev, generator=g) < 0.6                      # the hyperplane's side
    target = torch.rand((3, R, R), device=dev, generator=g)                  # stands in for the inpainted image
    yy, xx = torch.meshgrid(torch.arange(R, device=dev), torch.arange(R, device=dev), indexing="ij")
    region = (((xx - 280) ** 2 + (yy - 230) ** 2) < 170 ** 2).float()[None]  # dilated semantic mask [1,H,W]
    params = dict(pc.named_parameters())

    def edit_step(model, mask):
        for p in model.parameters():
            p.grad = None
        out = render_gui(cam, model, white, gaussian_mask=mask)
        loss = (((out["image"] - target) ** 2) * region).sum()               # masked MSE, reduction='none' * mask
        loss.backward()
        return out, {n: p.grad.clone() for n, p in model.named_parameters()}

    o1, g1 = edit_step(pc, keep)
    o2, g2 = edit_step(pc, keep)
    for k in ("image", "semantics", "depth", "alpha"):
        assert torch.equal(o1[k], o2[k]), f"forward {k} not reproducible"
    assert float(o1["alpha"].detach().mean()) > 0.5
    for n in g1:
        assert torch.equal(g1[n], g2[n]), f"gradient {n} not bit-reproducible"
        assert torch.isfinite(g1[n]).all()
        assert float(g1[n][~keep].abs().max()) == 0.0, f"{n}: a masked-out Gaussian got a gradient"
    for n in ("_xyz", "_features", "_opacity", "_scaling", "_rotation"):
        assert float(g1[n].abs().max()) > 0, f"{n}: no gradient from a dense colour loss"
    assert float(g1["_semantics"].abs().max()) == 0.0  # the loss does not touch the feature map

    # masked == subset, forward and backward
    sub = GaussianSet(*(params[n].detach()[keep].clone() for n in ("_xyz", "_scaling", "_rotation", "_opacity",
                                                                   "_features", "_semantics")))
    o3, g3 = edit_step(sub, None)
    for k in ("image", "semantics", "depth", "alpha"):
        assert torch.equal(o3[k], o1[k]), k
    for n in g3:
        assert torch.equal(g3[n], g1[n][keep]), n
    del sub, o3, g3, g2, o2

    # linearity of the backward in the upstream gradient
    gen = torch.Generator(device=dev).manual_seed(3)
    u1 = torch.randn((3, R, R), device=dev, generator=gen) * region
    u2 = torch.randn((3, R, R), device=dev, generator=gen) * region
    leaves = [params[n] for n in ("_xyz", "_features", "_opacity", "_scaling", "_rotation")]
    out = render(cam, pc, PipelineParams(), white, gaussian_mask=keep)
    ga = torch.autograd.grad(out["render"], leaves, u1, retain_graph=True)
    gb = torch.autograd.grad(out["render"], leaves, u2, retain_graph=True)
    gab = torch.autograd.grad(out["render"], leaves, u1 + u2)
    for a, b, ab, n in zip(ga, gb, gab, ("xyz", "sh", "opacity", "scaling", "rotation")):
        scale = float(ab.abs().max())
        assert float((ab - (a + b)).abs().max()) <= 2e-4 * scale, n  # (fp32 sums of ~1e2 terms in two orders)
    del ga, gb, gab, out

    # colour checksum through precomputed colours
    colors = torch.rand((P, 3), device=dev, generator=gen).requires_grad_()
    out = render(cam, pc, PipelineParams(), torch.zeros(3, device=dev), override_color=colors, gaussian_mask=keep)
    out["render"][1].sum().backward()
    alpha_mass = out["alpha"].double().sum().item()
    gcol = colors.grad.double()
    assert abs(gcol[:, 1].sum().item() - alpha_mass) < 1e-4 * alpha_mass
    assert float(gcol[:, 0].abs().max()) == 0.0 and float(gcol[:, 2].abs().max()) == 0.0
    assert float(gcol[~keep].abs().max()) == 0.0
