"""Fused semantic training losses (csrc/codebook_loss.hip) against the PyTorch restatement of
train.py:142-163 (semantic.codebook_losses, itself pinned to the reference's lines on the CPU).
Tolerances: loss terms 1e-5 relative; gradients 1e-3 of their scale (BASELINE north_star)."""
import pytest
import torch

from goi_hyperplane_amd.semantic import SemanticModel, codebook_losses, fused_codebook_losses

pytestmark = pytest.mark.gpu


def setup(H, W, S=16, C=300, D=256, seed=0, bias=True):
    g = torch.Generator(device="cuda").manual_seed(seed)
    r = lambda *s: torch.randn(*s, device="cuda", generator=g)  # noqa: E731
    sem = (0.5 * r(S, H, W)).requires_grad_(True)
    torch.manual_seed(seed)
    mlp = SemanticModel(dim_in=S, dim_out=C, num_layer=1, use_bias=bias, device="cuda")
    lut = torch.nn.Parameter(torch.rand(C, D, device="cuda", generator=g) * 0.03)
    # ground truth built from code-book rows + noise, so that max sim is well separated for most pixels
    idx = torch.randint(0, C, (H * W,), device="cuda", generator=g)
    gtl = (lut.detach()[idx] * 30 + 0.3 * r(H * W, D)).t().reshape(D, H, W).contiguous()
    return sem, mlp, lut, gtl


def grads(fn, sem, mlp, lut, gtl, it):
    for p in [sem, lut, *mlp.parameters()]:
        p.grad = None
    loss, terms = fn(sem, mlp, lut, gtl, it)
    loss.backward()
    lin = mlp.layers[0]
    return loss.detach(), {k: v.detach() for k, v in terms.items()}, dict(
        sem=sem.grad.clone(), W=lin.weight.grad.clone(), b=None if lin.bias is None else lin.bias.grad.clone(),
        lut=lut.grad.clone())


@pytest.mark.parametrize("H,W,S,C,it,bias", [(40, 56, 16, 300, 10, True), (33, 47, 16, 300, 2000, True),
                                             (24, 40, 10, 40, 10, False), (17, 19, 3, 65, 1500, True)])
def test_matches_pytorch_restatement(H, W, S, C, it, bias):
    sem, mlp, lut, gtl = setup(H, W, S=S, C=C, bias=bias)
    l0, t0, g0 = grads(codebook_losses, sem, mlp, lut, gtl, it)
    l1, t1, g1 = grads(fused_codebook_losses, sem, mlp, lut, gtl, it)
    assert abs(float(l0 - l1)) <= 1e-5 * abs(float(l0))
    for k in t0:
        assert abs(float(t0[k] - t1[k])) <= 1e-5 * max(abs(float(t0[k])), 1e-3), k
    # pixels whose two best codes (or two best logits) are nearly tied may pick another label: exclude them from
    # the per-pixel comparison, keep them in the reductions
    with torch.no_grad():
        gn = gtl.reshape(gtl.shape[0], -1).t()
        gn = gn / gn.norm(dim=1, keepdim=True)
        sim = gn @ (lut / lut.norm(dim=1, keepdim=True)).t()
        top = sim.topk(2, dim=1).values
        z = mlp(sem.detach().permute(1, 2, 0).reshape(-1, S))
        ztop = z.topk(2, dim=1).values
        solid = ((top[:, 0] - top[:, 1]) > 1e-5) & ((ztop[:, 0] - ztop[:, 1]) > 1e-5)
    a, b = g0["sem"].reshape(S, -1)[:, solid], g1["sem"].reshape(S, -1)[:, solid]
    assert solid.float().mean() > 0.98
    assert float((a - b).abs().max()) <= 1e-3 * float(a.abs().max())
    for k in ("W", "b", "lut"):
        if g0[k] is None:
            assert g1[k] is None
            continue
        assert float((g0[k] - g1[k]).abs().max()) <= 1e-3 * float(g0[k].abs().max()), k


def test_reproducible_and_scales_with_upstream_gradient():
    sem, mlp, lut, gtl = setup(48, 64)
    _, _, ga = grads(fused_codebook_losses, sem, mlp, lut, gtl, 10)
    _, _, gb = grads(fused_codebook_losses, sem, mlp, lut, gtl, 10)
    for k in ga:
        assert torch.equal(ga[k], gb[k]), k
    for p in [sem, lut, *mlp.parameters()]:
        p.grad = None
    loss, _ = fused_codebook_losses(sem, mlp, lut, gtl, 10)
    (3.0 * loss).backward()
    assert torch.allclose(sem.grad, 3.0 * ga["sem"], rtol=1e-6, atol=0)


def test_headline_resolution_runs_and_agrees_on_the_loss():
    sem, mlp, lut, gtl = setup(1056, 1600)
    l1, t1, g1 = grads(fused_codebook_losses, sem, mlp, lut, gtl, 10)
    with torch.no_grad():
        l0, t0 = codebook_losses(sem.detach(), mlp, lut.detach(), gtl, 10)
    assert abs(float(l0 - l1)) <= 2e-5 * abs(float(l0))
    assert torch.isfinite(g1["sem"]).all() and torch.isfinite(g1["lut"]).all()
