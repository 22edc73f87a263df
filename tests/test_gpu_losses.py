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


@pytest.mark.parametrize("HW,C", [(1, 4), (127, 300), (128, 304), (5000, 300), (33 * 47, 64), (1600 * 1056, 300)])
def test_similarity_kernel_against_fp64(HW, C):
    """goi_codebook_sim (split-bf16 MFMA, three partial products) against an fp64 product on the same inputs: the result
    must be as good as an fp32 GEMM's (2^-16-relative operand error squared is below fp32 rounding of a 256-term sum), so
    the bound is a few fp32 ulps of the scale.  Ragged pixel counts exercise the clamped tail block; C = 304 the full tile."""
    import ctypes as C_

    from goi_hyperplane_amd import _lib
    lib = _lib.load()
    D = 256
    g_ = torch.Generator(device="cuda").manual_seed(HW % 1000 + C)
    lut = torch.rand(C, D, device="cuda", generator=g_) * 0.03
    l1 = (lut / lut.norm(dim=1, keepdim=True)).contiguous()
    idx = torch.randint(0, C, (HW,), device="cuda", generator=g_)
    g = (lut[idx] * 30 + 0.3 * torch.randn(HW, D, device="cuda", generator=g_)).t().contiguous()  # [D, HW]
    sim = torch.full((HW, C), float("nan"), device="cuda")
    inv = torch.full((HW,), float("nan"), device="cuda")
    ws = torch.empty((int(lib.goi_codebook_sim_workspace_bytes()),), dtype=torch.uint8, device="cuda")
    p = lambda t: C_.c_void_p(t.data_ptr())  # noqa: E731
    rc = lib.goi_codebook_sim(p(g), p(l1), HW, C, D, p(sim), p(inv), p(ws),
                              C_.c_void_p(torch.cuda.current_stream().cuda_stream))
    assert rc == 0, _lib.last_error()
    n = min(HW, 100_000)
    ref = g.double().t()[:n] @ l1.double().t()
    scale = float(ref.abs().max())
    assert torch.isfinite(sim).all() and torch.isfinite(inv).all()
    assert float((sim[:n].double() - ref).abs().max()) <= 4e-6 * scale
    if HW > n:  # the rest against the library's fp32 product
        assert float((sim - g.t() @ l1.t()).abs().max()) <= 6e-6 * scale
    inv_ref = g.double().norm(dim=0).reciprocal()
    assert float(((inv.double() - inv_ref) / inv_ref).abs().max()) <= 2e-6


def test_similarity_kernel_rejects_shapes_it_does_not_cover():
    import ctypes as C_

    from goi_hyperplane_amd import _lib
    lib = _lib.load()
    t = torch.zeros(1024, device="cuda")
    p = C_.c_void_p(t.data_ptr())
    for HW, C, D in [(4, 300, 128), (4, 308, 256), (4, 302, 256), (0, 300, 256)]:
        assert lib.goi_codebook_sim(p, p, HW, C, D, p, p, p, None) != 0, (HW, C, D)


@pytest.mark.parametrize("H,W,S,C,it,bias", [(40, 56, 16, 300, 10, True), (36, 37, 16, 304, 2000, True),
                                             (12, 9, 7, 289, 10, False), (64, 128, 16, 300, 1500, True)])
def test_two_kernel_path_against_three_kernel_path(H, W, S, C, it, bias):
    """codebook_fused_k + codebook_dlut2_k (no [HW, C] matrix in memory) against sim kernel + row kernel + fp32 dLUT kernel on
    the same inputs: the loss terms to 1e-5 relative, gradients to 1e-3 of their scale away from near-ties (both paths
    evaluate sim with split-bf16 products, in different summation orders).  36x37 and 12x9 end inside a 16-pixel block."""
    from goi_hyperplane_amd import semantic
    sem, mlp, lut, gtl = setup(H, W, S=S, C=C, bias=bias)
    assert (H * W) % 4 == 0
    try:
        semantic._FUSED_KERNELS["on"] = False
        l0, t0, g0 = grads(fused_codebook_losses, sem, mlp, lut, gtl, it)
    finally:
        semantic._FUSED_KERNELS["on"] = True
    l1, t1, g1 = grads(fused_codebook_losses, sem, mlp, lut, gtl, it)
    assert abs(float(l0 - l1)) <= 1e-5 * abs(float(l0))
    for k in t0:
        assert abs(float(t0[k] - t1[k])) <= 1e-5 * max(abs(float(t0[k])), 1e-3), k
    for k in ("sem", "W", "b", "lut"):
        if g0[k] is None:
            assert g1[k] is None
            continue
        scale = float(g0[k].abs().max())
        bad = (g0[k] - g1[k]).abs() > 1e-3 * scale
        # a pixel whose best two codes tie to the last bit may label differently: allow a handful
        assert int(bad.sum()) <= (4 * S if k == "sem" else 0), (k, int(bad.sum()), float((g0[k] - g1[k]).abs().max()) / scale)


@pytest.mark.parametrize("fused", [True, False])
def test_duplicate_codebook_rows_label_every_maximum(fused):
    """train.py:151 labels EVERY code whose similarity equals the row maximum (`sim == sim.max(...)`).  With duplicate
    code-book rows (k-means can return them, train.py:78-84) whole groups of codes tie on every pixel they win: the label has
    several ones, the lab loss and dL/dz see all of them.  The five-kernel path carries that set as a 304-bit mask from
    codebook_simgrad_k to the decoder kernels -- this is the only test that reaches that branch."""
    from goi_hyperplane_amd import semantic
    H, W, S, C = 24, 40, 16, 300
    sem, mlp, lut, gtl = setup(H, W, S=S, C=C, bias=True)
    with torch.no_grad():
        lut[200] = lut[5]
        lut[18] = lut[17]
        lut[250] = lut[17]
        idx = torch.randint(0, C, (H * W,), device="cuda")
        idx[::3] = 5       # a third of the pixels are won by the pair, a fifth by the triple
        idx[1::5] = 17
        gtl.copy_((lut.detach()[idx] * 30 + 0.3 * torch.randn(H * W, 256, device="cuda")).t().reshape(256, H, W))
    l0, t0, g0 = grads(codebook_losses, sem, mlp, lut, gtl, 10)
    try:
        semantic._FUSED_KERNELS["on"] = fused
        l1, t1, g1 = grads(fused_codebook_losses, sem, mlp, lut, gtl, 10)
    finally:
        semantic._FUSED_KERNELS["on"] = True
    with torch.no_grad():  # the premise: many pixels really have 2 or 3 maxima
        gn = gtl.reshape(256, -1).t()
        sim = (gn / gn.norm(dim=1, keepdim=True)) @ (lut / lut.norm(dim=1, keepdim=True)).t()
        n_max = (sim == sim.max(dim=1, keepdim=True)[0]).sum(dim=1)
        assert int((n_max == 2).sum()) > 100 and int((n_max == 3).sum()) > 50
    assert abs(float(l0 - l1)) <= 1e-5 * abs(float(l0))
    for k in t0:
        assert abs(float(t0[k] - t1[k])) <= 1e-5 * max(abs(float(t0[k])), 1e-3), k
    for k in ("sem", "W", "b", "lut"):
        scale = float(g0[k].abs().max())
        bad = (g0[k] - g1[k]).abs() > 1e-3 * scale
        assert int(bad.sum()) <= (4 * S if k == "sem" else 0), (k, int(bad.sum()), float((g0[k] - g1[k]).abs().max()) / scale)


@pytest.mark.parametrize("seed", range(12))
def test_fused_path_fuzz(seed):
    """Random shapes inside goi_codebook_fused's domain (tab_len 289..304, semantic_dim 1..16, any H x W with HW % 4 = 0, with and
    without bias, both anneal stages) against the PyTorch restatement: every loss term to 1e-5, every gradient to 1e-3 of its
    scale away from near-ties."""
    import random

    from goi_hyperplane_amd import _lib, semantic
    rng = random.Random(1234 + seed)
    C = rng.randint(289, 304)
    S = rng.randint(1, 16)
    H = rng.randint(3, 70)
    W = 4 * rng.randint(1, 24)
    bias = rng.random() < 0.7
    it = rng.choice([10, 1500])
    sem, mlp, lut, gtl = setup(H, W, S=S, C=C, bias=bias, seed=seed)
    calls = {"n": 0}
    lib = _lib.load()
    real = lib.goi_codebook_fused

    class Spy:  # the fused entry point must be the one that runs for these shapes
        def __call__(self, *a):
            calls["n"] += 1
            return real(*a)
    try:
        lib.goi_codebook_fused = Spy()
        l1, t1, g1 = grads(fused_codebook_losses, sem, mlp, lut, gtl, it)
    finally:
        lib.goi_codebook_fused = real
    assert calls["n"] == 1, (C, S, H, W)
    l0, t0, g0 = grads(codebook_losses, sem, mlp, lut, gtl, it)
    assert abs(float(l0 - l1)) <= 1e-5 * abs(float(l0)), (C, S, H, W)
    for k in t0:
        assert abs(float(t0[k] - t1[k])) <= 1e-5 * max(abs(float(t0[k])), 1e-3), (k, C, S, H, W)
    for k in ("sem", "W", "b", "lut"):
        if g0[k] is None:
            assert g1[k] is None
            continue
        scale = float(g0[k].abs().max())
        bad = (g0[k] - g1[k]).abs() > 1e-3 * scale
        assert int(bad.sum()) <= (4 * S if k == "sem" else 0), (k, C, S, H, W, int(bad.sum()),
                                                               float((g0[k] - g1[k]).abs().max()) / scale)


def test_fused_entry_rejects_shapes_it_does_not_cover_and_the_wrapper_falls_back():
    """goi_codebook_fused covers tab_len 289..304, ape_dim 256, semantic_dim <= 16, HW % 4 = 0; anything else must come back as
    an error from the C ABI (nothing launched) and go through the three-step path in the Python wrapper."""
    import ctypes as C_

    from goi_hyperplane_amd import _lib
    lib = _lib.load()
    t = torch.zeros(4096, device="cuda")
    p = C_.c_void_p(t.data_ptr())
    for HW, C, D, S in [(64, 300, 128, 16), (64, 288, 256, 16), (64, 305, 256, 16), (64, 300, 256, 17), (66, 300, 256, 16),
                        (0, 300, 256, 16)]:
        assert lib.goi_codebook_fused(p, p, p, p, None, HW, C, D, S, 1.0, p, p, p, p, None) != 0, (HW, C, D, S)
        assert "goi_codebook_fused" in _lib.last_error()
    # 33 x 47 = 1551 pixels (not a multiple of 4) and 40 codes: the wrapper must still produce the reference's losses
    for H, W, C in [(33, 47, 300), (24, 40, 40)]:
        sem, mlp, lut, gtl = setup(H, W, S=10, C=C, bias=True)
        l0, t0, g0 = grads(codebook_losses, sem, mlp, lut, gtl, 10)
        l1, t1, g1 = grads(fused_codebook_losses, sem, mlp, lut, gtl, 10)
        assert abs(float(l0 - l1)) <= 1e-5 * abs(float(l0))
