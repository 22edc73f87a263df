"""FACTORED mode of the backward (dL/dSH exchanged as its factors in data-parallel training): the masked colour
gradient the backward leaves + goi_raster_sh_grad_from_views reproduce the sum of the per-view dL/dSH bit for bit, and
nothing else changes."""
import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def dev():
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    return torch.device("cuda:0")


@pytest.mark.parametrize("P,S,deg", [(5000, 8, 3), (1200, 16, 1), (3000, 4, 0)])
def test_factored_sh_gradient_is_bit_identical_to_the_sum_over_views(dev, P, S, deg):
    from goi_hyperplane_amd import _C, rasterizer
    from goi_hyperplane_amd.render import GaussianSet, PipelineParams, TorchCamera, render
    from goi_hyperplane_amd.scene import make_camera, make_scene
    from tests.sh_basis_ref import sh_grad_from_views as ref_sh_grad
    W, H = 200, 136
    sc = make_scene(P, S=S, sh_degree=deg, seed=11, log_scale_mean=-2.5)
    pc = GaussianSet.from_scene(sc, dev)
    bg = torch.tensor([0.2, 0.1, 0.4], device=dev)
    cams = [TorchCamera(make_camera(W, H, yaw=0.3 * v, pitch=-0.1 * v), dev) for v in range(3)]
    gen = torch.Generator(device=dev).manual_seed(3)
    ups = [(torch.randn((3, H, W), device=dev, generator=gen), torch.randn((S, H, W), device=dev, generator=gen))
           for _ in cams]

    def backward(cam, up, factored):
        rasterizer.set_backward_mode(sh_factored=factored)
        try:
            for p in pc.parameters():
                p.grad = None
            out = render(cam, pc, PipelineParams(), bg)
            torch.autograd.backward((out["render"], out["semantics"]), up)
            grads = {n: (None if p.grad is None else p.grad.clone()) for n, p in pc.named_parameters()}
            return grads, rasterizer.take_sh_factor()
        finally:
            rasterizer.set_backward_mode(sh_factored=False)

    sh_names = [n for n, _ in pc.named_parameters() if "features" in n]
    assert sh_names
    dsh_sum, gcols, campos = None, [], []
    for cam, up in zip(cams, ups):
        full, none = backward(cam, up, False)
        assert none is None
        lean, factor = backward(cam, up, True)
        assert factor is not None and tuple(factor["gcol"].shape) == (P, 3)
        for n in full:
            if n in sh_names:
                assert lean[n] is None  # autograd gives the SH leaves nothing in this mode
            else:
                assert torch.equal(full[n], lean[n]), n
        dsh = torch.cat([full[n] for n in sh_names], dim=1)
        dsh_sum = dsh if dsh_sum is None else dsh_sum + dsh
        gcols.append(factor["gcol"])
        campos.append(factor["campos"].reshape(3))
    M = dsh_sum.shape[1]
    means = pc.get_xyz.detach()
    got = _C.sh_grad_from_views(means, torch.stack(campos), torch.stack(gcols), deg, M)
    assert float(dsh_sum.abs().max()) > 0
    assert torch.equal(got, dsh_sum)
    ref = ref_sh_grad(means.double(), torch.stack(campos).double(), torch.stack(gcols).double(), deg, M)
    scale = float(ref.abs().max())
    assert float((got.double() - ref).abs().max()) <= 2e-6 * scale
