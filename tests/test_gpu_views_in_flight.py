"""render.render_views: a batch of independent views on several HIP streams must give what a loop over the views gives --
every output bit for bit, and the accumulated gradients bit for bit for two views (one addition, commutative) and to rounding
for more (the order of the additions may differ)."""
import pytest
import torch

from goi_hyperplane_amd.render import GaussianSet, PipelineParams, TorchCamera, render, render_views
from goi_hyperplane_amd.scene import make_camera, make_scene

pytestmark = pytest.mark.gpu


def _setup(P=20000, W=320, H=208, S=16, n_views=4):
    dev = torch.device("cuda", 0)
    sc = make_scene(P, S=S, sh_degree=3, seed=3, extent=(2.0, 1.5, 1.0), log_scale_mean=-3.2)
    cams = [TorchCamera(make_camera(W, H, fovx=1.0, yaw=0.05 * i, pitch=0.02 * (i % 3)), dev) for i in range(n_views)]
    gen = torch.Generator(device=dev).manual_seed(5)
    gc = [torch.randn((3, H, W), device=dev, generator=gen) for _ in range(n_views)]
    gs = [torch.randn((S, H, W), device=dev, generator=gen) for _ in range(n_views)]
    return dev, sc, cams, gc, gs


@pytest.mark.parametrize("n_views,streams", [(2, 2), (4, 2), (5, 3)])
def test_views_in_flight_equal_a_loop_over_the_views(n_views, streams):
    dev, sc, cams, gc, gs = _setup(n_views=n_views)
    bg = torch.zeros(3, device=dev)
    pipe = PipelineParams()
    loss = lambda i, o: (o["render"] * gc[i]).sum() + (o["semantics"] * gs[i]).sum() + 0.1 * o["depth"].sum()  # noqa: E731
    pc0 = GaussianSet.from_scene(sc, dev)
    ref = []
    for i, cam in enumerate(cams):
        o = render(cam, pc0, pipe, bg)
        loss(i, o).backward()
        ref.append({k: o[k].detach().clone() for k in ("render", "semantics", "depth", "alpha", "radii")})
    pc1 = GaussianSet.from_scene(sc, dev)
    outs = render_views(cams, pc1, pipe, bg, loss_fn=loss, streams=streams)
    torch.cuda.synchronize()
    for i in range(n_views):
        for k in ref[i]:
            assert torch.equal(ref[i][k], outs[i][k].detach()), (i, k)
    for (name, p0), (_, p1) in zip(pc0.named_parameters(), pc1.named_parameters()):
        assert p0.grad is not None and p1.grad is not None, name
        if n_views == 2:
            assert torch.equal(p0.grad, p1.grad), name
        else:
            scale = float(p0.grad.abs().max())
            assert float((p0.grad - p1.grad).abs().max()) <= 2e-6 * scale, name


def test_views_in_flight_without_a_loss_only_render():
    dev, sc, cams, _gc, _gs = _setup(n_views=3)
    bg = torch.zeros(3, device=dev)
    pc = GaussianSet.from_scene(sc, dev)
    with torch.no_grad():
        ref = [render(c, pc, PipelineParams(), bg)["render"].clone() for c in cams]
        outs = render_views(cams, pc, PipelineParams(), bg, streams=2)
    for r, o in zip(ref, outs):
        assert torch.equal(r, o["render"])
