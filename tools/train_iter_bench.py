"""One iteration of the reference's semantic-field training loop (train.py:112-199) on the headline
scene: render -> code-book losses -> backward -> Adam step, with this build's pieces
(rasterizer, fused losses, FusedAdam) and with the PyTorch pieces the reference uses around the
same rasterizer (unfused losses, torch.optim.Adam).  Prints ms per iteration for both."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from goi_hyperplane_amd import _lib
from goi_hyperplane_amd.optim import FusedAdam
from goi_hyperplane_amd.render import GaussianSet, PipelineParams, TorchCamera, render
from goi_hyperplane_amd.scene import HEADLINE, make_camera, make_scene
from goi_hyperplane_amd.semantic import SemanticModel, codebook_losses, fused_codebook_losses

dev = torch.device("cuda", 0)
_lib.load()
W, H, S = HEADLINE["W"], HEADLINE["H"], HEADLINE["S"]
sc = make_scene(HEADLINE["P"], S=S, sh_degree=3, seed=0, extent=HEADLINE["extent"],
                log_scale_mean=HEADLINE["log_scale_mean"], log_scale_std=HEADLINE["log_scale_std"])
cams = [TorchCamera(make_camera(W, H, fovx=HEADLINE["fovx"], yaw=0.02 * (i - 4)), dev) for i in range(8)]
bg = torch.zeros(3, device=dev)
pipe = PipelineParams()
gtl = torch.randn(256, H, W, device=dev)


def run(fused: bool, n=8):
    from goi_hyperplane_amd import rasterizer
    rasterizer.set_backward_mode(semantics_only=fused)  # this build: feature-gradient-only backward
    pc = GaussianSet.from_scene(sc, dev)
    for p in pc.parameters():  # the reference's default: semantic_finetune only (arguments/__init__.py:85-90)
        p.requires_grad_(False)
    pc._semantics.requires_grad_(True)
    mlp = SemanticModel(dim_in=S, dim_out=300, num_layer=1, use_bias=True, device=dev)
    lut = torch.nn.Parameter(torch.rand(300, 256, device=dev) * 0.03)
    Adam = FusedAdam if fused else torch.optim.Adam
    opts = [Adam([{"params": [pc._semantics], "lr": 5e-3, "name": "semantics"}], lr=0.0, eps=1e-15),
            Adam(mlp.parameters(), lr=0.003), Adam([lut], lr=0.001)]
    loss_fn = fused_codebook_losses if fused else codebook_losses

    def it(i):
        out = render(cams[i % len(cams)], pc, pipe, bg)
        loss, _ = loss_fn(out["semantics"], mlp, lut, gtl, 10 + i)
        loss.backward()
        for o in opts:
            o.step()
            o.zero_grad(set_to_none=True)
    for i in range(len(cams) + 1):  # every camera once (with the geometry cache on: the misses)
        it(i)
    torch.cuda.synchronize()
    t = time.perf_counter()
    for i in range(n):
        it(i)
    torch.cuda.synchronize()
    return (time.perf_counter() - t) / n * 1e3


from goi_hyperplane_amd import rasterizer  # noqa: E402
a = run(True)
rasterizer.set_geometry_cache(64 << 30)  # opt-in: only the semantic features train here, the 8 cameras repeat
ac = run(True, n=16)
rasterizer.set_geometry_cache(0)
b = run(False)
print(f"train iteration with the opt-in geometry cache (cameras repeat, geometry frozen): {ac:.2f} ms")
print(f"train iteration (1M Gaussians, {W}x{H}, S={S}, 300 codes): this build {a:.2f} ms   "
      f"same rasterizer + PyTorch losses + torch Adam {b:.2f} ms   ratio {b / a:.1f}x")
