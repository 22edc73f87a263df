"""Rasterizer step in the reference's default training configuration (only the semantic features are
trainable): full backward vs the feature-gradient-only backward (GOI_BACKWARD=semantics)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from goi_hyperplane_amd import _lib, rasterizer
from goi_hyperplane_amd.render import GaussianSet, PipelineParams, TorchCamera, render
from goi_hyperplane_amd.scene import HEADLINE, make_camera, make_headline_scene

dev = torch.device("cuda", 0)
_lib.load()
W, H, S = HEADLINE["W"], HEADLINE["H"], HEADLINE["S"]
pc = GaussianSet.from_scene(make_headline_scene(), dev)
for p in pc.parameters():
    p.requires_grad_(False)
pc._semantics.requires_grad_(True)
cams = [TorchCamera(make_camera(W, H, fovx=HEADLINE["fovx"], yaw=0.02 * (i - 4)), dev) for i in range(8)]
bg = torch.zeros(3, device=dev)
up = torch.randn((S, H, W), device=dev) / (W * H)
for lean in (False, True):
    rasterizer.set_backward_mode(semantics_only=lean)
    def step(i):
        pc._semantics.grad = None
        out = render(cams[i % 8], pc, PipelineParams(), bg)
        torch.autograd.backward((out["semantics"],), (up,))
    for i in range(3):
        step(i)
    _lib.profile_collect(); _lib.profile_enable(True)
    for i in range(8):
        step(i)
    torch.cuda.synchronize()
    _lib.profile_enable(False)
    st = _lib.profile_collect()
    torch.cuda.synchronize()
    t = time.perf_counter()
    for i in range(20):
        step(i)
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t) / 20 * 1e3
    print(("feature-gradient-only" if lean else "full backward        "), "%.3f ms/step  %.1f views/s   blend_bwd %.3f  reduce(+preprocess_bwd) %.3f" %
          (ms, 1e3 / ms, st["blend_bwd"][0] / st["blend_bwd"][1], st["preprocess_bwd"][0] / st["preprocess_bwd"][1]))
