"""Within-process interleaved A/B of an option switch on the WHOLE training step (render + backward, headline workload):
ms per step by wall clock between synchronisations, median of the rounds.   usage: python tools/ab_step.py <option> v0 v1 ..."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from goi_hyperplane_amd import _lib
from goi_hyperplane_amd.render import GaussianSet, PipelineParams, TorchCamera, render
from goi_hyperplane_amd.scene import HEADLINE, make_camera, make_scene

opt = sys.argv[1]
values = [int(v) for v in sys.argv[2:]]
h = HEADLINE
dev = torch.device("cuda:0")
sc = make_scene(h["P"], S=h["S"], seed=0, extent=h["extent"], log_scale_mean=h["log_scale_mean"], log_scale_std=h["log_scale_std"])
pc = GaussianSet.from_scene(sc, dev)
cams = [TorchCamera(make_camera(h["W"], h["H"], fovx=h["fovx"], yaw=0.02 * (i - 8), pitch=0.01 * ((i * 7) % 5 - 2)), dev) for i in range(16)]
bg = torch.zeros(3, device=dev)
gen = torch.Generator(device=dev).manual_seed(1234)
inv = 1.0 / (h["W"] * h["H"])
gc = torch.randn((3, h["H"], h["W"]), device=dev, generator=gen) * inv
gs = torch.randn((h["S"], h["H"], h["W"]), device=dev, generator=gen) * inv


def steps(n):
    for i in range(n):
        for p in pc.parameters():
            p.grad = None
        out = render(cams[i % 16], pc, PipelineParams(), bg)
        torch.autograd.backward((out["render"], out["semantics"]), (gc, gs))
    return out


res = {v: [] for v in values}
chk = {}
for rnd in range(7):
    for v in values:
        _lib.set_option(opt, v)
        steps(4)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        out = steps(32)
        torch.cuda.synchronize()
        if rnd:
            res[v].append((time.perf_counter() - t0) / 32 * 1e3)
        chk[v] = (out["render"].double().sum().item(), pc._semantics.grad.double().abs().sum().item(), pc._features.grad.double().abs().sum().item())
for v in values:
    print(f"{opt}={v}: ms/step median {np.median(res[v]):.4f} (min {min(res[v]):.4f}) checksums {chk[v]}")
