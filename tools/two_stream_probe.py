#!/usr/bin/env python3
"""Independent training views on K HIP streams of ONE GPU: does a view's small, latency-bound kernels (sorts, scans) and the
tails of its blend kernels overlap with another view's work?  Headline scene and cameras; every stream has its own parameter
leaves (same values), so gradients are produced per view exactly as in bench.py's step; views/s for K = 1, 2, 3."""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from goi_hyperplane_amd import _lib  # noqa: E402
from goi_hyperplane_amd.render import GaussianSet, PipelineParams, TorchCamera, render  # noqa: E402
from goi_hyperplane_amd.scene import HEADLINE, make_camera, make_scene  # noqa: E402

dev = torch.device("cuda", 0)
_lib.load()
W, H, S = HEADLINE["W"], HEADLINE["H"], HEADLINE["S"]
P_ = int(sys.argv[1]) if len(sys.argv) > 1 else HEADLINE["P"]
sc = make_scene(P_, S=S, sh_degree=3, seed=0, extent=HEADLINE["extent"], log_scale_mean=HEADLINE["log_scale_mean"],
                log_scale_std=HEADLINE["log_scale_std"])
cams = [TorchCamera(make_camera(W, H, fovx=HEADLINE["fovx"], yaw=0.02 * (i - 8), pitch=0.01 * ((i * 7) % 5 - 2)), dev) for i in range(16)]
bg = torch.zeros(3, device=dev)
pipe = PipelineParams()
gen = torch.Generator(device=dev).manual_seed(1234)
g_color = torch.randn((3, H, W), device=dev, generator=gen) / (H * W)
g_sem = torch.randn((S, H, W), device=dev, generator=gen) / (H * W)


def run(K, steps=60, warmup=6):
    pcs = [GaussianSet.from_scene(sc, dev) for _ in range(K)]
    streams = [torch.cuda.Stream(dev) for _ in range(K)]
    torch.cuda.synchronize()

    def step(i):
        k = i % K
        pc = pcs[k]
        with torch.cuda.stream(streams[k]):
            for p in pc.parameters():
                p.grad = None
            out = render(cams[i % len(cams)], pc, pipe, bg)
            torch.autograd.backward((out["render"], out["semantics"]), (g_color, g_sem))
    for i in range(warmup):
        step(i)
    torch.cuda.synchronize()
    t = time.perf_counter()
    for i in range(steps):
        step(warmup + i)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t
    chk = float(sum(float(p.grad.double().abs().sum()) for p in pcs[0].parameters()))
    return steps / dt, dt / steps * 1e3, chk


for K in ((1, 2, 3, 4, 6) if len(sys.argv) <= 2 else tuple(int(x) for x in sys.argv[2].split(","))):
    v, ms, chk = run(K, warmup=int(sys.argv[3]) if len(sys.argv) > 3 else 6)
    print("streams %d: %.1f views/s  (%.3f ms per view)   grad checksum of stream 0's last view %.6e" % (K, v, ms, chk))
