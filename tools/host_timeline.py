#!/usr/bin/env python3
"""Host-side timeline of a training step (no profiler attached): wall-clock spent in each host phase
while the GPU runs asynchronously.  Tells whether the step is GPU-bound or enqueue-bound."""
import sys
import time

import numpy as np
import torch

sys.path.insert(0, ".")
from goi_hyperplane_amd import _C, _lib  # noqa: E402
from goi_hyperplane_amd.render import GaussianSet, PipelineParams, TorchCamera, render  # noqa: E402
from goi_hyperplane_amd.scene import HEADLINE, make_camera, make_scene  # noqa: E402

dev = torch.device("cuda", 0)
_lib.load()
sc = make_scene(HEADLINE["P"], S=HEADLINE["S"], sh_degree=3, seed=0, extent=HEADLINE["extent"],
                log_scale_mean=HEADLINE["log_scale_mean"], log_scale_std=HEADLINE["log_scale_std"])
pc = GaussianSet.from_scene(sc, dev)
params = [pc._xyz, pc._features, pc._semantics, pc._opacity, pc._scaling, pc._rotation]
W, H = HEADLINE["W"], HEADLINE["H"]
cams = [TorchCamera(make_camera(W, H, fovx=HEADLINE["fovx"], yaw=0.02 * (i - 8), pitch=0.01 * ((i * 7) % 5 - 2)), dev)
        for i in range(16)]
bg = torch.zeros(3, device=dev)
pipe = PipelineParams()
inv = 1.0 / (W * H)
rec = []
for i in range(30):
    t0 = time.perf_counter()
    for p in params:
        p.grad = None
    t1 = time.perf_counter()
    out = render(cams[i % 16], pc, pipe, bg)
    t2 = time.perf_counter()
    loss = (out["render"].sum() + out["semantics"].sum()) * inv
    t3 = time.perf_counter()
    loss.backward()
    t4 = time.perf_counter()
    rec.append((t1 - t0, t2 - t1, t3 - t2, t4 - t3))
torch.cuda.synchronize()
r = np.array(rec[10:]) * 1e6
print("host us/step: zero_grad %.0f  render(incl. N sync) %.0f  loss %.0f  backward %.0f  total %.0f" %
      (*r.mean(0), r.sum(1).mean()))
# forward alone, host time with a preceding sync (pure enqueue + N sync latency on an idle GPU)
with torch.no_grad():
    ts = []
    for i in range(20):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        out = render(cams[i % 16], pc, pipe, bg)
        t1 = time.perf_counter()
        torch.cuda.synchronize()
        t2 = time.perf_counter()
        ts.append((t1 - t0, t2 - t0))
    ts = np.array(ts[5:]) * 1e6
    print("forward from idle: host returns after %.0f us, GPU done after %.0f us" % tuple(ts.mean(0)))
