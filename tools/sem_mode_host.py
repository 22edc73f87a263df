"""Host cost of the reference's DEFAULT training mode (only the semantic features trainable, arguments/__init__.py:85-90)
with the geometry cache on: wall clock per step, pure host time per step (GPU idle at the start), cProfile top functions.
usage: python tools/sem_mode_host.py [--profile]"""
import cProfile, os, pstats, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
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
cams = [TorchCamera(make_camera(W, H, fovx=HEADLINE["fovx"], yaw=0.02 * (i - 8), pitch=0.01 * ((i * 7) % 5 - 2)), dev) for i in range(16)]
bg = torch.zeros(3, device=dev)
pipe = PipelineParams()
g_sem = torch.randn((S, H, W), device=dev) / (W * H)
rasterizer.set_backward_mode(semantics_only=True)
rasterizer.set_geometry_cache(64 << 30)


def step(i):
    pc._semantics.grad = None
    out = render(cams[i % 16], pc, pipe, bg)
    torch.autograd.backward((out["semantics"],), (g_sem,))


for i in range(20):
    step(i)
torch.cuda.synchronize()
for rep in range(3):
    t = time.perf_counter()
    for i in range(64):
        step(i)
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t) / 64 * 1e3
    print("wall %.3f ms/step = %.0f views/s" % (ms, 1e3 / ms))
ts = []
for i in range(40):
    torch.cuda.synchronize()
    t = time.perf_counter()
    step(i)
    ts.append(time.perf_counter() - t)
torch.cuda.synchronize()
print("host enqueue per step (GPU idle at start): median %.0f us, p10 %.0f, p90 %.0f" % tuple(np.percentile(np.array(ts) * 1e6, [50, 10, 90])))
# GPU time per step: enqueue 64 steps, measure with events around
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
_lib.profile_collect(); _lib.profile_enable(True)
for i in range(16):
    step(i)
torch.cuda.synchronize()
_lib.profile_enable(False)
st = _lib.profile_collect()
print("stage ms/step:", {k: round(ms_ / n, 4) for k, (ms_, n) in st.items() if n})
if "--profile" in sys.argv:
    pr = cProfile.Profile()
    for i in range(40):
        torch.cuda.synchronize()
        pr.enable()
        step(i)
        pr.disable()
    ps = pstats.Stats(pr)
    ps.sort_stats("cumulative").print_stats(35)
    ps.sort_stats("tottime").print_stats(25)
