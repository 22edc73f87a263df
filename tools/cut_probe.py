"""Depth cut-off probe: instance counts and stage times of cut vs uncut frames on the headline workload (16 cameras)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from goi_hyperplane_amd import _C, _lib, rasterizer
from goi_hyperplane_amd.render import GaussianSet, PipelineParams, TorchCamera, render
from goi_hyperplane_amd.scene import make_camera, make_workload

name = sys.argv[1] if len(sys.argv) > 1 else "headline"
dev = torch.device("cuda:0")
sc, _cam, h = make_workload(name)
pc = GaussianSet.from_scene(sc, dev)
cams = [TorchCamera(make_camera(h["W"], h["H"], fovx=h["fovx"], yaw=h.get("yaw", 0.0) + 0.02 * (i - 8), pitch=h.get("pitch", 0.0) + 0.01 * ((i * 7) % 5 - 2), distance=h.get("distance", 5.0)), dev) for i in range(16)]
bg = torch.zeros(3, device=dev)
gen = torch.Generator(device=dev).manual_seed(1234)
inv = 1.0 / (h["W"] * h["H"])
gc = torch.randn((3, h["H"], h["W"]), device=dev, generator=gen) * inv
gs = torch.randn((h["S"], h["H"], h["W"]), device=dev, generator=gen) * inv

def steps(n, collect=None):
    for i in range(n):
        for p in pc.parameters():
            p.grad = None
        out = render(cams[i % 16], pc, PipelineParams(), bg)
        if collect is not None:
            collect.append(rasterizer.last_num_rendered())
        torch.autograd.backward((out["render"], out["semantics"]), (gc, gs))

for mode in (False, True):
    rasterizer.set_forward_mode(speculative=True, depth_cut=mode)
    steps(40)
    torch.cuda.synchronize()
    ns = []
    _lib.profile_collect(); _lib.profile_enable(True)
    steps(32, ns)
    torch.cuda.synchronize()
    _lib.profile_enable(False)
    st = _lib.profile_collect()
    print("depth_cut", mode, "N per frame", int(np.mean([int(x) for x in ns])), "cut frames", sum(1 for x in ns if getattr(x, "cut_key", None) is not None),
          {k: round(ms / c, 4) for k, (ms, c) in st.items() if c}, "sum", round(sum(ms / c for ms, c in st.values() if c), 4), rasterizer.speculation_stats())
