"""A plain loop of training steps on the headline workload (for rocprofv3 --kernel-trace --stats via tools/kstats.sh).
usage: python tools/step_loop.py [steps] [headline|clustered|closeup][:P]   (e.g. headline:3000000 = BASELINE config 2's size)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from goi_hyperplane_amd.render import GaussianSet, PipelineParams, TorchCamera, render
from goi_hyperplane_amd.scene import HEADLINE, make_camera, make_scene

n = int(sys.argv[1]) if len(sys.argv) > 1 else 40
name = sys.argv[2] if len(sys.argv) > 2 else "headline"  # headline | clustered | closeup (scene.make_workload)
dev = torch.device("cuda:0")
from goi_hyperplane_amd.scene import make_workload
name, _, _p = name.partition(":")
sc, _cam, h = make_workload(name, P=int(_p) if _p else None)
pc = GaussianSet.from_scene(sc, dev)
cams = [TorchCamera(make_camera(h["W"], h["H"], fovx=h["fovx"], yaw=h.get("yaw", 0.0) + 0.02 * (i - 8),
                                pitch=h.get("pitch", 0.0) + 0.01 * ((i * 7) % 5 - 2), distance=h.get("distance", 5.0)), dev) for i in range(16)]
bg = torch.zeros(3, device=dev)
gen = torch.Generator(device=dev).manual_seed(1234)
inv = 1.0 / (h["W"] * h["H"])
gc = torch.randn((3, h["H"], h["W"]), device=dev, generator=gen) * inv
gs = torch.randn((h["S"], h["H"], h["W"]), device=dev, generator=gen) * inv
for i in range(n):
    for p in pc.parameters():
        p.grad = None
    out = render(cams[i % 16], pc, PipelineParams(), bg)
    torch.autograd.backward((out["render"], out["semantics"]), (gc, gs))
torch.cuda.synchronize()
print("done", n)
