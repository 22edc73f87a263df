"""What would a TWO-PASS dL/dSH cost?  The step loop of tools/step_loop.py with the backward in sh_factored mode (preprocess_bwd_k<false>
leaves the clamp-masked colour gradient) followed by goi_raster_sh_grad_from_views for the one view (dL/dSH = basis x gcol), for
rocprofv3 kernel stats (tools/kstats.sh tools/two_pass_dsh.py 30 headline:3000000).  The sum of the two kernels is compared with
preprocess_bwd_k<true> of the default loop."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from goi_hyperplane_amd import _C, rasterizer
from goi_hyperplane_amd.render import GaussianSet, PipelineParams, TorchCamera, render
from goi_hyperplane_amd.scene import make_camera, make_workload

n = int(sys.argv[1]) if len(sys.argv) > 1 else 30
name = sys.argv[2] if len(sys.argv) > 2 else "headline"
dev = torch.device("cuda:0")
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
rasterizer.set_backward_mode(sh_factored=True)
for i in range(n):
    for p in pc.parameters():
        p.grad = None
    cam = cams[i % 16]
    out = render(cam, pc, PipelineParams(), bg)
    torch.autograd.backward((out["render"], out["semantics"]), (gc, gs))
    f = rasterizer.take_sh_factor()
    dsh = _C.sh_grad_from_views(pc._xyz.detach(), f["campos"].reshape(1, 3), f["gcol"].reshape(1, -1, 3), f["degree"], f["M"])
torch.cuda.synchronize()
print("done", n, tuple(dsh.shape))
