"""cProfile of the host side of training steps (GPU work is asynchronous; a synchronize before every step keeps
GPU waits out of the picture except the one num_rendered wait inside the forward)."""
import cProfile, os, pstats, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from goi_hyperplane_amd import _lib
from goi_hyperplane_amd.render import GaussianSet, PipelineParams, TorchCamera, render
from goi_hyperplane_amd.scene import HEADLINE, make_camera, make_headline_scene

dev = torch.device("cuda", 0)
_lib.load()
W, H, S = HEADLINE["W"], HEADLINE["H"], HEADLINE["S"]
pc = GaussianSet.from_scene(make_headline_scene(), dev)
params = list(pc.parameters())
cams = [TorchCamera(make_camera(W, H, fovx=HEADLINE["fovx"], yaw=0.02 * (i - 4)), dev) for i in range(8)]
bg = torch.zeros(3, device=dev)
pipe = PipelineParams()
g_c = torch.randn((3, H, W), device=dev) / (W * H)
g_s = torch.randn((S, H, W), device=dev) / (W * H)


def step(i):
    for p in params:
        p.grad = None
    out = render(cams[i % 8], pc, pipe, bg)
    torch.autograd.backward((out["render"], out["semantics"]), (g_c, g_s))


for i in range(5):
    step(i)
torch.cuda.synchronize()
# pure host time per step with an idle GPU at the start of every step
ts = []
for i in range(20):
    torch.cuda.synchronize()
    t = time.perf_counter()
    step(i)
    ts.append(time.perf_counter() - t)
torch.cuda.synchronize()
print("host time per step (GPU idle at start, includes the in-forward wait): %.0f us median" % (sorted(ts)[10] * 1e6))
pr = cProfile.Profile()
pr.enable()
for i in range(50):
    step(i)
torch.cuda.synchronize()
pr.disable()
st = pstats.Stats(pr)
st.sort_stats("tottime").print_stats(22)
