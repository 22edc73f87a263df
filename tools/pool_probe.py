"""Which host-side operations keep a pooled gradient buffer from being reused (grad_pool_stats after five steps each): none /
g.untyped_storage() / coalesce_shared_storage / a version bump / all_reduce per tensor / all_reduce of the coalesced span, one RCCL rank.
Found in round 5: untyped_storage() pins the StorageImpl for good (fresh buffer every step), and an all_reduce alone leaves the
version counter where it was (hits: the pool would skip rows other ranks wrote to -- why dist.py bumps the counter itself)."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch.distributed as dist
from goi_hyperplane_amd import _C
from goi_hyperplane_amd.dist import coalesce_shared_storage
from goi_hyperplane_amd.render import GaussianSet, PipelineParams, TorchCamera, render
from goi_hyperplane_amd.scene import make_camera, make_scene
dist.init_process_group(os.environ.get("DBG_BACKEND", "nccl"), rank=0, world_size=1, init_method="tcp://127.0.0.1:29512")
dev = torch.device("cuda:0")
sc = make_scene(20000, S=16, sh_degree=3, seed=5, log_scale_mean=-3.2)
pc = GaussianSet.from_scene(sc, dev)
W, H = 320, 208
cams = [TorchCamera(make_camera(W, H, fovx=0.45, yaw=0.1 * i - 0.4), dev) for i in range(5)]
gc = torch.randn((3, H, W), device=dev) / (W * H)
gs = torch.randn((16, H, W), device=dev) / (W * H)
params = list(pc.parameters())
ext = _C._ext()
def view_grads(cam):
    for p in params:
        p.grad = None
    out = render(cam, pc, PipelineParams(), torch.zeros(3, device=dev))
    torch.autograd.backward((out["render"], out["semantics"]), (gc, gs))
def do(kind):
    grads = [p.grad for p in params if p.grad is not None]
    if kind == "storage":
        for g in grads: g.untyped_storage().data_ptr()
    elif kind == "coalesce":
        coalesce_shared_storage(grads)
    elif kind == "touch":
        for g in coalesce_shared_storage(grads): torch.autograd.graph.increment_version(g)
    elif kind == "allreduce_plain":
        for g in grads: dist.all_reduce(g)
    elif kind == "allreduce_span":
        for g in coalesce_shared_storage(grads): dist.all_reduce(g)
for kind in ("none", "storage", "coalesce", "touch", "allreduce_plain", "allreduce_span"):
    ext.set_grad_pool(False); ext.set_grad_pool(True)
    s0 = ext.grad_pool_stats()
    for i in range(5):
        view_grads(cams[i]); do(kind)
    torch.cuda.synchronize()
    print(kind, [a - b for a, b in zip(ext.grad_pool_stats(), s0)])
