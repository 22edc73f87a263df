"""Times the fused Adam step against torch.optim.Adam (foreach and fused) on the 7 Gaussian groups."""
import sys, time
sys.path.insert(0, ".")
import torch
from goi_hyperplane_amd.optim import FusedAdam
from tests.test_adam_cpu import GROUPS, make_params, reference_groups

P = 1_000_000
for name, make in (("goi FusedAdam", lambda g: FusedAdam(g, lr=0.0, eps=1e-15)),
                   ("torch Adam (foreach, the reference's)", lambda g: torch.optim.Adam(g, lr=0.0, eps=1e-15)),
                   ("torch Adam (fused=True)", lambda g: torch.optim.Adam(g, lr=0.0, eps=1e-15, fused=True))):
    params = make_params(P, device="cuda")
    opt = make(reference_groups(params))
    for v in params.values():
        v.grad = torch.randn_like(v) * 1e-3
    for _ in range(3):
        opt.step()
    torch.cuda.synchronize()
    t = time.perf_counter()
    for _ in range(20):
        opt.step()
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t) / 20 * 1e3
    elems = sum(v.numel() for v in params.values())
    print(f"{name:40s} {ms:7.3f} ms/step   {elems * 28 / ms / 1e6:8.1f} GB/s of the 28 B/element floor")
