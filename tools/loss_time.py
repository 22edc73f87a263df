"""Times the (unfused, PyTorch) training losses of train.py:142-163 at the headline resolution."""
import sys, time
sys.path.insert(0, ".")
import torch
from goi_hyperplane_amd.semantic import SemanticModel, codebook_losses, fused_codebook_losses

dev = "cuda"
H, W, S, C, D = 1056, 1600, 16, 300, 256
torch.manual_seed(0)
sem = torch.randn(S, H, W, device=dev, requires_grad=True)
mlp = SemanticModel(dim_in=S, dim_out=C, num_layer=1, use_bias=True, device=dev)
lut = torch.nn.Parameter(torch.rand(C, D, device=dev) * 0.03)
gtl = torch.randn(D, H, W, device=dev)
for it in range(3):
    loss, _ = codebook_losses(sem, mlp, lut, gtl, 10)
    loss.backward()
torch.cuda.synchronize()
t = time.perf_counter()
n = 5
for it in range(n):
    loss, _ = codebook_losses(sem, mlp, lut, gtl, 10)
    loss.backward()
torch.cuda.synchronize()
print("codebook_losses fwd+bwd: %.2f ms/iter, peak mem %.1f GB" % ((time.perf_counter() - t) / n * 1e3,
      torch.cuda.max_memory_allocated() / 1e9))
torch.cuda.synchronize()
t = time.perf_counter()
for it in range(n):
    with torch.no_grad():
        g = gtl.permute(1, 2, 0).reshape(-1, D)
        s = (g / g.norm(dim=1, keepdim=True)) @ (lut / lut.norm(dim=1, keepdim=True)).T
torch.cuda.synchronize()
print("  of which normalise + [HW,256]x[256,300] GEMM: %.2f ms" % ((time.perf_counter() - t) / n * 1e3))

for it in range(3):
    loss, _ = fused_codebook_losses(sem, mlp, lut, gtl, 10)
    loss.backward()
torch.cuda.synchronize()
torch.cuda.reset_peak_memory_stats()
t = time.perf_counter()
for it in range(n):
    loss, _ = fused_codebook_losses(sem, mlp, lut, gtl, 10)
    loss.backward()
torch.cuda.synchronize()
print("fused_codebook_losses fwd+bwd: %.2f ms/iter, peak mem %.1f GB" % ((time.perf_counter() - t) / n * 1e3,
      torch.cuda.max_memory_allocated() / 1e9))
