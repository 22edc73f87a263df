"""Times the fused semantic decode (goi_semantic_decode) at 1600x1056, S = 16, 300 codes, and the
GUI frame = render + decode."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from goi_hyperplane_amd.semantic import LinearSVM, SemanticModel, compute_similarity, svm_score_fn

dev = "cuda"
H, W, S, C = 1056, 1600, 16, 300
torch.manual_seed(0)
sem = torch.randn(S, H, W, device=dev)
mlp = SemanticModel(dim_in=S, dim_out=C, num_layer=1, use_bias=True, device=dev)
lut = torch.rand(C, 256, device=dev) * 0.03
svm = LinearSVM().to(dev)
fn = svm_score_fn(svm)
for _ in range(3):
    compute_similarity(sem, mlp, lut, fn, 0.5)
torch.cuda.synchronize()
t = time.perf_counter()
for _ in range(20):
    compute_similarity(sem, mlp, lut, fn, 0.5)
torch.cuda.synchronize()
print("fused decode (incl. the per-call code-score table): %.3f ms" % ((time.perf_counter() - t) / 20 * 1e3))
