import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from goi_hyperplane_amd.semantic import SemanticModel, fused_codebook_losses
dev = "cuda"
H, W, S, C, D = 1056, 1600, 16, 300, 256
torch.manual_seed(0)
sem = torch.randn(S, H, W, device=dev, requires_grad=True)
mlp = SemanticModel(dim_in=S, dim_out=C, num_layer=1, use_bias=True, device=dev)
lut = torch.nn.Parameter(torch.rand(C, D, device=dev) * 0.03)
gtl = torch.randn(D, H, W, device=dev)
for it in range(6):
    loss, _ = fused_codebook_losses(sem, mlp, lut, gtl, 10)
    loss.backward()
torch.cuda.synchronize()
