#!/usr/bin/env python3
"""The training half of the semantic head at 1600x1056, 300 codes, D = 256, S = 16: fused_codebook_losses forward + backward
with goi_codebook_fused (five kernels, no [HW, C] matrix) and with the three-kernel path (sim, rows, fp32 dLUT),
ms per call from events on the stream, and the agreement of the two."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from goi_hyperplane_amd import semantic  # noqa: E402
from goi_hyperplane_amd.semantic import SemanticModel, fused_codebook_losses  # noqa: E402

dev = "cuda"
H, W, S, C, D = 1056, 1600, 16, 300, 256
torch.manual_seed(0)
sem = (0.5 * torch.randn(S, H, W, device=dev)).requires_grad_(True)
mlp = SemanticModel(dim_in=S, dim_out=C, num_layer=1, use_bias=True, device=dev)
lut = torch.nn.Parameter(torch.rand(C, D, device=dev) * 0.03)
idx = torch.randint(0, C, (H * W,), device=dev)
gtl = (lut.detach()[idx] * 30 + 0.3 * torch.randn(H * W, D, device=dev)).t().reshape(D, H, W).contiguous()


def run(n=10):
    out = None
    for i in range(3 + n):
        if i == 3:
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
        for p in [sem, lut, *mlp.parameters()]:
            p.grad = None
        loss, terms = fused_codebook_losses(sem, mlp, lut, gtl, 10)
        loss.backward()
        out = (loss.detach(), sem.grad, lut.grad, mlp.layers[0].weight.grad)
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n, out


semantic._FUSED_KERNELS["on"] = True
t2, o2 = run()
semantic._FUSED_KERNELS["on"] = False
t3, o3 = run()
print("goi_codebook_fused (five kernels) %.3f ms   three-kernel path %.3f ms" % (t2, t3))
print("loss %.7f vs %.7f; max |d| / scale: dsem %.2e  dlut %.2e  dW %.2e" % (
    float(o2[0]), float(o3[0]), *[float((a - b).abs().max() / b.abs().max()) for a, b in zip(o2[1:], o3[1:])]))
