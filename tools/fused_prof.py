#!/usr/bin/env python3
"""Phase clocks of codebook_simgrad_k at 1600x1056: K loop vs statistics + gradients + plane stores, per wave, from a library
built with -DGOI_FU_PROF (tools/build/exp_fu_prof.sh; that build writes the clocks where the product writes loss sums)."""
import ctypes as C
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from goi_hyperplane_amd import _lib  # noqa: E402

lib = _lib.load()
dev = torch.device("cuda:0")
H, W, Cn, D, S = 1056, 1600, 300, 256, 16
HW = H * W
torch.manual_seed(0)
lut = torch.rand(Cn, D, device=dev) * 0.03
l1 = (lut / lut.norm(dim=1, keepdim=True)).contiguous()
idx = torch.randint(0, Cn, (HW,), device=dev)
g = (lut[idx] * 30 + 0.3 * torch.randn(HW, D, device=dev)).t().contiguous()
sem = 0.5 * torch.randn(S, HW, device=dev)
Wd = torch.randn(Cn, S, device=dev) * 0.25
b = torch.randn(Cn, device=dev) * 0.1
p = lambda t: C.c_void_p(t.data_ptr())  # noqa: E731
stream = C.c_void_p(torch.cuda.current_stream(dev).cuda_stream)
dsem = torch.empty((S, HW), device=dev)
rows = lib.goi_codebook_fused_partial_rows()
partials = torch.zeros((rows, Cn * (S + 1) + 4), device=dev)
part = torch.empty((lib.goi_codebook_dlut_partial_blocks(), 304, D), device=dev)
ws = torch.empty((int(lib.goi_codebook_fused_workspace_bytes(HW)),), dtype=torch.uint8, device=dev)
for _ in range(3):
    assert lib.goi_codebook_fused(p(g), p(l1), p(sem), p(Wd), p(b), HW, Cn, D, S, 1.0, p(dsem), p(partials), p(part), p(ws),
                                  stream) == 0, _lib.last_error()
torch.cuda.synchronize()
# codebook_simgrad_k's per-wave record sits in the workspace after: planes | wz | wt | 6 records | tie masks
al = lambda x: (x + 255) & ~255  # noqa: E731
blocks = (HW + 127) // 128 * 8
npad = 16 * blocks
off = al(2 * 304 * 256 * 2) + al(2 * 304 * 32) + al(2 * 10 * 16 * 64) + 6 * al(npad * 4) + al(npad * 10 * 4)
n_a = (HW + 127) // 128 * 8
rec = ws[off: off + n_a * 16].view(torch.float32).view(n_a, 4).double()
print("codebook_simgrad_k per wave: K loop %.0f clocks, statistics + gradients + stores %.0f clocks" % (rec[:, 0].mean(), rec[:, 1].mean()))
