#!/usr/bin/env python3
"""sim_raw = g^T L1^T at 1600x1056, 300 codes, D = 256: the split-bf16 MFMA kernel (goi_codebook_sim, which also yields
1/|g|) against the library fp32 GEMM + norm kernel it replaces -- time and agreement."""
import ctypes as C
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from goi_hyperplane_amd import _lib  # noqa: E402

lib = _lib.load()
dev = torch.device("cuda:0")
H, W, Cn, D = 1056, 1600, 300, 256
HW = H * W
torch.manual_seed(0)
lut = torch.rand(Cn, D, device=dev) * 0.03
l1 = (lut / lut.norm(dim=1, keepdim=True)).contiguous()
idx = torch.randint(0, Cn, (HW,), device=dev)
g = (lut[idx] * 30 + 0.3 * torch.randn(HW, D, device=dev)).t().contiguous()  # [D, HW] channel-major
p = lambda t: C.c_void_p(t.data_ptr())  # noqa: E731
stream = C.c_void_p(torch.cuda.current_stream(dev).cuda_stream)
sim = torch.empty((HW, Cn), device=dev)
inv = torch.empty((HW,), device=dev)
ws = torch.empty((int(lib.goi_codebook_sim_workspace_bytes()),), dtype=torch.uint8, device=dev)


def ours():
    assert lib.goi_codebook_sim(p(g), p(l1), HW, Cn, D, p(sim), p(inv), p(ws), stream) == 0, _lib.last_error()


def library():
    return torch.linalg.vector_norm(g, dim=0).reciprocal_(), torch.matmul(g.t(), l1.t())


def timed(fn, n=10):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    t = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t) / n * 1e3


ours()
inv_ref, sim_ref = library()
ref64 = (g.double().t()[:200000] @ l1.double().t())
torch.cuda.synchronize()
scale = float(sim_ref.abs().max())
print("max |ours - library fp32| / scale: %.2e   |ours - fp64| %.2e   |library - fp64| %.2e   (first 200k pixels; scale %.3g)" % (
    float((sim - sim_ref).abs().max()) / scale, float((sim[:200000].double() - ref64).abs().max()) / scale,
    float((sim_ref[:200000].double() - ref64).abs().max()) / scale, scale))
print("1/|g|: max rel diff %.2e" % float(((inv - inv_ref) / inv_ref).abs().max()))
print("split-bf16 kernel (sim + 1/|g|): %.3f ms     library: norm + fp32 GEMM %.3f ms" % (timed(ours), timed(library)))
