"""`simple_knn._C.distCUDA2` over the C ABI (reference: submodules/simple-knn/ext.cpp:15-17,
spatial.cu:15-26).  points [P,3] float32 on the GPU -> [P] float32: mean squared distance to the three
nearest other points, used by scene/gaussian_model.py:147 to size the initial Gaussians."""
from __future__ import annotations

import ctypes as C

import torch

from . import _lib


def distCUDA2(points: torch.Tensor) -> torch.Tensor:
    lib = _lib.load()
    if not points.is_cuda:
        raise RuntimeError("distCUDA2: points must live on a ROCm GPU (cuda device); there is no CPU fallback")
    if points.dtype != torch.float32:
        raise TypeError(f"distCUDA2: points must be float32, got {points.dtype}")  # ref: .data<float>() throws
    P = int(points.size(0))
    dev = points.device
    pts = points.contiguous()
    with torch.cuda.device(dev):
        means = torch.zeros((P,), dtype=torch.float32, device=dev)  # spatial.cu:21: torch::full({P}, 0.0)
        if P == 0:
            return means
        if pts.numel() != 3 * P:
            raise RuntimeError("distCUDA2: points must have shape (P, 3)")
        ws = torch.empty(lib.goi_knn_workspace_bytes(P), dtype=torch.uint8, device=dev)
        r = lib.goi_knn_dist2(P, C.c_void_p(pts.data_ptr()), C.c_void_p(means.data_ptr()), C.c_void_p(ws.data_ptr()),
                              C.c_void_p(torch.cuda.current_stream(dev).cuda_stream))
        if r < 0:
            raise RuntimeError(_lib.last_error())
    return means
