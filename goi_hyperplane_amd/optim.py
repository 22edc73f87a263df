"""Fused Adam for the Gaussian parameter groups.

`FusedAdam` takes the same constructor arguments, keeps the same `param_groups` / `state` layout and
therefore the same `state_dict()` format as `torch.optim.Adam`, which is what the reference builds in
scene/gaussian_model.py:163-253 (`torch.optim.Adam(l, lr=0.0, eps=1e-15)` over named groups) and
serialises in `capture()` / `restore()`: checkpoints move in both directions.  `step()` runs ONE HIP
kernel over every group (csrc/adam.hip) instead of torch's ~10 foreach kernels per state tensor;
`step(nograd_mask=...)` also applies gui/main.py:480-513's per-Gaussian gradient mask on the fly.

Not supported (the reference uses none of them): weight_decay, amsgrad, maximize, capturable,
differentiable.  fp32 CUDA (ROCm) tensors only; there is no CPU fallback.
"""
from __future__ import annotations

import ctypes as C
import math

import torch

from . import _lib


class FusedAdam(torch.optim.Optimizer):
    def __init__(self, params, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=0, amsgrad=False):
        if weight_decay != 0 or amsgrad:
            raise NotImplementedError("FusedAdam covers the reference's configuration: no weight decay, no amsgrad")
        if not 0.0 <= lr or not 0.0 <= eps or not (0.0 <= betas[0] < 1.0) or not (0.0 <= betas[1] < 1.0):
            raise ValueError("invalid Adam hyper-parameter")
        defaults = dict(lr=lr, betas=betas, eps=eps, weight_decay=0, amsgrad=False, maximize=False, foreach=None,
                        capturable=False, differentiable=False, fused=None)
        super().__init__(params, defaults)

    @torch.no_grad()
    def step(self, closure=None, nograd_mask: torch.Tensor | None = None, skip_if: torch.Tensor | None = None):
        """One Adam update of every parameter that has a gradient.  nograd_mask: optional [P] bool/uint8
        tensor; Gaussians with a non-zero entry are updated as if their gradient rows were zero.
        skip_if: optional int32[1] DEVICE tensor read by the kernel: non-zero makes this step a no-op on the device
        (parameters and both moments untouched).  Pass rasterizer.truncated_flag() to skip the view of a truncated
        speculative forward without any host round trip -- or rasterizer.truncated_flag(accumulated=True) when the step
        accumulates several views (a truncated EARLIER view must skip it too).  The skip is opt-in: without skip_if a
        truncated frame's zero gradients still move every parameter by its momentum and decay both moments.
        Known approximation: the step count lives on the host (as in torch.optim.Adam) and advances whether or not the
        device skipped, so after k skipped steps the bias corrections 1 - beta^t are those of step t rather than t - k --
        exact from a few hundred steps on (beta2^t -> 0), at most a factor 1 - beta2^(t-k) / 1 - beta2^t off before
        (k = 1 at t = 10, beta2 = 0.999: 10 % on the second-moment correction of that one step)."""
        loss = None
        if closure is not None:
            with torch.enable_grad():
                loss = closure()
        if nograd_mask is not None and nograd_mask.dim() != 1:
            raise ValueError("nograd_mask must be a 1-D [P] tensor")
        if skip_if is not None and (not skip_if.is_cuda or skip_if.dtype not in (torch.int32, torch.uint32)
                                    or skip_if.numel() != 1):
            raise ValueError("skip_if must be a 1-element int32 tensor on the GPU")
        lib = _lib.load()
        launches = {}  # (beta1, beta2, eps, device) -> list of GoiAdamGroup
        keep = []
        for group in self.param_groups:
            beta1, beta2 = group["betas"]
            for p in group["params"]:
                if p.grad is None:
                    continue
                if not p.is_cuda:
                    raise RuntimeError("FusedAdam: parameters must live on a ROCm GPU; there is no CPU fallback")
                if p.dtype != torch.float32 or p.grad.dtype != torch.float32 or p.grad.is_sparse:
                    raise TypeError("FusedAdam: dense float32 parameters and gradients only")
                if not p.is_contiguous():
                    raise RuntimeError("FusedAdam: parameters must be contiguous")
                state = self.state[p]
                if len(state) == 0:
                    state["step"] = torch.tensor(0.0, dtype=torch.float32)  # torch keeps a host-side fp32 scalar
                    state["exp_avg"] = torch.zeros_like(p, memory_format=torch.preserve_format)
                    state["exp_avg_sq"] = torch.zeros_like(p, memory_format=torch.preserve_format)
                state["step"] += 1
                t = float(state["step"])
                grad = p.grad.contiguous()
                keep.append(grad)
                step_size = group["lr"] / (1.0 - beta1 ** t)
                bc2_sqrt = math.sqrt(1.0 - beta2 ** t)
                row_len = p.numel() // p.shape[0] if p.dim() > 0 and p.shape[0] > 0 else 1
                if nograd_mask is not None:
                    if p.dim() == 0:
                        raise ValueError("nograd_mask: a 0-dim parameter has no per-Gaussian rows to mask")
                    if p.shape[0] != nograd_mask.shape[0]:
                        raise ValueError("nograd_mask must have one entry per Gaussian (parameter rows)")
                launches.setdefault((beta1, beta2, group["eps"], p.device), []).append(_lib.GoiAdamGroup(
                    p.data_ptr(), grad.data_ptr(), state["exp_avg"].data_ptr(), state["exp_avg_sq"].data_ptr(),
                    p.numel(), max(row_len, 1), step_size, bc2_sqrt))
        for (beta1, beta2, eps, dev), groups in launches.items():
            # the kernel dereferences the mask on the launch device: a CPU (or other-GPU) mask is copied there, never
            # handed over as a foreign pointer
            mask = None if nograd_mask is None else nograd_mask.to(device=dev, dtype=torch.uint8).contiguous()
            with torch.cuda.device(dev):
                stream = C.c_void_p(torch.cuda.current_stream(dev).cuda_stream)
                for i in range(0, len(groups), _lib.ADAM_MAX_GROUPS):
                    chunk = groups[i:i + _lib.ADAM_MAX_GROUPS]
                    arr = (_lib.GoiAdamGroup * len(chunk))(*chunk)
                    mp = C.c_void_p(mask.data_ptr()) if mask is not None else None
                    sp = None
                    if skip_if is not None:
                        if skip_if.device != dev:
                            raise ValueError("skip_if lives on another device than the parameters")
                        sp = C.c_void_p(skip_if.data_ptr())
                    if lib.goi_adam_step_guarded(arr, len(chunk), beta1, beta2, eps, mp, sp, stream) < 0:
                        raise RuntimeError(_lib.last_error())
        return loss
