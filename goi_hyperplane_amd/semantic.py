"""Semantic head around the rasterizer: the code-book classifier that turns the
rendered S-dim feature into one of `tab_len` codes, the 256-d code book (LUT), the hyperplane
(LinearSVM) score, and the training losses that tie them together.

Mirrors, with the same names / constructor arguments / file formats so that the reference's
checkpoints load unchanged:
    SemanticModel            scene/semantic_model.py:13-63   ({"args", "state_dict"} save format)
    LinearSVM.forward        networks.py:12-59               (x / 0.3438 -> Linear(256, 1))
    compute_similarity       gui/main.py:364-386             (inference decode)
    codebook_losses          train.py:142-163                (training losses)

Inference (`compute_similarity`) is the hot part at GUI frame rate: it runs as ONE fused HIP kernel
(csrc/semantic_head.hip: fp32 MFMA contraction + argmax + per-code score lookup) that reads the
rasterizer's [S, H, W] output directly.  The training losses exist twice: `codebook_losses` restates
train.py line by line in PyTorch (the parity reference: ~40 kernels over [HW,300] tensors, 109 ms
and 21 GB at 1600x1056 on MI355X); `fused_codebook_losses` is the product path.  For the reference's shapes (256-d
features, 289..304 codes, S <= 16) that is `goi_codebook_fused`: four hand-written kernels of csrc/codebook_loss.hip,
no [HW, C] fp32 matrix in memory (the similarity and its gradient exist as MFMA tiles and two bf16 planes), ~3.4 ms and
8.6 GB at 1600x1056.  Other shapes take the three-kernel path (similarity kernel or library GEMM, one row kernel, a
split-K MFMA GEMM for dL/dLUT), which is also the cross-check of tests/test_gpu_losses.py; LOSS_PATH_COUNTS says which ran.
"""
from __future__ import annotations

import ctypes as C
import ctypes as C_

import torch
import torch.nn.functional as F

from . import _lib


class FeatureNorm(torch.nn.Module):
    def forward(self, x):
        return x / x.norm(dim=-1, keepdim=True)


class SemanticModel(torch.nn.Module):
    """num_layer x Linear(+ReLU); the reference instantiates it as a single Linear(sem_dim -> tab_len,
    bias=True) (train.py:64)."""

    def __init__(self, dim_in=64, dim_hidden=128, dim_out=40, num_layer=3, device="cuda", use_bias=False, norm=False):
        super().__init__()
        self.dim_in, self.dim_hidden, self.dim_out, self.num_layer, self.device = dim_in, dim_hidden, dim_out, num_layer, device
        self.args = {"dim_in": dim_in, "dim_hidden": dim_hidden, "dim_out": dim_out, "num_layer": num_layer,
                     "device": device, "use_bias": use_bias, "norm": norm}
        layers = []
        for ind in range(num_layer):
            d_in = dim_in if ind == 0 else dim_hidden
            d_out = dim_out if ind == num_layer - 1 else dim_hidden
            layer = torch.nn.Linear(d_in, d_out, device=device, bias=use_bias)
            torch.nn.init.xavier_uniform_(layer.weight.data)
            act = torch.nn.ReLU() if ind < num_layer - 1 else (FeatureNorm() if norm else torch.nn.Identity())
            layers.extend([layer, act])
        self.layers = torch.nn.Sequential(*layers)

    def forward(self, semantic_features):
        return self.layers(semantic_features)

    @staticmethod
    def load(path, map_location=None):
        pth = torch.load(path, map_location=map_location)
        model = SemanticModel(**pth["args"])
        model.load_state_dict(pth["state_dict"])
        return model

    def save(self, path):
        torch.save({"args": self.args, "state_dict": self.state_dict()}, path)


class LinearSVM(torch.nn.Module):
    """The hyperplane of the paper: one Linear(input_dim -> 1) applied to x / 0.3438 (networks.py:12-59)."""

    def __init__(self, set_bias=0.86, input_dim=256):
        super().__init__()
        self.linear = torch.nn.Linear(input_dim, 1)
        b = torch.tensor(set_bias)
        torch.nn.init.constant_(self.linear.bias, float(2 - torch.log(b / (1 - b))))

    def forward(self, x):
        return self.linear(x / 0.3438)


@torch.no_grad()
def code_scores(lut: torch.Tensor, score_fn) -> torch.Tensor:
    """Everything after the argmax of gui/main.py:364-386 depends on the code only: fold
    LUT[c] -> L2 normalise -> score_fn (LinearSVM + sigmoid, or a VLM similarity) into a table."""
    normed = lut / lut.norm(dim=-1, keepdim=True)
    return score_fn(normed).reshape(-1).float().contiguous()


def svm_score_fn(svm: LinearSVM):
    return lambda feat: svm(feat).squeeze(-1).sigmoid()


@torch.no_grad()
def compute_similarity(sem_chw: torch.Tensor, mlp: SemanticModel, lut: torch.Tensor, score_fn, thresh: float = 0.5,
                       out_bg_mask: torch.Tensor | None = None, return_index: bool = False):
    """Fused decode of a rendered semantic map `sem_chw` [S, H, W] (the rasterizer's output, NOT
    permuted): returns sim[H*W] with background (sim < thresh) zeroed, like the reference's
    compute_similarity(embedding_feature=[HW,S]).  Runs only on the GPU through libgoi_raster.so."""
    if mlp.num_layer != 1:
        raise NotImplementedError("the fused decode covers the reference's configuration: one Linear(S -> tab_len)")
    if not sem_chw.is_cuda:
        raise RuntimeError("goi_hyperplane_amd.semantic: tensors must live on a ROCm GPU; there is no CPU fallback")
    lib = _lib.load()
    lin = mlp.layers[0]
    S = int(sem_chw.shape[0])
    HW = int(sem_chw[0].numel())
    dev = sem_chw.device
    n_codes = int(lin.weight.shape[0])
    if lin.weight.shape[1] != S or lut.shape[0] != n_codes:
        raise ValueError("shape mismatch between features, MLP and LUT")
    sem = sem_chw.contiguous().float()
    w = lin.weight.detach().contiguous().float()
    b = (lin.bias.detach() if lin.bias is not None else torch.zeros(n_codes, device=dev)).contiguous().float()
    table = code_scores(lut, score_fn).to(dev)
    sim = torch.empty(HW, dtype=torch.float32, device=dev)
    idx = torch.empty(HW, dtype=torch.int32, device=dev) if return_index else None
    mask = torch.empty(HW, dtype=torch.uint8, device=dev) if out_bg_mask is not None else None
    p = lambda t: None if t is None else C.c_void_p(t.data_ptr())  # noqa: E731
    with torch.cuda.device(dev):
        r = lib.goi_semantic_decode(p(sem), S, HW, p(w), p(b), n_codes, p(table), float(thresh), p(sim), p(idx), p(mask),
                                    C.c_void_p(torch.cuda.current_stream(dev).cuda_stream))
    if r < 0:
        raise RuntimeError(_lib.last_error())
    if out_bg_mask is not None:
        out_bg_mask[:] = mask.bool()
    return (sim, idx) if return_index else sim


@torch.no_grad()
def compute_similarity_reference(embedding_feature: torch.Tensor, mlp, lut, score_fn, thresh: float = 0.5,
                                 out_bg_mask=None):
    """The unfused restatement of gui/main.py:364-386 (argmax over softmax(10 * dec), LUT gather,
    normalise, score, threshold) on [HW, S] features.  Used by the tests as the torch reference."""
    dec_feature = mlp(embedding_feature)
    sem_logit = torch.softmax(dec_feature * 10, dim=-1).argmax(dim=-1)
    sem_feature = lut[sem_logit]
    normed_feature = sem_feature / sem_feature.norm(dim=-1, keepdim=True)
    sim = score_fn(normed_feature).reshape(-1)
    bg = sim < thresh
    if out_bg_mask is not None:
        out_bg_mask[:] = bg
    sim = sim.clone()
    sim[bg] = 0
    return sim, sem_logit


def codebook_losses(sem_feature_chw: torch.Tensor, semantic_mlp: SemanticModel, lut: torch.Tensor, gtl_chw: torch.Tensor,
                    iteration: int):
    """Training losses of train.py:142-163.  sem_feature_chw [S,H,W] is the rasterizer output
    (gradients flow back into it), gtl_chw [ape_dim,H,W] the per-view ground-truth APE feature.
    Returns (loss, dict of the four terms)."""
    S = sem_feature_chw.shape[0]
    sem_feature = sem_feature_chw.permute(1, 2, 0).reshape(-1, S)
    sem_label = torch.softmax(semantic_mlp(sem_feature), dim=-1)
    gtl = gtl_chw.permute(1, 2, 0).reshape(-1, gtl_chw.shape[0]).float()
    gtl = gtl / gtl.norm(dim=1, keepdim=True)
    lut1 = lut / lut.norm(dim=1, keepdim=True)
    sim = gtl @ lut1.T  # [HW, tab_len]: the dense code book x feature contraction (library GEMM)
    sim_val = sim.max(dim=1, keepdim=True)[0]
    label = (sim == sim_val).float().detach()
    lab = F.mse_loss(sem_label, label) * 50
    sl = 1 - sim_val.mean()
    recc = 1 - F.cosine_similarity(lut[sem_label.argmax(-1)], gtl, dim=-1).mean()
    t = 1 if iteration < 1000 else 2
    anneal = sim * t
    sl1 = -1.0 * (torch.softmax(anneal, dim=1) * torch.log_softmax(anneal, dim=1)).sum(dim=-1).mean()
    loss = lab + sl + 0.3 * sl1 + recc
    return loss, {"lab": lab, "sl": sl, "sl1": sl1, "recc": recc}


# which implementation the last loss evaluations took, counted (tests assert on it: the product path for the reference's shapes
# is "fused"; "three_kernel" / "library_gemm" are the paths for other shapes)
LOSS_PATH_COUNTS = {"fused": 0, "three_kernel": 0, "library_gemm": 0}
_SIM_KERNEL = {"on": True}  # False: the library fp32 GEMM for sim (A/B and a fallback for other shapes)
_FUSED_KERNELS = {"on": True}  # False: the three-kernel path (sim, rows, dLUT) -- kept for other shapes and as a cross-check


class _FusedCodebookLoss(torch.autograd.Function):
    """loss = lab + sl + 0.3 sl1 + recc of train.py:142-163, with all gradients produced in the forward
    (the loss is a scalar: backward only scales them)."""

    @staticmethod
    def forward(ctx, sem_chw, weight, bias, lut1, gtl_chw, t):
        lib = _lib.load()
        if not sem_chw.is_cuda:
            raise RuntimeError("goi_hyperplane_amd.semantic: tensors must live on a ROCm GPU; there is no CPU fallback")
        dev = sem_chw.device
        S = int(sem_chw.shape[0])
        HW = int(sem_chw[0].numel())
        C, D = int(lut1.shape[0]), int(lut1.shape[1])
        sem = sem_chw.detach().contiguous().float().view(S, HW)
        g = gtl_chw.detach().contiguous().float().view(D, HW)  # [D, HW]: used as g^T through views, never copied
        w = weight.detach().contiguous().float()
        b = None if bias is None else bias.detach().contiguous().float()
        l1 = lut1.detach().contiguous().float()
        with torch.cuda.device(dev):
            p = lambda x: None if x is None else C_.c_void_p(x.data_ptr())  # noqa: E731
            stream = C_.c_void_p(torch.cuda.current_stream(dev).cuda_stream)
            # (the full shape predicate of launch_codebook_fused, csrc/codebook_loss.hip: anything else takes the
            # three-kernel path below instead of raising)
            if (_FUSED_KERNELS["on"] and D == 256 and 288 < C <= 304 and 1 <= S <= 16 and HW % 4 == 0
                    and 4 <= HW < (1 << 25)):
                # sim -> losses -> gradients in two kernels, no [HW, C] fp32 matrix (csrc/codebook_loss.hip: codebook_fused_k)
                dsem = torch.empty((S, HW), dtype=torch.float32, device=dev)
                partials = torch.empty((lib.goi_codebook_fused_partial_rows(), C * (S + 1) + 4), dtype=torch.float32, device=dev)
                part = torch.empty((lib.goi_codebook_dlut_partial_blocks(), 304, D), dtype=torch.float32, device=dev)
                ws = torch.empty((int(lib.goi_codebook_fused_workspace_bytes(HW)),), dtype=torch.uint8, device=dev)
                if lib.goi_codebook_fused(p(g), p(l1), p(sem), p(w), p(b), HW, C, D, S, float(t), p(dsem), p(partials), p(part),
                                          p(ws), stream) < 0:
                    raise RuntimeError(_lib.last_error())
                del ws
                LOSS_PATH_COUNTS["fused"] += 1
                return _FusedCodebookLoss._finish(ctx, sem_chw, bias, partials, dsem, part.sum(dim=0)[:C].contiguous(), HW, C, S)
            sim_raw = inv_gnorm = None
            if D == 256 and C <= 304 and C % 4 == 0 and _SIM_KERNEL["on"]:
                # one pass over g: split-bf16 MFMA contraction + 1/|g| (csrc/codebook_loss.hip: codebook_sim_k)
                sim_raw = torch.empty((HW, C), dtype=torch.float32, device=dev)
                inv_gnorm = torch.empty((HW,), dtype=torch.float32, device=dev)
                ws = torch.empty((int(lib.goi_codebook_sim_workspace_bytes()),), dtype=torch.uint8, device=dev)
                if lib.goi_codebook_sim(p(g), p(l1), HW, C, D, p(sim_raw), p(inv_gnorm), p(ws), stream) < 0:
                    raise RuntimeError(_lib.last_error())
                LOSS_PATH_COUNTS["three_kernel"] += 1
            else:
                LOSS_PATH_COUNTS["library_gemm"] += 1
                inv_gnorm = torch.linalg.vector_norm(g, dim=0).reciprocal_()        # [HW]
                sim_raw = torch.matmul(g.t(), l1.t())                                 # [HW, C]  (library GEMM, fp32)
            dsim = torch.empty_like(sim_raw)
            dsem = torch.empty((S, HW), dtype=torch.float32, device=dev)
            rows = lib.goi_codebook_loss_partial_rows()
            partials = torch.empty((rows, C * (S + 1) + 4), dtype=torch.float32, device=dev)
            r = lib.goi_codebook_loss_rows(p(sim_raw), p(inv_gnorm), p(sem), p(w), p(b), HW, C, S, float(t), p(dsim),
                                           p(dsem), p(partials), C_.c_void_p(torch.cuda.current_stream(dev).cuda_stream))
            if r < 0:
                raise RuntimeError(_lib.last_error())
            del sim_raw
            if D == 256 and 288 < C <= 304 and HW % 4 == 0:
                # split-K MFMA GEMM over the pixel axis with a persistent [304, 256] accumulator per CU
                blocks = lib.goi_codebook_dlut_partial_blocks()
                part = torch.empty((blocks, 304, D), dtype=torch.float32, device=dev)
                if lib.goi_codebook_dlut(p(dsim), p(g), HW, C, D, p(part),
                                         C_.c_void_p(torch.cuda.current_stream(dev).cuda_stream)) < 0:
                    raise RuntimeError(_lib.last_error())
                dl1 = part.sum(dim=0)[:C].contiguous()
            else:
                dl1 = torch.matmul(dsim.t(), g.t())                                   # [C, D]   (library GEMM)
        return _FusedCodebookLoss._finish(ctx, sem_chw, bias, partials, dsem, dl1, HW, C, S)

    @staticmethod
    def _finish(ctx, sem_chw, bias, partials, dsem, dl1, HW, C, S):
        tot = partials.sum(dim=0)                                                     # fixed order: reproducible
        dWb = tot[: C * (S + 1)].view(C, S + 1)
        sums = tot[C * (S + 1):]
        lab = sums[0] * (50.0 / (HW * C))
        sl = 1.0 - sums[1] / HW
        sl1 = sums[2] / HW
        recc = 1.0 - sums[3] / HW
        ctx.save_for_backward(dsem.view_as(sem_chw), dWb[:, :S].contiguous(), dWb[:, S].contiguous(), dl1)
        ctx.has_bias = bias is not None
        terms = torch.stack([lab, sl, sl1, recc])
        ctx.mark_non_differentiable(terms)
        return lab + sl + 0.3 * sl1 + recc, terms

    @staticmethod
    def backward(ctx, grad_loss, _grad_terms):
        dsem, dW, db, dl1 = ctx.saved_tensors
        return (grad_loss * dsem, grad_loss * dW, (grad_loss * db) if ctx.has_bias else None, grad_loss * dl1, None, None)


def fused_codebook_losses(sem_feature_chw: torch.Tensor, semantic_mlp: SemanticModel, lut: torch.Tensor,
                          gtl_chw: torch.Tensor, iteration: int):
    """Drop-in for `codebook_losses` (same arguments, same (loss, dict) result, same gradients into the
    rasterizer output, the decoder and the code book) running on the GPU as described in
    csrc/codebook_loss.hip.  Covers the reference's configuration: one Linear(S -> tab_len) decoder,
    S <= 16, tab_len <= 512."""
    if semantic_mlp.num_layer != 1:
        raise NotImplementedError("the fused losses cover the reference's configuration: one Linear(S -> tab_len)")
    lin = semantic_mlp.layers[0]
    lut1 = lut / lut.norm(dim=1, keepdim=True)  # differentiable: the kernel returns dL/dlut1
    t = 1.0 if iteration < 1000 else 2.0
    loss, terms = _FusedCodebookLoss.apply(sem_feature_chw, lin.weight, lin.bias, lut1, gtl_chw, t)
    return loss, {"lab": terms[0], "sl": terms[1], "sl1": terms[2], "recc": terms[3]}
