"""Python operator API of the rasterizer: the names, parameter order, defaults, return tuples and
error messages of the reference's `diff_gaussian_rasterization` package
(submodules/diff-gaussian-rasterization/diff_gaussian_rasterization/__init__.py), so that
gaussian_renderer/__init__.py:14,36-95 and gui/gs_renderer.py:10-13,263-334 run unchanged.

    GaussianRasterizationSettings   NamedTuple of per-view constants           (ref :246-258)
    GaussianRasterizer              nn.Module: forward / trace / markVisible   (ref :260-349)
    rasterize_gaussians, trace_gaussians                                        (ref :21-69)
    _RasterizeGaussians             torch.autograd.Function                     (ref :71-244)

Differences, all behind the same surface:
  * compute goes through goi_hyperplane_amd._C (C ABI of libgoi_raster.so, hand-written HIP);
  * the backward is gated on ctx.needs_input_grad: with only the semantic features trainable (the reference's
    default training configuration) the feature-gradient-only kernel runs, with the SH coefficients frozen dL/dSH
    is not formed (see set_backward_mode);
  * the number of semantic channels S is read from `semantics.shape[1]` at run time (the reference
    compiles SEM_CHANNELS = 10 in, cuda_rasterizer/config.h:18).
"""
from __future__ import annotations

import os

from typing import NamedTuple

import torch
import torch.nn as nn

from . import _C


def _cpu_snapshot(args):
    return tuple(a.detach().cpu().clone() if isinstance(a, torch.Tensor) else a for a in args)


class GaussianRasterizationSettings(NamedTuple):
    image_height: int
    image_width: int
    tanfovx: float
    tanfovy: float
    bg: torch.Tensor
    scale_modifier: float
    viewmatrix: torch.Tensor
    projmatrix: torch.Tensor
    sh_degree: int
    campos: torch.Tensor
    prefiltered: bool
    debug: bool


def _forward_args(rs, means3D, colors_precomp, payload, opacities, scales, rotations, cov3Ds_precomp, sh):
    """Argument order of _C.rasterize_gaussians / _C.rasterize_gaussians_trace (ext.cpp:15-20)."""
    return (rs.bg, means3D, colors_precomp, payload, opacities, scales, rotations, rs.scale_modifier, cov3Ds_precomp,
            rs.viewmatrix, rs.projmatrix, rs.tanfovx, rs.tanfovy, rs.image_height, rs.image_width, sh, rs.sh_degree,
            rs.campos, rs.prefiltered, rs.debug)


def _call_with_snapshot(fn, args, debug, dump_name, what):
    """debug=True: keep a CPU copy of the arguments and dump it if the call raises (ref :112-119,165-172)."""
    if not debug:
        return fn(*args)
    snapshot = _cpu_snapshot(args)
    try:
        return fn(*args)
    except Exception:
        torch.save(snapshot, dump_name)
        print(f"\nAn error occured in {what}. Please forward {dump_name} for debugging.")
        raise


# Gradient gating ("skip unneeded gradients").  The reference's kernels always compute every
# gradient; its DEFAULT training configuration optimises only the semantic features (arguments/__init__.py:85-90,
# scene/gaussian_model.py:185-246 freeze the rest).  ctx.needs_input_grad says which inputs want a gradient:
#
#   * only `semantics` (and the screen-space placeholder means2D): the feature-gradient-only backward
#     (goi_raster_backward_semantics) runs -- dL/dsemantics is BIT-IDENTICAL to the full backward's, 2.3x faster.  The
#     one observable difference: viewspace_points.grad is zero instead of the screen-space gradient (nothing in a
#     semantics-only run consumes it; train.py has no densification).  This is the default ("auto");
#     GOI_BACKWARD=full or set_backward_mode(semantics_only=False) always runs the full kernel,
#     GOI_BACKWARD=semantics / semantics_only=True is the same as "auto" (kept for round-1 callers).
#   * `sh` frozen while other geometry trains: the 192-byte-per-Gaussian dL/dSH row is not formed.
#
# sh_factored (data-parallel training, dist.allreduce_gradients_sh_factored): the backward does not form dL/dSH
# (192 of the 300 gradient bytes per Gaussian); the SH tensor gets NO gradient from autograd, and the factor of
#     dL/dSH[g][k] = basis_k(direction camera -> g) * gcol[g]
# -- the clamp-masked colour gradient gcol [P,3] -- is left for take_sh_factor().  The ranks then all-gather 12 bytes
# per Gaussian and view instead of all-reducing 192, and every rank rebuilds the summed dL/dSH locally.
def _env_backward_mode():
    v = os.environ.get("GOI_BACKWARD", "auto").strip().lower()
    if v in ("full", "0", "off"):
        return False
    if v in ("auto", "", "semantics", "1", "on"):
        return "auto"
    raise ValueError(f"GOI_BACKWARD={v!r}: expected auto, semantics or full")


_BACKWARD_MODE = {"semantics_only": _env_backward_mode(), "sh_factored": False}
_SH_FACTOR = {"last": None}
_LAST_BACKWARD = {"kernel": None}  # "semantics" | "full" | "full_no_dsh" | "factored": which path the last backward took


def set_backward_mode(semantics_only=None, sh_factored: bool = None) -> None:
    """semantics_only: "auto" / True -- take the feature-gradient-only backward whenever nothing but the semantic
    features needs a gradient (default); False -- always the full kernel; None -- leave unchanged."""
    if semantics_only is not None:
        _BACKWARD_MODE["semantics_only"] = "auto" if semantics_only in ("auto", True) else False
    if sh_factored is not None:
        _BACKWARD_MODE["sh_factored"] = bool(sh_factored)
        _SH_FACTOR["last"] = None


def set_forward_mode(speculative=None, headroom=None, capacity="keep", on_overflow=None, max_ahead=None,
                     inference_speculative=None, min_history=None, depth_cut=None) -> None:
    """Forward without the host round trip (default for frames a backward may follow) or the reference's synchronous
    forward (default for frames rendered without autograd: their image is the product): see _C.set_forward_mode and
    include/goi_raster.h (goi_raster_forward_async).  GOI_FORWARD=exact|speculative, GOI_FORWARD_INFERENCE=exact|speculative,
    GOI_BINNING_HEADROOM, GOI_OVERFLOW=warn|raise set the process defaults.  depth_cut (GOI_DEPTH_CUT=0|1, default OFF: opt-in):
    speculative training frames of a camera that was rendered before list, per tile, only Gaussians up to the depth the
    camera's previous frame found worth listing (_C._depth_cut_for): bit-identical frames while the learnt cut holds, a redone
    or skipped view when it does not."""
    _C.set_forward_mode(speculative, headroom, capacity, on_overflow, max_ahead, inference_speculative, min_history, depth_cut)


def truncated_flag(accumulated: bool = False):
    """int32[1] device tensor (or None): non-zero iff the most recent forward of this thread was a speculative frame
    whose instance list did not fit its capacity.  Such a frame back-propagates ZERO gradients on the device;
    `FusedAdam.step(skip_if=truncated_flag())` skips its optimiser step on the device as well -- the skip is opt-in: a plain
    `step()` still applies momentum (see _C.truncated_flag).  accumulated=True: the OR over all frames of the open
    truncation window (accumulate_truncation / reset_truncation), for steps that accumulate several views."""
    return _C.truncated_flag(accumulated)


def accumulate_truncation(device=None) -> None:
    """Opens (device) / closes (None) this thread's truncation window: see _C.accumulate_truncation."""
    _C.accumulate_truncation(device)


def reset_truncation() -> None:
    _C.reset_truncation()


def speculation_stats() -> dict:
    """Counters of the speculative forward since import: exact_frames, speculative_frames, overflows, redone, waits."""
    return dict(_C.SPECULATION_STATS)


_LAST_FORWARD = {"num_rendered": None}


def last_num_rendered():
    """`num_rendered` of the most recent forward of this process: an int (exact frame) or a _C.LazyCount (speculative
    frame).  int(last_num_rendered()) right after a render is the "read before use" hook of the speculative forward:
    it waits for the frame's count and, had the frame overflowed its capacity, redoes it in place before anything
    consumes the outputs."""
    return _LAST_FORWARD["num_rendered"]


def last_backward_kernel():
    """Which backward path the most recent _RasterizeGaussians.backward took (tests, bench reporting)."""
    return _LAST_BACKWARD["kernel"]


def take_sh_factor():
    """-> dict(gcol [P,3], campos [3], degree, M) of the most recent backward run in sh_factored mode (None if
    there was none since the last call)."""
    f, _SH_FACTOR["last"] = _SH_FACTOR["last"], None
    return f


class _RasterizeGaussians(torch.autograd.Function):
    @staticmethod
    def forward(ctx, means3D, means2D, sh, colors_precomp, semantics, opacities, scales, rotations, cov3Ds_precomp,
                raster_settings):
        args = _forward_args(raster_settings, means3D, colors_precomp, semantics, opacities, scales, rotations,
                             cov3Ds_precomp, sh)
        (num_rendered, color, semant, depth, alpha, radii, geomBuffer, binningBuffer, imgBuffer) = _call_with_snapshot(
            _C.rasterize_gaussians, args, raster_settings.debug, "snapshot_fw.dump", "forward")
        ctx.raster_settings = raster_settings
        ctx.num_rendered = num_rendered  # an int, or the LazyCount of a speculative frame (never forced here)
        _LAST_FORWARD["num_rendered"] = num_rendered
        ctx.set_materialize_grads(False)  # unused outputs (depth, alpha, ...) reach backward as None, not as zero tensors
        ctx.save_for_backward(colors_precomp, semantics, means3D, scales, rotations, cov3Ds_precomp, radii, sh,
                              geomBuffer, binningBuffer, imgBuffer, alpha)
        ctx.mark_non_differentiable(radii)
        return color, semant, radii, depth, alpha

    @staticmethod
    def backward(ctx, grad_out_color, grad_out_sem, grad_out_radii, grad_depth, grad_alpha):
        rs = ctx.raster_settings
        (colors_precomp, semantics, means3D, scales, rotations, cov3Ds_precomp, radii, sh, geomBuffer, binningBuffer,
         imgBuffer, alpha) = ctx.saved_tensors
        need = ctx.needs_input_grad  # means3D, means2D, sh, colors, semantics, opacities, scales, rotations, cov3D
        if (_BACKWARD_MODE["semantics_only"] and need[4] and not rs.debug
                and not (need[0] or need[2] or need[3] or need[5] or need[6] or need[7] or need[8])):
            _LAST_BACKWARD["kernel"] = "semantics"
            # only the semantic features are trainable: feature-gradient-only kernel; the screen-space
            # placeholder (means2D) gets zeros -- nothing in a semantics-only run consumes it
            if grad_out_sem is None:  # the loss does not touch the semantic map
                return (None, None, None, None, torch.zeros_like(semantics), None, None, None, None, None)
            g_sem = _C.rasterize_gaussians_backward_semantics(
                rs.bg, means3D, radii, semantics, rs.viewmatrix, rs.projmatrix, rs.tanfovx, rs.tanfovy, grad_out_sem,
                rs.campos, geomBuffer, ctx.num_rendered, binningBuffer, imgBuffer, alpha, rs.sh_degree, rs.debug)
            g_2d = torch.zeros((means3D.shape[0], 3), dtype=means3D.dtype, device=means3D.device) if need[1] else None
            return (None, g_2d, None, None, g_sem, None, None, None, None, None)
        args = (rs.bg, means3D, radii, colors_precomp, semantics, scales, rotations, rs.scale_modifier, cov3Ds_precomp,
                rs.viewmatrix, rs.projmatrix, rs.tanfovx, rs.tanfovy, grad_out_color, grad_out_sem, grad_depth,
                grad_alpha, sh, rs.sh_degree, rs.campos, geomBuffer, ctx.num_rendered, binningBuffer, imgBuffer, alpha,
                rs.debug)
        has_sh = sh is not None and sh.numel() > 0
        factored = bool(_BACKWARD_MODE["sh_factored"] and has_sh and need[2] and not rs.debug)
        if factored:
            (grad_means2D, gcol, grad_semantics, grad_opacities, grad_means3D, grad_cov3Ds_precomp, grad_sh,
             grad_scales, grad_rotations) = _C.rasterize_gaussians_backward_sh_factored(*args)
            grad_colors_precomp = None
            _SH_FACTOR["last"] = dict(gcol=gcol, campos=rs.campos, degree=int(rs.sh_degree), M=int(sh.size(1)))
            _LAST_BACKWARD["kernel"] = "factored"
        elif has_sh and not need[2] and not rs.debug and _BACKWARD_MODE["semantics_only"]:
            # the SH coefficients are frozen: same kernels without the dL/dSH row (the colour gradient that mode leaves
            # behind belongs to no input here -- colours come from the SH -- and is dropped)
            (grad_means2D, _gcol, grad_semantics, grad_opacities, grad_means3D, grad_cov3Ds_precomp, grad_sh,
             grad_scales, grad_rotations) = _C.rasterize_gaussians_backward_sh_factored(*args)
            grad_colors_precomp = None
            _LAST_BACKWARD["kernel"] = "full_no_dsh"
        elif _ACCUM["state"] is not None and not rs.debug and _C._ext() is not None:
            # a multi-view batch (accumulate_gradients / dist.backward_views): this view's gradients are ADDED, on the device, to
            # the tensors the batch's first view returned; autograd gets nothing from here -- the batch back-propagates the sums
            # through the activations once, at its end
            st = _ACCUM["state"]
            first = st.get("grads")
            res = _C.rasterize_gaussians_backward_accumulate(None if first is None else first[4], *args)
            if first is None:
                st["grads"] = res
            st["views"] = st.get("views", 0) + 1
            _LAST_BACKWARD["kernel"] = "full_accumulate"
            return (None,) * 10
        else:
            _LAST_BACKWARD["kernel"] = "full"
            (grad_means2D, grad_colors_precomp, grad_semantics, grad_opacities, grad_means3D, grad_cov3Ds_precomp,
             grad_sh, grad_scales, grad_rotations) = _call_with_snapshot(_C.rasterize_gaussians_backward, args,
                                                                         rs.debug, "snapshot_bw.dump", "backward")
        # input order: means3D, means2D, sh, colors_precomp, semantics, opacities, scales, rotations, cov3Ds_precomp
        return (grad_means3D, grad_means2D, grad_sh, grad_colors_precomp, grad_semantics, grad_opacities, grad_scales,
                grad_rotations, grad_cov3Ds_precomp, None)

    @staticmethod
    def trace(means3D, means2D, sh, colors_precomp, img_sem, opacities, scales, rotations, cov3Ds_precomp,
              raster_settings):
        args = _forward_args(raster_settings, means3D, colors_precomp, img_sem, opacities, scales, rotations,
                             cov3Ds_precomp, sh)
        (num_rendered, color, gau_sem, num_gsem, geomBuffer, binningBuffer, imgBuffer) = _call_with_snapshot(
            _C.rasterize_gaussians_trace, args, raster_settings.debug, "snapshot_fw.dump", "forward")
        return color, gau_sem, num_gsem


def rasterize_gaussians(means3D, means2D, sh, colors_precomp, semantics, opacities, scales, rotations, cov3Ds_precomp,
                        raster_settings):
    # can a backward follow this frame?  (inside Function.forward grad mode is always off, so it is decided here.)  A frame
    # rendered only for its image takes the exact forward by default, see _C._CALL
    # (means2D is the caller's gradient SINK -- the reference's harness makes it require a gradient on every call -- so it
    # does not count)
    tensors = (means3D, sh, colors_precomp, semantics, opacities, scales, rotations, cov3Ds_precomp)
    if _ACCUM["state"] is not None:  # (the operands of the batch's views are the same activations of the same parameters)
        _ACCUM["state"]["inputs"] = tensors
    _C._CALL.inference = not (torch.is_grad_enabled()
                              and any(isinstance(t, torch.Tensor) and t.requires_grad for t in tensors))
    # eligible for the opt-in geometry cache: no input but the semantic features (and the sink) can receive a gradient
    _C._CALL.geometry_frozen = not any(isinstance(t, torch.Tensor) and t.requires_grad
                                       for t in (means3D, sh, colors_precomp, opacities, scales, rotations, cov3Ds_precomp))
    try:
        return _RasterizeGaussians.apply(means3D, means2D, sh, colors_precomp, semantics, opacities, scales, rotations,
                                         cov3Ds_precomp, raster_settings)
    finally:
        _C._CALL.inference = False
        _C._CALL.geometry_frozen = False


_ACCUM = {"state": None}  # process-wide on purpose: Function.backward runs on autograd's device thread, not the caller's


class accumulate_gradients:
    """Context of a MULTI-VIEW BATCH (dist.backward_views): while it is active, the full backward of every rasterizer call adds
    its per-Gaussian gradients ON THE DEVICE to the tensors the first such call returned (goi_raster_backward3 with
    GOI_BACKWARD_ACCUMULATE: a view reads and rewrites the rows of the Gaussians it sees, nothing else) and hands autograd no
    gradient; `state` collects {"grads": the nine tensors in the op's order (dL_dmeans2D, dL_dcolors, dL_dsemantics, dL_dopacity,
    dL_dmeans3D, dL_dcov3D, dL_dsh, dL_dscales, dL_drotations), "inputs": the operands of the last call (means3D, sh,
    colors_precomp, semantics, opacities, scales, rotations, cov3Ds_precomp), "views": how many views are summed}.  The caller
    back-propagates the sums through whatever produced the operands (the activations), once.  Valid only while the parameters do
    not change between the views of the batch.  One batch at a time per process; the semantics-only, factored and debug
    backwards, and the ctypes binding, do not accumulate (they return gradients as usual: state["views"] stays put)."""

    def __init__(self, state: dict):
        self.state = state

    def __enter__(self):
        if _ACCUM["state"] is not None and _ACCUM["state"] is not self.state:
            raise RuntimeError("accumulate_gradients: another batch is being accumulated")
        _ACCUM["state"] = self.state
        return self.state

    def __exit__(self, *exc):
        _ACCUM["state"] = None
        return False


def set_geometry_cache(max_bytes) -> None:
    """Opt-in cache of per-camera geometry and tile lists for runs in which ONLY the semantic features are trained (the
    reference's semantic stage): a camera seen before renders with the blend alone.  max_bytes of device memory (~270 MB per
    camera at 1 M Gaussians, 1600x1056); 0 / None disables and empties it.  The caller vouches that positions, covariances,
    opacities and colours do not change while the cache is on; env GOI_GEOMETRY_CACHE_GB sets it at import."""
    _C.set_geometry_cache(max_bytes)


def geometry_cache_stats() -> dict:
    return _C.geometry_cache_stats()


def trace_gaussians(means3D, means2D, sh, colors_precomp, img_sem, opacities, scales, rotations, cov3Ds_precomp,
                    raster_settings):
    return _RasterizeGaussians.trace(means3D, means2D, sh, colors_precomp, img_sem, opacities, scales, rotations,
                                     cov3Ds_precomp, raster_settings)


def _check_exclusive(shs, colors_precomp, scales, rotations, cov3D_precomp):
    if (shs is None and colors_precomp is None) or (shs is not None and colors_precomp is not None):
        raise Exception('Please provide excatly one of either SHs or precomputed colors!')
    if ((scales is None or rotations is None) and cov3D_precomp is None) or (
            (scales is not None or rotations is not None) and cov3D_precomp is not None):
        raise Exception('Please provide exactly one of either scale/rotation pair or precomputed 3D covariance!')


def _absent_to_empty(*tensors):
    """None -> an empty CPU tensor, the reference's marker for an absent optional input (ref :286-297)."""
    return tuple(torch.Tensor([]) if t is None else t for t in tensors)


class GaussianRasterizer(nn.Module):
    def __init__(self, raster_settings):
        super().__init__()
        self.raster_settings = raster_settings

    def markVisible(self, positions):
        with torch.no_grad():
            rs = self.raster_settings
            return _C.mark_visible(positions, rs.viewmatrix, rs.projmatrix)

    def forward(self, means3D, means2D, opacities, shs=None, colors_precomp=None, semantics=None, scales=None,
                rotations=None, cov3D_precomp=None):
        _check_exclusive(shs, colors_precomp, scales, rotations, cov3D_precomp)
        shs, colors_precomp, semantics, scales, rotations, cov3D_precomp = _absent_to_empty(
            shs, colors_precomp, semantics, scales, rotations, cov3D_precomp)
        return rasterize_gaussians(means3D, means2D, shs, colors_precomp, semantics, opacities, scales, rotations,
                                   cov3D_precomp, self.raster_settings)

    def trace(self, means3D, means2D, opacities, shs=None, colors_precomp=None, img_sem=None, scales=None,
              rotations=None, cov3D_precomp=None):
        _check_exclusive(shs, colors_precomp, scales, rotations, cov3D_precomp)
        shs, colors_precomp, img_sem, scales, rotations, cov3D_precomp = _absent_to_empty(
            shs, colors_precomp, img_sem, scales, rotations, cov3D_precomp)
        return trace_gaussians(means3D, means2D, shs, colors_precomp, img_sem, opacities, scales, rotations,
                               cov3D_precomp, self.raster_settings)
