"""Render harness: the build's equivalent of gaussian_renderer.render (reference
gaussian_renderer/__init__.py:18-105) and of gui/gs_renderer.py:231-348's Renderer.render, working on
a minimal Gaussian container instead of scene.GaussianModel (which needs plyfile / simple_knn).

Same settings construction, same choice between SH / precomputed colours and scale+rotation /
precomputed covariance, same output dictionary keys.
"""
from __future__ import annotations

import math
from dataclasses import dataclass

import torch

from .rasterizer import GaussianRasterizationSettings, GaussianRasterizer

SH_C0 = 0.28209479177387814
SH_C1 = 0.4886025119029199
SH_C2 = (1.0925484305920792, -1.0925484305920792, 0.31539156525252005, -1.0925484305920792, 0.5462742152960396)
SH_C3 = (-0.5900435899266435, 2.890611442640554, -0.4570457994644658, 0.3731763325901154, -0.4570457994644658,
         1.445305721320277, -0.5900435899266435)


@dataclass
class PipelineParams:
    """arguments/__init__.py:56-62"""
    convert_SHs_python: bool = False
    compute_cov3D_python: bool = False
    debug: bool = False


class GaussianSet(torch.nn.Module):
    """The accessor surface of scene.GaussianModel that render() touches
    (get_xyz / get_opacity / get_scaling / get_rotation / get_features / get_semantics /
    get_covariance, scene/gaussian_model.py:90-127), holding already-activated tensors."""

    def __init__(self, means3D, scales, rotations, opacities, shs, semantics, sh_degree=3, max_sh_degree=3):
        super().__init__()
        P = torch.nn.Parameter
        self._xyz = P(means3D)
        self._scaling = P(scales)
        self._rotation = P(rotations)
        self._opacity = P(opacities)
        self._features = P(shs)
        self._semantics = P(semantics)
        self.active_sh_degree = sh_degree
        self.max_sh_degree = max_sh_degree
        self._semantics_masks = None

    @classmethod
    def from_scene(cls, scene, device):
        t = lambda a: torch.tensor(a, dtype=torch.float32, device=device)  # noqa: E731
        return cls(t(scene.means3D), t(scene.scales), t(scene.rotations), t(scene.opacities), t(scene.shs),
                   t(scene.semantics), scene.sh_degree)

    @classmethod
    def from_ply(cls, path, device, sh_degree=3, semantic_dim=None):
        """A scene saved by the reference (point_cloud.ply with sem_* columns, scene/gaussian_model.py:308-358),
        activated the way its get_* properties do.  semantic_dim=None: the width the file has (io.load_ply)."""
        from . import io as gio
        a = gio.activate(gio.load_ply(path, max_sh_degree=sh_degree, semantic_dim=semantic_dim))
        return cls(*(a[k].to(device) for k in ("means3D", "scales", "rotations", "opacities", "shs", "semantics")),
                   sh_degree=sh_degree, max_sh_degree=sh_degree)

    get_xyz = property(lambda self: self._xyz)
    get_scaling = property(lambda self: self._scaling)
    get_rotation = property(lambda self: self._rotation)
    get_opacity = property(lambda self: self._opacity)
    get_features = property(lambda self: self._features)

    @property
    def get_semantics(self):  # scene/gaussian_model.py:108-113
        return self._semantics if self._semantics_masks is None else self._semantics * self._semantics_masks

    def set_semantic_masks(self, masks=None):  # scene/gaussian_model.py:119-123
        self._semantics_masks = None if masks is None else masks.unsqueeze(1)

    def get_covariance(self, scaling_modifier=1.0):  # scene/gaussian_model.py:33-37,125-126
        return covariance_from_scaling_rotation(self._scaling, scaling_modifier, self._rotation)


def covariance_from_scaling_rotation(scaling, scaling_modifier, rotation):
    """L = R(q/|q|) diag(s*mod); Sigma = L L^T; packed xx,xy,xz,yy,yz,zz (utils/general_utils.py:70-122)."""
    q = rotation / rotation.norm(dim=1, keepdim=True)
    r, x, y, z = q[:, 0], q[:, 1], q[:, 2], q[:, 3]
    R = torch.stack([1 - 2 * (y * y + z * z), 2 * (x * y - r * z), 2 * (x * z + r * y),
                     2 * (x * y + r * z), 1 - 2 * (x * x + z * z), 2 * (y * z - r * x),
                     2 * (x * z - r * y), 2 * (y * z + r * x), 1 - 2 * (x * x + y * y)], dim=-1).reshape(-1, 3, 3)
    L = R * (scaling_modifier * scaling)[:, None, :]
    S = L @ L.transpose(1, 2)
    return torch.stack([S[:, 0, 0], S[:, 0, 1], S[:, 0, 2], S[:, 1, 1], S[:, 1, 2], S[:, 2, 2]], dim=-1)


def eval_sh(deg, sh, dirs):
    """sh [P,3,(deg+1)^2...], dirs [P,3] unit -> [P,3]; the polynomial form of utils/sh_utils.py:57-112."""
    res = SH_C0 * sh[..., 0]
    if deg > 0:
        x, y, z = dirs[..., 0:1], dirs[..., 1:2], dirs[..., 2:3]
        res = res - SH_C1 * y * sh[..., 1] + SH_C1 * z * sh[..., 2] - SH_C1 * x * sh[..., 3]
        if deg > 1:
            xx, yy, zz, xy, yz, xz = x * x, y * y, z * z, x * y, y * z, x * z
            res = (res + SH_C2[0] * xy * sh[..., 4] + SH_C2[1] * yz * sh[..., 5]
                   + SH_C2[2] * (2.0 * zz - xx - yy) * sh[..., 6] + SH_C2[3] * xz * sh[..., 7]
                   + SH_C2[4] * (xx - yy) * sh[..., 8])
            if deg > 2:
                res = (res + SH_C3[0] * y * (3 * xx - yy) * sh[..., 9] + SH_C3[1] * xy * z * sh[..., 10]
                       + SH_C3[2] * y * (4 * zz - xx - yy) * sh[..., 11]
                       + SH_C3[3] * z * (2 * zz - 3 * xx - 3 * yy) * sh[..., 12]
                       + SH_C3[4] * x * (4 * zz - xx - yy) * sh[..., 13] + SH_C3[5] * z * (xx - yy) * sh[..., 14]
                       + SH_C3[6] * x * (xx - 3 * yy) * sh[..., 15])
    return res


class TorchCamera:
    """Device-side view constants (scene/cameras.py:17-48 fields that render() reads)."""

    def __init__(self, cam, device):
        self.image_width, self.image_height = cam.image_width, cam.image_height
        self.FoVx, self.FoVy = cam.FoVx, cam.FoVy
        self.world_view_transform = torch.tensor(cam.world_view_transform, device=device)
        self.full_proj_transform = torch.tensor(cam.full_proj_transform, device=device)
        self.camera_center = torch.tensor(cam.camera_center, device=device)


class RenderResult(dict):
    """The dictionary render() returns: the reference's keys (gaussian_renderer/__init__.py:99-105).  `visibility_filter`
    (= radii > 0, :103) is formed when somebody first looks at it -- a caller that only consumes the images (a plain
    training view whose loop has no densification, an evaluation render) does not pay the extra kernel in every frame.  Every
    way of looking (indexing, get, in, iteration, keys / items / values, len, ==, repr, copy) sees the key."""

    _LAZY = "visibility_filter"

    def _fill(self):
        if not dict.__contains__(self, self._LAZY):
            dict.__setitem__(self, self._LAZY, dict.__getitem__(self, "radii") > 0)

    def __missing__(self, key):
        if key != self._LAZY:
            raise KeyError(key)
        self._fill()
        return dict.__getitem__(self, key)

    def get(self, key, default=None):
        if key == self._LAZY:
            self._fill()
        return dict.get(self, key, default)

    def __contains__(self, key):
        return key == self._LAZY or dict.__contains__(self, key)

    def __iter__(self):
        self._fill()
        return dict.__iter__(self)

    def __len__(self):
        self._fill()
        return dict.__len__(self)

    def keys(self):
        self._fill()
        return dict.keys(self)

    def items(self):
        self._fill()
        return dict.items(self)

    def values(self):
        self._fill()
        return dict.values(self)

    def copy(self):
        self._fill()
        return dict(self)

    def __eq__(self, other):
        self._fill()
        return dict.__eq__(self, other)

    __hash__ = None

    def __repr__(self):
        self._fill()
        return dict.__repr__(self)


_ZERO_ROW = {}  # (device, dtype) -> a [1, 3] zero tensor the screen-space placeholders of every frame are views of


def _screenspace_placeholder(xyz):
    """The reference's `screenspace_points = torch.zeros_like(xyz, requires_grad=True) + 0` (gaussian_renderer/__init__.py:
    27-31): a tensor of zeros whose only job is to RECEIVE the 2-D mean gradients.  Nothing reads its values (the rasterizer
    ignores means2D in the forward), so here it is a stride-0 view of one resident zero row, made a leaf: the same zeros, the
    gradient lands in `.grad` as before, and the frame pays neither the 12 MB fill nor the `+ 0` pass over it."""
    key = (xyz.device, xyz.dtype)
    z = _ZERO_ROW.get(key)
    if z is None:
        z = _ZERO_ROW[key] = torch.zeros((1, 3), dtype=xyz.dtype, device=xyz.device)
    return z.expand(xyz.shape[0], 3).detach().requires_grad_(True)


def render(viewpoint_camera, pc, pipe, bg_color, scaling_modifier=1.0, override_color=None, gaussian_mask=None):
    """gaussian_renderer.render; `gaussian_mask` adds gui/gs_renderer.py:315-321's index-select."""
    screenspace_points = _screenspace_placeholder(pc.get_xyz)
    raster_settings = GaussianRasterizationSettings(
        image_height=int(viewpoint_camera.image_height), image_width=int(viewpoint_camera.image_width),
        tanfovx=math.tan(viewpoint_camera.FoVx * 0.5), tanfovy=math.tan(viewpoint_camera.FoVy * 0.5), bg=bg_color,
        scale_modifier=scaling_modifier, viewmatrix=viewpoint_camera.world_view_transform,
        projmatrix=viewpoint_camera.full_proj_transform, sh_degree=pc.active_sh_degree,
        campos=viewpoint_camera.camera_center, prefiltered=False, debug=pipe.debug)
    rasterizer = GaussianRasterizer(raster_settings=raster_settings)

    means3D, means2D, opacity = pc.get_xyz, screenspace_points, pc.get_opacity
    scales = rotations = cov3D_precomp = None
    if pipe.compute_cov3D_python:
        cov3D_precomp = pc.get_covariance(scaling_modifier)
    else:
        scales, rotations = pc.get_scaling, pc.get_rotation
    shs = colors_precomp = None
    if override_color is None:
        if pipe.convert_SHs_python:
            shs_view = pc.get_features.transpose(1, 2).view(-1, 3, (pc.max_sh_degree + 1) ** 2)
            dir_pp = pc.get_xyz - viewpoint_camera.camera_center.repeat(pc.get_features.shape[0], 1)
            dir_pp_normalized = dir_pp / dir_pp.norm(dim=1, keepdim=True)
            colors_precomp = torch.clamp_min(eval_sh(pc.active_sh_degree, shs_view, dir_pp_normalized) + 0.5, 0.0)
        else:
            shs = pc.get_features
    else:
        colors_precomp = override_color
    semantics = pc.get_semantics

    if gaussian_mask is not None:
        sel = lambda t: None if t is None else t[gaussian_mask]  # noqa: E731
        semantics, means3D, scales, rotations, opacity, shs = map(sel, (semantics, means3D, scales, rotations, opacity, shs))
        cov3D_precomp, colors_precomp, means2D = sel(cov3D_precomp), sel(colors_precomp), sel(means2D)

    rendered_image, rendered_sem, radii, depth, alpha = rasterizer(
        means3D=means3D, means2D=means2D, shs=shs, colors_precomp=colors_precomp, semantics=semantics,
        opacities=opacity, scales=scales, rotations=rotations, cov3D_precomp=cov3D_precomp)
    return RenderResult({"render": rendered_image, "semantics": rendered_sem, "depth": depth, "alpha": alpha,
                         "viewspace_points": screenspace_points, "radii": radii})


_VIEW_STREAMS = {}  # device index -> list of side streams (created on first use, reused)


def render_views(cameras, pc, pipe, bg_color, loss_fn=None, streams=2, **render_kwargs):
    """A batch of INDEPENDENT views of one model on `streams` HIP streams of its device: view i runs on stream
    i % streams -- forward and, when `loss_fn(i, out) -> scalar` is given, `loss.backward()` right behind it, so the gradients of
    the batch accumulate in the model's leaves exactly as a loop over the views would leave them.  One view's small, latency-bound
    kernels and kernel tails overlap with another view's work: +12 % (two streams) to +18 % (four) on the headline scene, +13 %
    at 3 M Gaussians, +11 % at 6 M / 512x512 (bench.py reports `value_two_views_in_flight`).  Every stream in flight owns its
    workspaces -- above all the backward's row scratch, 4 x capacity x 129 bytes (31 GB per stream at 3 M Gaussians) -- and the
    gain only shows once the streams' allocator pools have stopped growing (a couple of dozen views).
    Returns the list of render() dictionaries (every tensor safe to use on the caller's current stream).
    Not for the reference's own loop (one view, optimizer step, next view: train.py:112-199) -- there the views are not
    independent."""
    dev = pc.get_xyz.device
    cur = torch.cuda.current_stream(dev)
    pool = _VIEW_STREAMS.setdefault(dev.index, [])
    while len(pool) < streams:
        pool.append(torch.cuda.Stream(dev))
    side = pool[:streams]
    for s in side:
        s.wait_stream(cur)  # parameters, cameras and upstream state written on the caller's stream are visible
    if loss_fn is not None:
        # the leaves were created on the caller's stream and their gradients arrive from the side streams: intended here
        quiet = getattr(torch.autograd.graph, "set_warn_on_accumulate_grad_stream_mismatch", None)
        if quiet is not None:
            quiet(False)
    outs = []
    for i, cam in enumerate(cameras):
        s = side[i % streams]
        with torch.cuda.stream(s):
            out = render(cam, pc, pipe, bg_color, **render_kwargs)
            if loss_fn is not None:
                loss_fn(i, out).backward()
        for t in out.values():
            if torch.is_tensor(t) and t.is_cuda:
                t.record_stream(cur)  # allocated on a side stream, handed to the caller's stream
        outs.append(out)
    for s in side:
        cur.wait_stream(s)
    return outs


def render_gui(viewpoint_camera, pc, bg_color, scaling_modifier=1.0, override_color=None, compute_cov3D_python=False,
               convert_SHs_python=False, gaussian_mask=None):
    """gui/gs_renderer.py:231-348 (Renderer.render): same rasterizer call as render(), optional
    `gaussian_mask` index-select of every per-Gaussian tensor (:315-321), image clamped to [0,1] (:336),
    result keyed "image" instead of "render"."""
    pipe = PipelineParams(convert_SHs_python=convert_SHs_python, compute_cov3D_python=compute_cov3D_python)
    out = render(viewpoint_camera, pc, pipe, bg_color, scaling_modifier, override_color, gaussian_mask)
    return {"image": out["render"].clamp(0, 1), "semantics": out["semantics"], "depth": out["depth"],
            "alpha": out["alpha"], "viewspace_points": out["viewspace_points"],
            "visibility_filter": out["visibility_filter"], "radii": out["radii"]}
