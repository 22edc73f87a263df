"""`_C`: the four functions the reference's pybind module exports
(submodules/diff-gaussian-rasterization/ext.cpp:15-20), re-implemented over the C ABI of
libgoi_raster.so.  Positional signatures, return tuples, the "empty tensor = absent" convention and
the error behaviour mirror rasterize_points.cu:35-123 (forward), :125-211 (trace), :213-306
(backward), :308-327 (mark_visible).  Tensors stay torch tensors on this side; only raw device
pointers, sizes and the current HIP stream cross into the library.
"""
from __future__ import annotations

import ctypes as C

import torch

from . import _lib
from ._lib import ALLOC_FN, GoiRasterScene


def _ptr(t):
    """Device pointer of a tensor, or None for the reference's 'empty tensor means absent'."""
    if t is None or t.numel() == 0:
        return None
    return C.c_void_p(t.data_ptr())


def _prep(t, name, dev, dtype=torch.float32):
    """contiguous + checked view of an input (the reference calls .contiguous() and assumes fp32)."""
    if t is None or t.numel() == 0:
        return None
    if t.dtype != dtype:
        raise TypeError(f"{name} must be {dtype}, got {t.dtype}")
    if t.device != dev:
        raise ValueError(f"{name} is on {t.device}, expected {dev}")
    return t.contiguous()


def _stream(dev):
    return C.c_void_p(torch.cuda.current_stream(dev).cuda_stream)


def _check_device(means3D):
    if means3D.ndimension() != 2 or means3D.size(1) != 3:
        raise RuntimeError("means3D must have dimensions (num_points, 3)")  # rasterize_points.cu:58-60
    if not means3D.is_cuda:
        raise RuntimeError(
            "goi_hyperplane_amd: tensors must live on a ROCm GPU (cuda device); there is no CPU fallback in this "
            "package")
    return means3D.device


class _BinningAllocator:
    """The allocation callback handed to the library (reference: resizeFunctional,
    rasterize_points.cu:27-33): allocates the binning workspace as a uint8 torch tensor."""

    def __init__(self, dev):
        self.dev = dev
        self.tensor = torch.empty(0, dtype=torch.uint8, device=dev)
        self.error = None
        self.cb = ALLOC_FN(self._alloc)

    def _alloc(self, _user, nbytes):
        try:
            # num_rendered changes with every view; rounding the request up to 16 MiB steps lets the caching allocator
            # hand back the same block instead of growing a new one (a fresh hipMalloc inside a training step)
            step = 16 << 20
            self.tensor = torch.empty((int(nbytes) + step - 1) // step * step, dtype=torch.uint8, device=self.dev)
            return self.tensor.data_ptr()
        except Exception as ex:  # never let an exception cross the C boundary
            self.error = ex
            return None


_SCRATCH = {}


def _backward_scratch(nbytes: int, dev) -> torch.Tensor:
    """Grow-only backward scratch per (device, stream): the buffer is 4*R*129 bytes (4 GB at the headline
    size) and R changes with every view, so going through the caching allocator each step leaves it
    hunting for a block of a new size -- an occasional 4 GB hipMalloc inside a training step.  The
    scratch is dead when goi_raster_backward returns (same stream), so one buffer serves all calls."""
    key = (dev.index, torch.cuda.current_stream(dev).cuda_stream)
    buf = _SCRATCH.get(key)
    if buf is None or buf.numel() < nbytes:
        _SCRATCH.pop(key, None)
        buf = None
        buf = torch.empty(int(nbytes * 1.25) + 256, dtype=torch.uint8, device=dev)
        _SCRATCH[key] = buf
    return buf


def release_scratch(device=None) -> int:
    """Frees the grow-only backward scratch buffers (all devices, or one): they are dead between calls, hold
    ~1.25 x the largest view's 4*R*129 bytes per (device, stream) and are invisible to torch.cuda.empty_cache()
    while referenced here.  Returns the number of bytes released to the caching allocator."""
    freed = 0
    for key in [k for k in _SCRATCH if device is None or k[0] == torch.device(device).index]:
        freed += _SCRATCH.pop(key).numel()
    return freed


def _scene(P, S, H, W, bg, means3D, sh, colors, semantics, opacity, scales, rotations, scale_modifier, cov3D,
           viewmatrix, projmatrix, tan_fovx, tan_fovy, degree, campos, prefiltered, debug):
    M = 0 if (sh is None or sh.numel() == 0) else int(sh.size(1))
    return GoiRasterScene(
        int(P), int(degree), M, int(S), int(W), int(H), _ptr(bg), _ptr(means3D), _ptr(sh), _ptr(colors),
        _ptr(semantics), _ptr(opacity), _ptr(scales), float(scale_modifier), _ptr(rotations), _ptr(cov3D),
        _ptr(viewmatrix), _ptr(projmatrix), _ptr(campos), float(tan_fovx), float(tan_fovy), int(bool(prefiltered)),
        int(bool(debug)))


def rasterize_gaussians(background, means3D, colors, semantics, opacity, scales, rotations, scale_modifier,
                        cov3D_precomp, viewmatrix, projmatrix, tan_fovx, tan_fovy, image_height, image_width, sh,
                        degree, campos, prefiltered, debug):
    """-> (num_rendered, color[3,H,W], semantic[S,H,W], depth[1,H,W], alpha[1,H,W], radii[P] i32,
    geomBuffer u8, binningBuffer u8, imgBuffer u8)"""
    lib = _lib.load()
    dev = _check_device(means3D)
    P, H, W = int(means3D.size(0)), int(image_height), int(image_width)
    if semantics is None or semantics.numel() == 0:
        if P > 0:
            raise RuntimeError("semantics [P,S] is required (the reference dereferences it unconditionally, "
                               "cuda_rasterizer/forward.cu:363)")
        S = 10
    else:
        if semantics.ndimension() != 2 or semantics.size(0) != P:
            raise RuntimeError("semantics must have dimensions (num_points, S)")
        S = int(semantics.size(1))
    if not (1 <= S <= 32):
        raise RuntimeError(f"unsupported number of semantic channels S={S} (1..32)")
    f32 = dict(dtype=torch.float32, device=dev)
    with torch.cuda.device(dev):
        out_color = torch.empty((3, H, W), **f32)
        out_sem = torch.empty((S, H, W), **f32)
        out_depth = torch.empty((1, H, W), **f32)
        out_alpha = torch.empty((1, H, W), **f32)
        radii = torch.empty((P,), dtype=torch.int32, device=dev)
        ten = dict(bg=_prep(background, "background", dev), means3D=_prep(means3D, "means3D", dev),
                   sh=_prep(sh, "sh", dev), colors=_prep(colors, "colors_precomp", dev),
                   semantics=_prep(semantics, "semantics", dev), opacity=_prep(opacity, "opacities", dev),
                   scales=_prep(scales, "scales", dev), rotations=_prep(rotations, "rotations", dev),
                   cov3D=_prep(cov3D_precomp, "cov3D_precomp", dev), viewmatrix=_prep(viewmatrix, "viewmatrix", dev),
                   projmatrix=_prep(projmatrix, "projmatrix", dev), campos=_prep(campos, "campos", dev))
        geom = torch.empty(lib.goi_raster_geom_bytes(P) if P > 0 else 0, dtype=torch.uint8, device=dev)
        img = torch.empty(lib.goi_raster_image_bytes(W, H) if P > 0 else 0, dtype=torch.uint8, device=dev)
        alloc = _BinningAllocator(dev)
        sc = _scene(P, S, H, W, ten["bg"], ten["means3D"], ten["sh"], ten["colors"], ten["semantics"], ten["opacity"],
                    ten["scales"], ten["rotations"], scale_modifier, ten["cov3D"], ten["viewmatrix"],
                    ten["projmatrix"], tan_fovx, tan_fovy, degree, ten["campos"], prefiltered, debug)
        n = lib.goi_raster_forward(C.byref(sc), _ptr(geom), _ptr(img), alloc.cb, None, _ptr(out_color), _ptr(out_sem),
                                   _ptr(out_depth), _ptr(out_alpha), _ptr(radii), _stream(dev))
        if alloc.error is not None:
            raise alloc.error
        if n < 0:
            raise RuntimeError(_lib.last_error())
    return n, out_color, out_sem, out_depth, out_alpha, radii, geom, alloc.tensor, img


def _backward_impl(background, means3D, radii, colors, semantics, scales, rotations, scale_modifier,
                                 cov3D_precomp, viewmatrix, projmatrix, tan_fovx, tan_fovy, dL_dout_color,
                                 dL_dout_semantic, dL_dout_depth, dL_dout_alpha, sh, degree, campos, geomBuffer, R,
                                 binningBuffer, imageBuffer, alphas, debug, sh_factored=False):
    """-> (dL_dmeans2D[P,3], dL_dcolors[P,3], dL_dsemantics[P,S], dL_dopacity[P,1], dL_dmeans3D[P,3],
    dL_dcov3D[P,6], dL_dsh[P,M,3], dL_dscales[P,3], dL_drotations[P,4])

    sh_factored (not in the reference's pybind module; FACTORED mode of goi_raster_backward): dL_dsh is not formed
    (returned as None) and dL_dcolors is the clamp-masked colour gradient g, the factor of
    dL/dSH[k] = basis_k(view direction) * g -- see sh_grad_from_views and dist.allreduce_gradients_sh_factored."""
    lib = _lib.load()
    dev = _check_device(means3D)
    P = int(means3D.size(0))
    # the reference reads H, W off dL_dout_color (rasterize_points.cu:243-244); here any upstream gradient may be
    # None (an output the loss does not use), so the sizes come from tensors that always exist
    H, W = int(alphas.size(-2)), int(alphas.size(-1))
    S = int(semantics.size(1))
    M = 0 if (sh is None or sh.numel() == 0) else int(sh.size(1))
    sh_factored = bool(sh_factored) and M > 0
    f32 = dict(dtype=torch.float32, device=dev)
    with torch.cuda.device(dev):
        # every element is written by the library (atomically accumulated ones are zeroed there).
        # The gradients of the six Gaussian parameter tensors are views of ONE flat buffer (256-byte
        # aligned sections): autograd hands these views to the leaves as .grad, so the data-parallel
        # exchange can be a single all-reduce over the buffer instead of one per tensor (dist.py).
        sections = (("means3D", (P, 3)), ("sh", (P, 0 if sh_factored else M, 3)), ("semantics", (P, S)),
                    ("opacity", (P, 1)), ("scales", (P, 3)), ("rotations", (P, 4)))
        offs, total = {}, 0
        for name, shape in sections:
            offs[name] = total
            n = 1
            for d in shape:
                n *= d
            total += (n + 63) // 64 * 64
        flat = torch.empty((total,), **f32)

        def view(name, shape):
            n = 1
            for d in shape:
                n *= d
            return flat[offs[name]:offs[name] + n].view(shape)
        dL_dmeans3D = view("means3D", (P, 3))
        dL_dsh = None if sh_factored else view("sh", (P, M, 3))
        dL_dsemantics = view("semantics", (P, S))
        dL_dopacity = view("opacity", (P, 1))
        dL_dscales = view("scales", (P, 3))
        dL_drotations = view("rotations", (P, 4))
        del flat
        dL_dmeans2D = torch.empty((P, 3), **f32)
        dL_dcolors = torch.empty((P, 3), **f32)
        dL_ddepths = torch.empty((P, 1), **f32)
        dL_dconic = torch.empty((P, 2, 2), **f32)
        dL_dcov3D = torch.empty((P, 6), **f32)
        if P != 0:
            ten = dict(bg=_prep(background, "background", dev), means3D=_prep(means3D, "means3D", dev),
                       sh=_prep(sh, "sh", dev), colors=_prep(colors, "colors_precomp", dev),
                       semantics=_prep(semantics, "semantics", dev), scales=_prep(scales, "scales", dev),
                       rotations=_prep(rotations, "rotations", dev), cov3D=_prep(cov3D_precomp, "cov3D_precomp", dev),
                       viewmatrix=_prep(viewmatrix, "viewmatrix", dev), projmatrix=_prep(projmatrix, "projmatrix", dev),
                       campos=_prep(campos, "campos", dev), radii=_prep(radii, "radii", dev, torch.int32),
                       alphas=_prep(alphas, "alphas", dev), g_c=_prep(dL_dout_color, "dL_dout_color", dev),
                       g_s=_prep(dL_dout_semantic, "dL_dout_semantic", dev),
                       g_d=_prep(dL_dout_depth, "dL_dout_depth", dev), g_a=_prep(dL_dout_alpha, "dL_dout_alpha", dev))
            sc = _scene(P, S, H, W, ten["bg"], ten["means3D"], ten["sh"], ten["colors"], ten["semantics"], None,
                        ten["scales"], ten["rotations"], scale_modifier, ten["cov3D"], ten["viewmatrix"],
                        ten["projmatrix"], tan_fovx, tan_fovy, degree, ten["campos"], False, debug)
            scratch = _backward_scratch(lib.goi_raster_backward_scratch_bytes(int(R), S), dev)
            r = lib.goi_raster_backward(
                C.byref(sc), int(R), _ptr(geomBuffer), _ptr(binningBuffer), _ptr(imageBuffer), _ptr(ten["radii"]),
                _ptr(ten["alphas"]), _ptr(ten["g_c"]), _ptr(ten["g_s"]), _ptr(ten["g_d"]), _ptr(ten["g_a"]),
                _ptr(dL_dmeans2D), _ptr(dL_dconic), _ptr(dL_dopacity), _ptr(dL_dcolors), _ptr(dL_dsemantics),
                _ptr(dL_ddepths), _ptr(dL_dmeans3D), _ptr(dL_dcov3D), _ptr(dL_dsh), _ptr(dL_dscales),
                _ptr(dL_drotations), _ptr(scratch), _stream(dev))
            if r < 0:
                raise RuntimeError(_lib.last_error())
    return (dL_dmeans2D, dL_dcolors, dL_dsemantics, dL_dopacity, dL_dmeans3D, dL_dcov3D, dL_dsh, dL_dscales,
            dL_drotations)


def rasterize_gaussians_backward(background, means3D, radii, colors, semantics, scales, rotations, scale_modifier,
                                 cov3D_precomp, viewmatrix, projmatrix, tan_fovx, tan_fovy, dL_dout_color,
                                 dL_dout_semantic, dL_dout_depth, dL_dout_alpha, sh, degree, campos, geomBuffer, R,
                                 binningBuffer, imageBuffer, alphas, debug):
    """-> (dL_dmeans2D[P,3], dL_dcolors[P,3], dL_dsemantics[P,S], dL_dopacity[P,1], dL_dmeans3D[P,3],
    dL_dcov3D[P,6], dL_dsh[P,M,3], dL_dscales[P,3], dL_drotations[P,4]) -- the reference's RasterizeGaussiansBackwardCUDA
    (rasterize_points.cu:213-306), same argument list."""
    return _backward_impl(background, means3D, radii, colors, semantics, scales, rotations, scale_modifier, cov3D_precomp, viewmatrix, projmatrix, tan_fovx, tan_fovy, dL_dout_color, dL_dout_semantic, dL_dout_depth, dL_dout_alpha, sh, degree, campos, geomBuffer, R, binningBuffer, imageBuffer, alphas, debug)


def rasterize_gaussians_backward_sh_factored(background, means3D, radii, colors, semantics, scales, rotations, scale_modifier,
                                 cov3D_precomp, viewmatrix, projmatrix, tan_fovx, tan_fovy, dL_dout_color,
                                 dL_dout_semantic, dL_dout_depth, dL_dout_alpha, sh, degree, campos, geomBuffer, R,
                                 binningBuffer, imageBuffer, alphas, debug):
    """Same arguments and result tuple as rasterize_gaussians_backward (not in the reference's pybind module), in the
    FACTORED mode of goi_raster_backward: dL_dsh is None and dL_dcolors is the clamp-masked colour gradient."""
    return _backward_impl(background, means3D, radii, colors, semantics, scales, rotations, scale_modifier, cov3D_precomp, viewmatrix, projmatrix, tan_fovx, tan_fovy, dL_dout_color, dL_dout_semantic, dL_dout_depth, dL_dout_alpha, sh, degree, campos, geomBuffer, R, binningBuffer, imageBuffer, alphas, debug, sh_factored=True)


def sh_grad_from_views(means3D, campos, gcol, degree, M):
    """dL/dSH [P,M,3] of V views from means3D [P,3], the camera centres campos [V,3] and the clamp-masked colour
    gradients gcol [V,P,3] that rasterize_gaussians_backward(sh_factored=True) returns as dL_dcolors
    (goi_raster_sh_grad_from_views): sum over the views, in index order, of basis(view direction) x gcol."""
    lib = _lib.load()
    dev = _check_device(means3D)
    P, V = int(means3D.size(0)), int(campos.size(0))
    if tuple(gcol.shape) != (V, P, 3):
        raise ValueError(f"gcol must be [V={V}, P={P}, 3], got {tuple(gcol.shape)}")
    with torch.cuda.device(dev):
        out = torch.empty((P, int(M), 3), dtype=torch.float32, device=dev)
        if P and V:
            m, c, g = _prep(means3D, "means3D", dev), _prep(campos, "campos", dev), _prep(gcol, "gcol", dev)
            if lib.goi_raster_sh_grad_from_views(P, int(degree), int(M), V, _ptr(m), _ptr(c), _ptr(g), _ptr(out), _stream(dev)) < 0:
                raise RuntimeError(_lib.last_error())
        elif P:
            out.zero_()
    return out


def rasterize_gaussians_backward_semantics(background, means3D, radii, semantics, viewmatrix, projmatrix, tan_fovx,
                                           tan_fovy, dL_dout_semantic, campos, geomBuffer, R, binningBuffer,
                                           imageBuffer, alphas, sh_degree=0, debug=False):
    """-> dL_dsemantics[P,S] only (not in the reference's pybind module): the feature-gradient-only
    backward behind goi_raster_backward_semantics, for training runs that optimise only the semantic
    features (the reference's default, arguments/__init__.py:85-90).  Bit-identical to the
    dL_dsemantics of rasterize_gaussians_backward."""
    lib = _lib.load()
    dev = _check_device(means3D)
    P = int(means3D.size(0))
    S = int(dL_dout_semantic.size(0))
    H, W = int(dL_dout_semantic.size(1)), int(dL_dout_semantic.size(2))
    with torch.cuda.device(dev):
        dL_dsemantics = torch.empty((P, S), dtype=torch.float32, device=dev)
        if P != 0:
            ten = dict(bg=_prep(background, "background", dev), means3D=_prep(means3D, "means3D", dev),
                       semantics=_prep(semantics, "semantics", dev), viewmatrix=_prep(viewmatrix, "viewmatrix", dev),
                       projmatrix=_prep(projmatrix, "projmatrix", dev), campos=_prep(campos, "campos", dev),
                       radii=_prep(radii, "radii", dev, torch.int32), alphas=_prep(alphas, "alphas", dev),
                       g_s=_prep(dL_dout_semantic, "dL_dout_semantic", dev))
            # the geometry inputs are not read by this path (everything it needs is in the forward's
            # workspaces); the scene only has to pass validation
            sc = _scene(P, S, H, W, ten["bg"], ten["means3D"], None, ten["means3D"], ten["semantics"], None,
                        None, None, 1.0, ten["means3D"], ten["viewmatrix"], ten["projmatrix"], tan_fovx, tan_fovy,
                        sh_degree, ten["campos"], False, debug)
            scratch = _backward_scratch(lib.goi_raster_backward_scratch_bytes(int(R), S), dev)
            r = lib.goi_raster_backward_semantics(C.byref(sc), int(R), _ptr(geomBuffer), _ptr(binningBuffer),
                                                  _ptr(imageBuffer), _ptr(ten["radii"]), _ptr(ten["alphas"]),
                                                  _ptr(ten["g_s"]), _ptr(dL_dsemantics), _ptr(scratch), _stream(dev))
            if r < 0:
                raise RuntimeError(_lib.last_error())
    return dL_dsemantics


def rasterize_gaussians_trace(background, means3D, colors, img_sem, opacity, scales, rotations, scale_modifier,
                              cov3D_precomp, viewmatrix, projmatrix, tan_fovx, tan_fovy, image_height, image_width, sh,
                              degree, campos, prefiltered, debug):
    """-> (num_rendered, color[3,H,W], gau_sem[P,S], num_gsem[P] i32, geomBuffer, binningBuffer, imgBuffer)"""
    lib = _lib.load()
    dev = _check_device(means3D)
    P, H, W = int(means3D.size(0)), int(image_height), int(image_width)
    if img_sem is None or img_sem.numel() == 0:
        raise RuntimeError("img_sem [S,H,W] is required")
    S = int(img_sem.size(0))
    f32 = dict(dtype=torch.float32, device=dev)
    with torch.cuda.device(dev):
        out_color = torch.empty((3, H, W), **f32)
        gau_sem = torch.zeros((P, S), **f32)
        num_gsem = torch.zeros((P,), dtype=torch.int32, device=dev)
        radii = torch.empty((P,), dtype=torch.int32, device=dev)
        ten = dict(bg=_prep(background, "background", dev), means3D=_prep(means3D, "means3D", dev),
                   sh=_prep(sh, "sh", dev), colors=_prep(colors, "colors_precomp", dev),
                   img=_prep(img_sem, "img_sem", dev), opacity=_prep(opacity, "opacities", dev),
                   scales=_prep(scales, "scales", dev), rotations=_prep(rotations, "rotations", dev),
                   cov3D=_prep(cov3D_precomp, "cov3D_precomp", dev), viewmatrix=_prep(viewmatrix, "viewmatrix", dev),
                   projmatrix=_prep(projmatrix, "projmatrix", dev), campos=_prep(campos, "campos", dev))
        geom = torch.empty(lib.goi_raster_geom_bytes(P) if P > 0 else 0, dtype=torch.uint8, device=dev)
        img = torch.empty(lib.goi_raster_image_bytes(W, H) if P > 0 else 0, dtype=torch.uint8, device=dev)
        alloc = _BinningAllocator(dev)
        sc = _scene(P, S, H, W, ten["bg"], ten["means3D"], ten["sh"], ten["colors"], None, ten["opacity"],
                    ten["scales"], ten["rotations"], scale_modifier, ten["cov3D"], ten["viewmatrix"],
                    ten["projmatrix"], tan_fovx, tan_fovy, degree, ten["campos"], prefiltered, debug)
        n = lib.goi_raster_trace(C.byref(sc), _ptr(ten["img"]), _ptr(geom), _ptr(img), alloc.cb, None, _ptr(out_color),
                                 _ptr(gau_sem), _ptr(num_gsem), _ptr(radii), _stream(dev))
        if alloc.error is not None:
            raise alloc.error
        if n < 0:
            raise RuntimeError(_lib.last_error())
    return n, out_color, gau_sem, num_gsem, geom, alloc.tensor, img


def mark_visible(means3D, viewmatrix, projmatrix):
    """-> bool[P]: view-space z > 0.2 (cuda_rasterizer/auxiliary.h:139-164)."""
    lib = _lib.load()
    dev = _check_device(means3D)
    P = int(means3D.size(0))
    with torch.cuda.device(dev):
        present = torch.zeros((P,), dtype=torch.bool, device=dev)
        if P != 0:
            m = _prep(means3D, "means3D", dev)
            v = _prep(viewmatrix, "viewmatrix", dev)
            p = _prep(projmatrix, "projmatrix", dev)
            if lib.goi_raster_mark_visible(P, _ptr(m), _ptr(v), _ptr(p), C.c_void_p(present.data_ptr()),
                                           _stream(dev)) < 0:
                raise RuntimeError(_lib.last_error())
    return present


def debug_views(P, W, H, R, geomBuffer, binningBuffer, imgBuffer):
    """Tests only: decoded copies of the opaque workspaces as a dict of tensors."""
    lib = _lib.load()
    dev = geomBuffer.device
    T = ((W + 15) // 16) * ((H + 15) // 16)
    with torch.cuda.device(dev):
        out = dict(depths=torch.zeros(P, device=dev), means2D=torch.zeros(P, 2, device=dev),
                   conic_opacity=torch.zeros(P, 4, device=dev), rgb=torch.zeros(P, 3, device=dev),
                   tiles_touched=torch.zeros(P, dtype=torch.int32, device=dev),
                   point_list=torch.zeros(max(R, 0), dtype=torch.int32, device=dev),
                   ranges=torch.zeros(T, 2, dtype=torch.int32, device=dev),
                   n_contrib=torch.zeros(H * W, dtype=torch.int32, device=dev))
        r = lib.goi_raster_debug_views(P, W, H, R, _ptr(geomBuffer), _ptr(binningBuffer), _ptr(imgBuffer),
                                       *[_ptr(out[k]) for k in ("depths", "means2D", "conic_opacity", "rgb",
                                                                "tiles_touched", "point_list", "ranges", "n_contrib")],
                                       _stream(dev))
        if r < 0:
            raise RuntimeError(_lib.last_error())
        torch.cuda.synchronize(dev)
    return out
