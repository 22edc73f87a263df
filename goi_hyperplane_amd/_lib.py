"""ctypes loader for libgoi_raster.so (the C ABI of include/goi_raster.h).

The library is hand-written HIP for gfx950 and is the ONLY compute path of this package: there is
no CPU or PyTorch fallback.  Loading fails loudly when the shared object is missing; calling a
compute entry point without a GPU fails inside HIP with a clear error.
"""
from __future__ import annotations

import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "lib", "libgoi_raster.so")

ABI_VERSION = 6

STAGES = ("preprocess", "depth_sort", "scan", "emit", "tile_sort", "ranges", "blend_fwd", "blend_bwd",
          "preprocess_bwd")


class GoiRasterScene(C.Structure):
    """Mirror of `struct GoiRasterScene` (include/goi_raster.h)."""
    _fields_ = [
        ("P", C.c_int), ("D", C.c_int), ("M", C.c_int), ("S", C.c_int), ("W", C.c_int), ("H", C.c_int),
        ("bg", C.c_void_p), ("means3D", C.c_void_p), ("shs", C.c_void_p), ("colors_precomp", C.c_void_p),
        ("semantics", C.c_void_p), ("opacities", C.c_void_p), ("scales", C.c_void_p),
        ("scale_modifier", C.c_float), ("rotations", C.c_void_p), ("cov3D_precomp", C.c_void_p),
        ("viewmatrix", C.c_void_p), ("projmatrix", C.c_void_p), ("campos", C.c_void_p),
        ("tan_fovx", C.c_float), ("tan_fovy", C.c_float), ("prefiltered", C.c_int), ("debug", C.c_int),
    ]


ALLOC_FN = C.CFUNCTYPE(C.c_void_p, C.c_void_p, C.c_size_t)

# name -> (restype, argtypes); every symbol include/goi_raster.h declares
SYMBOLS = {
    "goi_raster_abi_version": (C.c_int, []),
    "goi_raster_last_error": (C.c_char_p, []),
    "goi_raster_geom_bytes": (C.c_size_t, [C.c_int]),
    "goi_raster_image_bytes": (C.c_size_t, [C.c_int, C.c_int]),
    "goi_raster_binning_bytes": (C.c_size_t, [C.c_int]),
    "goi_raster_backward_scratch_bytes": (C.c_size_t, [C.c_int, C.c_int]),
    "goi_raster_forward": (C.c_int, [C.POINTER(GoiRasterScene), C.c_void_p, C.c_void_p, ALLOC_FN, C.c_void_p]
                           + [C.c_void_p] * 5 + [C.c_void_p]),
    "goi_raster_forward_async": (C.c_int, [C.POINTER(GoiRasterScene), C.c_void_p, C.c_void_p, C.c_void_p, C.c_int]
                                 + [C.c_void_p] * 5 + [C.c_void_p]),
    "goi_raster_ticket_result": (C.c_int, [C.c_int, C.c_int, C.POINTER(C.c_int)]),
    "goi_raster_forward_async_cut": (C.c_int, [C.POINTER(GoiRasterScene), C.c_void_p, C.c_void_p, C.c_void_p, C.c_int]
                                     + [C.c_void_p] * 5 + [C.c_void_p, C.c_void_p, C.c_void_p]),
    "goi_raster_ticket_result2": (C.c_int, [C.c_int, C.c_int, C.POINTER(C.c_int), C.POINTER(C.c_uint)]),
    "goi_raster_forward_redo": (C.c_int, [C.POINTER(GoiRasterScene), C.c_int, C.c_void_p, C.c_void_p, C.c_void_p]
                                + [C.c_void_p] * 5 + [C.c_void_p]),
    "goi_raster_forward_reblend": (C.c_int, [C.POINTER(GoiRasterScene), C.c_int] + [C.c_void_p] * 9),
    "goi_raster_backward": (C.c_int, [C.POINTER(GoiRasterScene), C.c_int] + [C.c_void_p] * 3 + [C.c_void_p] * 6
                            + [C.c_void_p] * 11 + [C.c_void_p, C.c_void_p]),
    "goi_raster_backward2": (C.c_int, [C.POINTER(GoiRasterScene), C.c_int] + [C.c_void_p] * 3 + [C.c_void_p] * 6
                             + [C.c_void_p] * 11 + [C.c_void_p, C.c_void_p, C.c_void_p]),
    "goi_raster_backward3": (C.c_int, [C.POINTER(GoiRasterScene), C.c_int, C.c_int, C.c_int] + [C.c_void_p] * 3 + [C.c_void_p] * 6
                             + [C.c_void_p] * 11 + [C.c_void_p, C.c_void_p, C.c_void_p]),
    "goi_raster_backward_semantics": (C.c_int, [C.POINTER(GoiRasterScene), C.c_int] + [C.c_void_p] * 9),
    "goi_raster_trace": (C.c_int, [C.POINTER(GoiRasterScene), C.c_void_p, C.c_void_p, C.c_void_p, ALLOC_FN,
                                   C.c_void_p] + [C.c_void_p] * 4 + [C.c_void_p]),
    "goi_raster_mark_visible": (C.c_int, [C.c_int] + [C.c_void_p] * 4 + [C.c_void_p]),
    "goi_codebook_loss_partial_rows": (C.c_int, []),
    "goi_codebook_loss_rows": (C.c_int, [C.c_void_p] * 5 + [C.c_longlong, C.c_int, C.c_int, C.c_float] + [C.c_void_p] * 4),
    "goi_codebook_sim_workspace_bytes": (C.c_size_t, []),
    "goi_codebook_sim": (C.c_int, [C.c_void_p, C.c_void_p, C.c_longlong, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p,
                                   C.c_void_p]),
    "goi_codebook_dlut_partial_blocks": (C.c_int, []),
    "goi_codebook_dlut": (C.c_int, [C.c_void_p, C.c_void_p, C.c_longlong, C.c_int, C.c_int, C.c_void_p, C.c_void_p]),
    "goi_codebook_fused_workspace_bytes": (C.c_size_t, [C.c_longlong]),
    "goi_codebook_fused_partial_rows": (C.c_int, []),
    "goi_codebook_fused": (C.c_int, [C.c_void_p] * 5 + [C.c_longlong, C.c_int, C.c_int, C.c_int, C.c_float] + [C.c_void_p] * 5),
    "goi_adam_step": (C.c_int, [C.c_void_p, C.c_int, C.c_double, C.c_double, C.c_double, C.c_void_p, C.c_void_p]),
    "goi_adam_step_guarded": (C.c_int, [C.c_void_p, C.c_int, C.c_double, C.c_double, C.c_double, C.c_void_p, C.c_void_p,
                                        C.c_void_p]),
    "goi_raster_truncated_flag": (C.c_void_p, [C.c_void_p, C.c_int]),
    "goi_knn_workspace_bytes": (C.c_size_t, [C.c_int]),
    "goi_knn_dist2": (C.c_int, [C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    "goi_raster_sh_grad_from_views": (C.c_int, [C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
                                        C.c_void_p]),
    "goi_raster_profile_enable": (None, [C.c_int]),
    "goi_raster_profile_stages": (None, [C.c_uint]),
    "goi_raster_profile_collect": (C.c_int, [C.POINTER(C.c_double), C.POINTER(C.c_int)]),
    "goi_semantic_decode": (C.c_int, [C.c_void_p, C.c_int, C.c_longlong, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p,
                                      C.c_float, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    "goi_raster_set_option": (C.c_int, [C.c_char_p, C.c_int]),
    "goi_raster_blend_stats": (C.c_int, [C.c_int] * 4 + [C.c_void_p] * 5),
    "goi_raster_debug_views": (C.c_int, [C.c_int] * 4 + [C.c_void_p] * 3 + [C.c_void_p] * 8 + [C.c_void_p]),
}

_lib = None


def load():
    """Returns the loaded library; raises if it has not been built (python -m goi_hyperplane_amd.build)."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise ImportError(
                f"{LIB_PATH} is missing: the HIP extension has not been built. Run "
                "`python -m goi_hyperplane_amd.build` (needs hipcc; cross-compiles for gfx950 without a GPU). "
                "There is deliberately no CPU fallback.")
        # The tensors this library is handed live in the HIP runtime PyTorch loaded: that copy of libamdhip64 must
        # be the one in the process before ours resolves its dependency (loaded first, libgoi_raster.so would pull in
        # the system copy and the two runtimes would not see each other's devices and allocations).
        import torch  # noqa: F401
        lib = C.CDLL(LIB_PATH)
        for name, (res, args) in SYMBOLS.items():
            fn = getattr(lib, name)  # AttributeError if the .so does not export a declared symbol
            fn.restype = res
            fn.argtypes = args
        got = lib.goi_raster_abi_version()
        if got != ABI_VERSION:
            raise ImportError(f"libgoi_raster.so ABI {got} != expected {ABI_VERSION}; rebuild")
        _lib = lib
        # experiment switches, e.g. GOI_OPTIONS="sort_variant=1,bwd_variant=1"
        for item in filter(None, os.environ.get("GOI_OPTIONS", "").split(",")):
            name, _, value = item.partition("=")
            if lib.goi_raster_set_option(name.strip().encode(), int(value)) < 0:
                raise RuntimeError(lib.goi_raster_last_error().decode())
            OPTIONS[name.strip()] = int(value)
    return _lib


def last_error() -> str:
    return load().goi_raster_last_error().decode("utf-8", "replace")


class GoiAdamGroup(C.Structure):
    """include/goi_raster.h: GoiAdamGroup"""
    _fields_ = [("param", C.c_void_p), ("grad", C.c_void_p), ("exp_avg", C.c_void_p), ("exp_avg_sq", C.c_void_p),
                ("numel", C.c_longlong), ("row_len", C.c_int), ("step_size", C.c_float), ("bc2_sqrt", C.c_float)]


ADAM_MAX_GROUPS = 8


OPTIONS = {}  # the switches set through set_option / GOI_OPTIONS in this process (name -> value)


def set_option(name: str, value: int) -> None:
    if load().goi_raster_set_option(name.encode(), int(value)) < 0:
        raise RuntimeError(last_error())
    OPTIONS[name] = int(value)


def profile_enable(on: bool) -> None:
    load().goi_raster_profile_enable(1 if on else 0)


def profile_stages(names) -> None:
    """Record events only around the named stages (cheaper than profile_enable inside a timed region)."""
    load().goi_raster_profile_stages(sum(1 << STAGES.index(n) for n in names))


def profile_collect() -> dict:
    """{stage: (milliseconds, launches)} accumulated since the previous collect."""
    n = len(STAGES)
    ms = (C.c_double * n)()
    calls = (C.c_int * n)()
    if load().goi_raster_profile_collect(ms, calls) < 0:
        raise RuntimeError(last_error())
    return {STAGES[i]: (ms[i], calls[i]) for i in range(n)}
