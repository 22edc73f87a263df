"""Drop-in package name of the reference's rasterizer.

`from diff_gaussian_rasterization import GaussianRasterizationSettings, GaussianRasterizer`
(gaussian_renderer/__init__.py:14, gui/gs_renderer.py:10-13) resolves here when the repository
root is on sys.path; everything is implemented in goi_hyperplane_amd (hand-written HIP for gfx950
behind the C ABI of include/goi_raster.h).  `_C` exposes the reference's four pybind functions.
"""
from goi_hyperplane_amd import _C  # noqa: F401
from goi_hyperplane_amd.rasterizer import (  # noqa: F401
    GaussianRasterizationSettings,
    GaussianRasterizer,
    _RasterizeGaussians,
    rasterize_gaussians,
    trace_gaussians,
)

__all__ = ["GaussianRasterizationSettings", "GaussianRasterizer", "rasterize_gaussians", "trace_gaussians", "_C"]
