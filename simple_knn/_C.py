"""The one function the reference's pybind module exports (submodules/simple-knn/ext.cpp:15-17)."""
from goi_hyperplane_amd.knn import distCUDA2  # noqa: F401

__all__ = ["distCUDA2"]
