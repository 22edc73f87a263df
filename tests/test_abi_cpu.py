"""CPU-only checks of the product boundary (no GPU, no compute calls):
  * libgoi_raster.so builds, loads and exports every symbol include/goi_raster.h declares;
  * the ctypes mirror of GoiRasterScene has the C layout;
  * workspace-size functions behave (monotone, non-zero);
  * the Python operator surface has the reference's names, parameter order and defaults
    (diff_gaussian_rasterization/__init__.py:21-69,246-349 of the reference) and raises the
    reference's messages for invalid argument combinations;
  * the product path never falls back to the oracle or to a CPU implementation.
"""
import ctypes as C
import inspect
import os
import re

import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def lib():
    from goi_hyperplane_amd import build
    build.build()
    from goi_hyperplane_amd import _lib
    return _lib.load()


def test_library_exports_every_declared_symbol(lib):
    hdr = open(os.path.join(ROOT, "include", "goi_raster.h")).read()
    declared = set(re.findall(r"\b(goi_(?:raster|semantic|knn|adam|codebook)_[a-z_0-9]+)\s*\(", hdr))
    assert {"goi_raster_forward", "goi_raster_backward", "goi_raster_trace", "goi_raster_mark_visible",
            "goi_raster_geom_bytes", "goi_raster_image_bytes", "goi_raster_binning_bytes",
            "goi_raster_last_error", "goi_raster_abi_version"} <= declared
    from goi_hyperplane_amd import _lib
    assert declared == set(_lib.SYMBOLS), "ctypes table and header disagree"
    for name in declared:
        assert hasattr(lib, name), f"{name} is declared in goi_raster.h but not exported"
    assert lib.goi_raster_abi_version() == _lib.ABI_VERSION == int(re.search(r"#define GOI_RASTER_ABI_VERSION (\d+)", hdr).group(1))


def test_compiled_binding_is_built_and_exports_the_reference_surface(lib):
    """lib/_goi_C.so (csrc/torch_binding.cpp, host C++ against the torch headers): the four functions of the reference's
    pybind module (ext.cpp:15-20) under the reference's names, built against the same ABI version."""
    from goi_hyperplane_amd import _C, _lib
    from goi_hyperplane_amd.build import EXT
    assert os.path.exists(EXT), "build() must produce the compiled binding"
    _C.set_binding("compiled")
    ext = _C._ext()
    for name in ("rasterize_gaussians", "rasterize_gaussians_backward", "rasterize_gaussians_trace", "mark_visible"):
        assert callable(getattr(ext, name)), name
    assert ext.abi_version() == _lib.ABI_VERSION
    assert _C.binding() == "compiled"
    import torch
    with pytest.raises(RuntimeError, match="no CPU fallback"):  # the loud CPU refusal survives the binding
        ext.mark_visible(torch.zeros(4, 3), torch.eye(4), torch.eye(4))


def test_scene_struct_layout_matches_c(tmp_path):
    """The ctypes mirror must have exactly the layout gcc gives `struct GoiRasterScene`."""
    import subprocess
    from goi_hyperplane_amd._lib import GoiRasterScene
    fields = [f[0] for f in GoiRasterScene._fields_]
    src = tmp_path / "layout.c"
    src.write_text('#include <stdio.h>\n#include <stddef.h>\n#include "goi_raster.h"\nint main(void){\n'
                   + "".join(f'printf("{f} %zu\\n", offsetof(GoiRasterScene, {f}));\n' for f in fields)
                   + 'printf("sizeof %zu\\n", sizeof(GoiRasterScene));return 0;}\n')
    exe = tmp_path / "layout"
    subprocess.check_call(["gcc", "-I", os.path.join(ROOT, "include"), str(src), "-o", str(exe)])
    out = dict(line.split() for line in subprocess.check_output([str(exe)], text=True).splitlines())
    for f in fields:
        assert getattr(GoiRasterScene, f).offset == int(out[f]), f
    assert C.sizeof(GoiRasterScene) == int(out["sizeof"])


def test_adam_group_struct_layout_matches_c(tmp_path):
    import subprocess
    from goi_hyperplane_amd._lib import GoiAdamGroup
    fields = [f[0] for f in GoiAdamGroup._fields_]
    src = tmp_path / "layout.c"
    src.write_text('#include <stdio.h>\n#include <stddef.h>\n#include "goi_raster.h"\nint main(void){\n'
                   + "".join(f'printf("{f} %zu\\n", offsetof(GoiAdamGroup, {f}));\n' for f in fields)
                   + 'printf("sizeof %zu\\n", sizeof(GoiAdamGroup));return 0;}\n')
    exe = tmp_path / "layout"
    subprocess.check_call(["gcc", "-I", os.path.join(ROOT, "include"), str(src), "-o", str(exe)])
    out = dict(line.split() for line in subprocess.check_output([str(exe)], text=True).splitlines())
    for f in fields:
        assert getattr(GoiAdamGroup, f).offset == int(out[f]), f
    assert C.sizeof(GoiAdamGroup) == int(out["sizeof"])


def test_workspace_sizes(lib):
    g1, g2 = lib.goi_raster_geom_bytes(1000), lib.goi_raster_geom_bytes(1_000_000)
    assert 0 < g1 < g2 and g2 < 200 * 1_000_000  # < 200 B per Gaussian
    assert lib.goi_raster_image_bytes(1600, 1056) >= 1600 * 1056 * 4
    b0, b1 = lib.goi_raster_binning_bytes(0), lib.goi_raster_binning_bytes(8_000_000)
    assert 0 < b0 < b1 and b1 < 40 * 8_000_000  # < 40 B per instance


def test_python_surface_matches_reference_signatures():
    import diff_gaussian_rasterization as dgr
    from goi_hyperplane_amd import _C
    assert dgr.GaussianRasterizationSettings._fields == (
        "image_height", "image_width", "tanfovx", "tanfovy", "bg", "scale_modifier", "viewmatrix", "projmatrix",
        "sh_degree", "campos", "prefiltered", "debug")
    fwd = inspect.signature(dgr.GaussianRasterizer.forward)
    assert list(fwd.parameters) == ["self", "means3D", "means2D", "opacities", "shs", "colors_precomp", "semantics",
                                    "scales", "rotations", "cov3D_precomp"]
    assert all(fwd.parameters[p].default is None for p in list(fwd.parameters)[4:])
    tr = inspect.signature(dgr.GaussianRasterizer.trace)
    assert list(tr.parameters) == ["self", "means3D", "means2D", "opacities", "shs", "colors_precomp", "img_sem",
                                   "scales", "rotations", "cov3D_precomp"]
    assert list(inspect.signature(dgr.rasterize_gaussians).parameters) == [
        "means3D", "means2D", "sh", "colors_precomp", "semantics", "opacities", "scales", "rotations",
        "cov3Ds_precomp", "raster_settings"]
    # the four pybind entry points (ext.cpp:15-20), positional arities 20 / 26 / 20 / 3
    assert len(inspect.signature(_C.rasterize_gaussians).parameters) == 20
    assert len(inspect.signature(_C.rasterize_gaussians_backward).parameters) == 26
    assert len(inspect.signature(_C.rasterize_gaussians_trace).parameters) == 20
    assert len(inspect.signature(_C.mark_visible).parameters) == 3
    assert hasattr(dgr, "_C") and hasattr(dgr._RasterizeGaussians, "trace")


def test_argument_validation_messages_and_no_cpu_fallback():
    from goi_hyperplane_amd.rasterizer import GaussianRasterizationSettings, GaussianRasterizer
    z = torch.zeros
    rs = GaussianRasterizationSettings(32, 32, 0.5, 0.5, z(3), 1.0, torch.eye(4), torch.eye(4), 3, z(3), False, False)
    r = GaussianRasterizer(rs)
    m = z(4, 3)
    with pytest.raises(Exception, match="excatly one of either SHs or precomputed colors"):
        r(m, m, z(4, 1), scales=z(4, 3), rotations=z(4, 4), semantics=z(4, 10))
    with pytest.raises(Exception, match="excatly one of either SHs or precomputed colors"):
        r(m, m, z(4, 1), shs=z(4, 16, 3), colors_precomp=z(4, 3), scales=z(4, 3), rotations=z(4, 4))
    with pytest.raises(Exception, match="exactly one of either scale/rotation pair"):
        r(m, m, z(4, 1), colors_precomp=z(4, 3), semantics=z(4, 10))
    with pytest.raises(Exception, match="exactly one of either scale/rotation pair"):
        r(m, m, z(4, 1), colors_precomp=z(4, 3), scales=z(4, 3), rotations=z(4, 4), cov3D_precomp=z(4, 6))
    # CPU tensors are refused loudly: the package has no CPU path
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        r(m, m, z(4, 1), colors_precomp=z(4, 3), scales=z(4, 3), rotations=z(4, 4), semantics=z(4, 10))


def test_product_never_imports_the_oracle():
    pkg = os.path.join(ROOT, "goi_hyperplane_amd")
    offenders = []
    for dirpath, _dirs, files in os.walk(pkg):
        if os.sep + "build" in dirpath:
            continue
        for f in files:
            if f.endswith((".py", ".hip", ".h", ".cpp")):
                txt = open(os.path.join(dirpath, f), errors="replace").read()
                if re.search(r"^\s*(from|import)\s+oracle\b", txt, re.M) or "liboracle" in txt or "goi_oracle" in txt:
                    offenders.append(os.path.join(dirpath, f))
    assert not offenders, f"product code references the oracle: {offenders}"
    for name in ("diff_gaussian_rasterization/__init__.py",):
        assert "oracle" not in open(os.path.join(ROOT, name)).read()


def test_missing_library_fails_loudly(monkeypatch):
    from goi_hyperplane_amd import _lib
    monkeypatch.setattr(_lib, "_lib", None)
    monkeypatch.setattr(_lib, "LIB_PATH", os.path.join(ROOT, "does", "not", "exist.so"))
    with pytest.raises(ImportError, match="no CPU fallback"):
        _lib.load()
