"""The reference's OWN caller, executed unchanged against this repository's drop-in package (runs only where
/root/reference exists, i.e. in the authoring container; the GPU box has no reference and skips).

gaussian_renderer/__init__.py:14 does `from diff_gaussian_rasterization import GaussianRasterizationSettings,
GaussianRasterizer` and, in render() (:18-105) and trace() (:107-190), builds the settings tuple and calls the module with
keyword arguments.  Here that file is imported as it is -- only `scene.gaussian_model` (needed for a type annotation;
its own imports want plyfile and a CUDA device) is replaced by an empty stand-in module -- and its render() / trace() are
called on CPU tensors.  There is no GPU in this container, so the call must travel through the reference's code, bind every
field and keyword of this package's API, and stop at the one place this package refuses: "no CPU fallback".  Any mismatch
in names, field order or keywords would raise something else (TypeError) first."""
import importlib
import os
import sys
import types

import pytest
import torch

REF = "/root/reference"
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
pytestmark = pytest.mark.skipif(not os.path.isdir(os.path.join(REF, "gaussian_renderer")),
                                reason="the reference checkout is only present in the authoring container")


@pytest.fixture()
def ref_renderer(monkeypatch):
    for name in ("scene", "scene.gaussian_model"):
        m = types.ModuleType(name)
        m.__path__ = []  # a package without content
        monkeypatch.setitem(sys.modules, name, m)
    sys.modules["scene.gaussian_model"].GaussianModel = object
    monkeypatch.syspath_prepend(REF)
    monkeypatch.syspath_prepend(ROOT)  # this repository's diff_gaussian_rasterization wins
    for name in ("gaussian_renderer", "utils", "utils.sh_utils"):
        sys.modules.pop(name, None)
    mod = importlib.import_module("gaussian_renderer")
    assert mod.__file__.startswith(REF)
    import diff_gaussian_rasterization as dgr
    assert dgr.__file__.startswith(ROOT) and mod.GaussianRasterizer is dgr.GaussianRasterizer
    # the reference allocates its screen-space placeholder with device="cuda" (:26); there is none here
    real = torch.zeros_like
    monkeypatch.setattr(torch, "zeros_like", lambda t, **k: real(t, **{kk: v for kk, v in k.items() if kk != "device"}))
    yield mod
    for name in ("gaussian_renderer", "utils", "utils.sh_utils"):
        sys.modules.pop(name, None)


def _model_and_camera():
    from goi_hyperplane_amd.render import GaussianSet, TorchCamera
    from goi_hyperplane_amd.scene import make_camera, make_scene
    pc = GaussianSet.from_scene(make_scene(50, S=10, seed=0), torch.device("cpu"))
    cam = TorchCamera(make_camera(64, 48), torch.device("cpu"))
    return pc, cam


class _Pipe:
    convert_SHs_python = False
    compute_cov3D_python = False
    debug = False


@pytest.mark.parametrize("python_paths", [False, True])
def test_reference_render_reaches_this_packages_operator(ref_renderer, python_paths):
    pc, cam = _model_and_camera()
    pipe = _Pipe()
    pipe.convert_SHs_python = pipe.compute_cov3D_python = python_paths
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        ref_renderer.render(cam, pc, pipe, torch.zeros(3))


def test_reference_trace_reaches_this_packages_operator(ref_renderer):
    pc, cam = _model_and_camera()
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        ref_renderer.trace(cam, pc, torch.zeros(10, 48, 64), _Pipe(), torch.zeros(3))
