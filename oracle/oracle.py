"""ctypes wrapper of the CPU oracle (liboracle.so).  TEST INFRASTRUCTURE ONLY.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module; the
shipped package (goi_hyperplane_amd) never does.  All arrays are numpy float32, C-contiguous.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess
from dataclasses import dataclass

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))


class _Scene(C.Structure):
    _fields_ = [
        ("P", C.c_int), ("D", C.c_int), ("M", C.c_int), ("S", C.c_int), ("W", C.c_int), ("H", C.c_int),
        ("bg", C.c_void_p), ("means3D", C.c_void_p), ("shs", C.c_void_p), ("colors_precomp", C.c_void_p),
        ("semantics", C.c_void_p), ("opacities", C.c_void_p), ("scales", C.c_void_p),
        ("scale_modifier", C.c_float), ("rotations", C.c_void_p), ("cov3D_precomp", C.c_void_p),
        ("viewmatrix", C.c_void_p), ("projmatrix", C.c_void_p), ("campos", C.c_void_p),
        ("tan_fovx", C.c_float), ("tan_fovy", C.c_float), ("prefiltered", C.c_int),
    ]


def build(force: bool = False) -> None:
    """Compile liboracle.so / liboracle_fma.so with the committed Makefile (g++ only)."""
    if force or not all(os.path.exists(os.path.join(_HERE, n)) for n in ("liboracle.so", "liboracle_f64p.so", "libknn_oracle.so")):
        subprocess.check_call(["make", "-C", _HERE, "all"], stdout=subprocess.DEVNULL)


_libs: dict = {}


def _lib(variant: str = ""):
    name = {"fma": "liboracle_fma.so", "f64power": "liboracle_f64p.so"}.get(variant, "liboracle.so")
    if name not in _libs:
        path = os.path.join(_HERE, name)
        if not os.path.exists(path):
            build(force=True)
        lib = C.CDLL(path)
        lib.goi_oracle_state_new.restype = C.c_void_p
        lib.goi_oracle_state_free.argtypes = [C.c_void_p]
        lib.goi_oracle_forward.restype = C.c_int
        lib.goi_oracle_forward.argtypes = [C.POINTER(_Scene), C.c_void_p] + [C.c_void_p] * 6 + [C.c_float, C.c_int]
        lib.goi_oracle_backward.restype = C.c_int
        lib.goi_oracle_backward.argtypes = [C.POINTER(_Scene), C.c_void_p] + [C.c_void_p] * 16 + [C.c_int]
        lib.goi_oracle_trace.restype = C.c_int
        lib.goi_oracle_trace.argtypes = [C.POINTER(_Scene), C.c_void_p] + [C.c_void_p] * 5 + [C.c_int]
        lib.goi_oracle_mark_visible.argtypes = [C.c_int] + [C.c_void_p] * 4
        lib.goi_oracle_state_counts.argtypes = [C.c_void_p] + [C.POINTER(C.c_int)] * 3
        lib.goi_oracle_state_get.argtypes = [C.c_void_p] + [C.c_void_p] * 11
        _libs[name] = lib
    return _libs[name]


def _f32(a):
    if a is None:
        return None
    return np.ascontiguousarray(a, dtype=np.float32)


def _ptr(a):
    return None if a is None else a.ctypes.data_as(C.c_void_p)


@dataclass
class ForwardResult:
    num_rendered: int
    color: np.ndarray
    semantic: np.ndarray
    depth: np.ndarray
    alpha: np.ndarray
    radii: np.ndarray
    fragile: np.ndarray


class Oracle:
    """One rasterizer call context: holds the inputs (kept alive) and the forward state."""

    def __init__(self, *, W, H, bg, means3D, opacities, semantics, viewmatrix, projmatrix, campos, tan_fovx,
                 tan_fovy, shs=None, colors_precomp=None, scales=None, rotations=None, cov3D_precomp=None,
                 sh_degree=3, scale_modifier=1.0, prefiltered=False, variant="", threads=0):
        if (shs is None) == (colors_precomp is None):
            raise ValueError("exactly one of shs / colors_precomp")
        if ((scales is None or rotations is None) and cov3D_precomp is None) or (
                (scales is not None or rotations is not None) and cov3D_precomp is not None):
            raise ValueError("exactly one of scales+rotations / cov3D_precomp")
        self.lib = _lib(variant)
        self.threads = int(threads)
        self.W, self.H = int(W), int(H)
        self.a = dict(
            bg=_f32(bg), means3D=_f32(means3D).reshape(-1, 3), shs=_f32(shs), colors_precomp=_f32(colors_precomp),
            semantics=_f32(semantics), opacities=_f32(opacities).reshape(-1), scales=_f32(scales),
            rotations=_f32(rotations), cov3D_precomp=_f32(cov3D_precomp), viewmatrix=_f32(viewmatrix).reshape(16),
            projmatrix=_f32(projmatrix).reshape(16), campos=_f32(campos).reshape(3))
        a = self.a
        self.P = a["means3D"].shape[0]
        self.S = a["semantics"].shape[1]
        self.M = 0 if a["shs"] is None else a["shs"].shape[1]
        assert a["semantics"].shape[0] == self.P
        self.sc = _Scene(self.P, int(sh_degree), self.M, self.S, self.W, self.H, _ptr(a["bg"]), _ptr(a["means3D"]),
                         _ptr(a["shs"]), _ptr(a["colors_precomp"]), _ptr(a["semantics"]), _ptr(a["opacities"]),
                         _ptr(a["scales"]), float(scale_modifier), _ptr(a["rotations"]), _ptr(a["cov3D_precomp"]),
                         _ptr(a["viewmatrix"]), _ptr(a["projmatrix"]), _ptr(a["campos"]), float(tan_fovx),
                         float(tan_fovy), int(bool(prefiltered)))
        self.st = C.c_void_p(self.lib.goi_oracle_state_new())
        self.fwd: ForwardResult | None = None

    def __del__(self):
        try:
            self.lib.goi_oracle_state_free(self.st)
        except Exception:
            pass

    def forward(self, fragile_eps: float = 1e-4) -> ForwardResult:
        W, H, S, P = self.W, self.H, self.S, self.P
        color = np.zeros((3, H, W), np.float32)
        sem = np.zeros((S, H, W), np.float32)
        depth = np.zeros((1, H, W), np.float32)
        alpha = np.zeros((1, H, W), np.float32)
        radii = np.zeros((P,), np.int32)
        fragile = np.zeros((H, W), np.uint8)
        n = self.lib.goi_oracle_forward(C.byref(self.sc), self.st, _ptr(color), _ptr(sem), _ptr(depth), _ptr(alpha),
                                        _ptr(radii), _ptr(fragile), float(fragile_eps), self.threads)
        if n < 0:
            raise RuntimeError(f"oracle forward failed ({n})")
        self.fwd = ForwardResult(n, color, sem, depth, alpha, radii, fragile)
        return self.fwd

    def backward(self, dL_dcolor, dL_dsem, dL_ddepth=None, dL_dalpha=None) -> dict:
        assert self.fwd is not None, "call forward() first"
        W, H, S, P, M = self.W, self.H, self.S, self.P, self.M
        g_c = _f32(dL_dcolor).reshape(3, H, W)
        g_s = _f32(dL_dsem).reshape(S, H, W)
        g_d = np.zeros((H, W), np.float32) if dL_ddepth is None else _f32(dL_ddepth).reshape(H, W)
        g_a = np.zeros((H, W), np.float32) if dL_dalpha is None else _f32(dL_dalpha).reshape(H, W)
        o = dict(means2D=np.zeros((P, 3), np.float32), conic=np.zeros((P, 4), np.float32),
                 opacity=np.zeros((P, 1), np.float32), colors=np.zeros((P, 3), np.float32),
                 semantics=np.zeros((P, S), np.float32), depths=np.zeros((P, 1), np.float32),
                 means3D=np.zeros((P, 3), np.float32), cov3D=np.zeros((P, 6), np.float32),
                 sh=np.zeros((P, M, 3), np.float32), scales=np.zeros((P, 3), np.float32),
                 rotations=np.zeros((P, 4), np.float32))
        r = self.lib.goi_oracle_backward(
            C.byref(self.sc), self.st, _ptr(self.fwd.alpha), _ptr(g_c), _ptr(g_s), _ptr(g_d), _ptr(g_a),
            _ptr(o["means2D"]), _ptr(o["conic"]), _ptr(o["opacity"]), _ptr(o["colors"]), _ptr(o["semantics"]),
            _ptr(o["depths"]), _ptr(o["means3D"]), _ptr(o["cov3D"]), _ptr(o["sh"]) if M > 0 else None,
            _ptr(o["scales"]), _ptr(o["rotations"]), self.threads)
        if r < 0:
            raise RuntimeError(f"oracle backward failed ({r})")
        return o

    def trace(self, img_sem):
        W, H, S, P = self.W, self.H, self.S, self.P
        img = _f32(img_sem).reshape(S, H, W)
        color = np.zeros((3, H, W), np.float32)
        gau_sem = np.zeros((P, S), np.float32)
        num = np.zeros((P,), np.int32)
        radii = np.zeros((P,), np.int32)
        n = self.lib.goi_oracle_trace(C.byref(self.sc), self.st, _ptr(img), _ptr(color), _ptr(gau_sem), _ptr(num),
                                      _ptr(radii), self.threads)
        if n < 0:
            raise RuntimeError(f"oracle trace failed ({n})")
        return n, color, gau_sem, num

    def state(self) -> dict:
        """Intermediate forward state for stage-by-stage parity."""
        p, n, t = C.c_int(), C.c_int(), C.c_int()
        self.lib.goi_oracle_state_counts(self.st, C.byref(p), C.byref(n), C.byref(t))
        P, N, T = p.value, n.value, t.value
        HW = self.W * self.H
        s = dict(depths=np.zeros(P, np.float32), means2D=np.zeros((P, 2), np.float32),
                 conic_opacity=np.zeros((P, 4), np.float32), rgb=np.zeros((P, 3), np.float32),
                 cov3D=np.zeros((P, 6), np.float32), clamped=np.zeros((P, 3), np.uint8),
                 tiles_touched=np.zeros(P, np.uint32), point_list=np.zeros(N, np.uint32),
                 point_list_keys=np.zeros(N, np.uint64), ranges=np.zeros((T, 2), np.uint32),
                 n_contrib=np.zeros(HW, np.uint32))
        self.lib.goi_oracle_state_get(self.st, *[_ptr(s[k]) for k in (
            "depths", "means2D", "conic_opacity", "rgb", "cov3D", "clamped", "tiles_touched", "point_list",
            "point_list_keys", "ranges", "n_contrib")])
        s.update(P=P, N=N, T=T)
        return s


def mark_visible(means3D, viewmatrix, projmatrix) -> np.ndarray:
    m = _f32(means3D).reshape(-1, 3)
    v = _f32(viewmatrix).reshape(16)
    p = _f32(projmatrix).reshape(16)
    out = np.zeros(m.shape[0], np.uint8)
    _lib().goi_oracle_mark_visible(m.shape[0], _ptr(m), _ptr(v), _ptr(p), _ptr(out))
    return out.astype(bool)


def from_scene(scene, cam, bg=(0.0, 0.0, 0.0), **kw) -> Oracle:
    """Convenience: goi_hyperplane_amd.scene.GaussianScene + Camera -> Oracle."""
    args = dict(W=cam.image_width, H=cam.image_height, bg=np.asarray(bg, np.float32), means3D=scene.means3D,
                opacities=scene.opacities, semantics=scene.semantics, viewmatrix=cam.world_view_transform,
                projmatrix=cam.full_proj_transform, campos=cam.camera_center, tan_fovx=cam.tanfovx,
                tan_fovy=cam.tanfovy, shs=scene.shs, scales=scene.scales, rotations=scene.rotations,
                sh_degree=scene.sh_degree)
    args.update(kw)
    return Oracle(**args)


def knn_mean_dist2(points, fma: bool = True, brute: bool = False):
    """distCUDA2 oracle (oracle/knn_oracle.cpp): points [P,3] float32 -> mean squared distance to the
    3 nearest other points, [P] float32.  brute=True runs the O(P^2) definition instead of the
    restated Morton/box search."""
    path = os.path.join(_HERE, "libknn_oracle.so")
    if not os.path.exists(path):
        build(force=True)
    if "knn" not in _libs:
        lib = C.CDLL(path)
        for fn in (lib.goi_knn_oracle, lib.goi_knn_oracle_brute):
            fn.argtypes = [C.c_int, C.c_void_p, C.c_void_p, C.c_int]
            fn.restype = None
        _libs["knn"] = lib
    lib = _libs["knn"]
    pts = np.ascontiguousarray(points, dtype=np.float32).reshape(-1, 3)
    out = np.zeros(pts.shape[0], dtype=np.float32)
    fn = lib.goi_knn_oracle_brute if brute else lib.goi_knn_oracle
    fn(pts.shape[0], pts.ctypes.data, out.ctypes.data, 1 if fma else 0)
    return out


def adam_step(param, grad, exp_avg, exp_avg_sq, step, lr, beta1=0.9, beta2=0.999, eps=1e-8, nograd_rows=None):
    """numpy fp32 restatement of one torch.optim.Adam update as the reference runs it on the GPU
    (torch/optim/adam.py _multi_tensor_adam, non-capturable: _foreach_lerp_, _foreach_mul_ +
    _foreach_addcmul_, sqrt / bias_correction2_sqrt + eps, _foreach_addcdiv_), one rounding per op.
    `step` is the 1-based step count.  nograd_rows: optional bool [P]; masked rows see grad = 0
    (gui/main.py:480-513).  Returns (param, exp_avg, exp_avg_sq)."""
    f = np.float32
    p, g, m, v = (np.array(a, dtype=np.float32, copy=True) for a in (param, grad, exp_avg, exp_avg_sq))
    if nograd_rows is not None:
        g[np.asarray(nograd_rows, dtype=bool)] = 0
    m = m + (g - m) * f(1.0 - beta1)
    v = v * f(beta2) + (f(1.0 - beta2) * g) * g
    bc1 = 1.0 - beta1 ** step
    bc2_sqrt = (1.0 - beta2 ** step) ** 0.5
    den = np.sqrt(v) / f(bc2_sqrt) + f(eps)
    p = p + f(-(lr / bc1)) * (m / den)
    return p, m, v
