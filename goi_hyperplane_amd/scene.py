"""Synthetic Gaussian scenes and cameras for tests and bench (numpy only, seeded).

The camera conventions restate the reference's (no code is imported from it):
  * ``world_view_transform = W2C.T`` and ``full_proj_transform = world_view_transform @ P.T``
    (scene/cameras.py:45-48), i.e. matrices are stored transposed / row-vector style;
  * ``getProjectionMatrix`` (utils/graphics_utils.py:51-71), znear 0.01 / zfar 100
    (scene/cameras.py:39-40);
  * ``camera_center = inverse(world_view_transform)[3, :3]`` (scene/cameras.py:48).
tests/test_oracle_pins.py checks these restatements against golden matrices produced by the
reference's own functions (tests/golden/make_golden.py).

The Gaussian generator is the one SURVEY.md section 8(d) specifies: every array comes from its own
RNG stream (seed + array index) so P, V and N do not depend on S.
"""
from __future__ import annotations

import math
from dataclasses import dataclass, field

import numpy as np

ZNEAR = 0.01
ZFAR = 100.0


def projection_matrix(znear: float, zfar: float, fovx: float, fovy: float) -> np.ndarray:
    """Row-major 4x4 perspective matrix, restating utils/graphics_utils.py:51-71."""
    tan_half_y = math.tan(fovy / 2)
    tan_half_x = math.tan(fovx / 2)
    top = tan_half_y * znear
    bottom = -top
    right = tan_half_x * znear
    left = -right
    P = np.zeros((4, 4), dtype=np.float32)
    P[0, 0] = 2.0 * znear / (right - left)
    P[1, 1] = 2.0 * znear / (top - bottom)
    P[0, 2] = (right + left) / (right - left)
    P[1, 2] = (top + bottom) / (top - bottom)
    P[3, 2] = 1.0
    P[2, 2] = zfar / (zfar - znear)
    P[2, 3] = -(zfar * znear) / (zfar - znear)
    return P


def world_to_view(R: np.ndarray, t: np.ndarray) -> np.ndarray:
    """W2C 4x4 from a camera rotation ``R`` (stored transposed, as COLMAP readers do) and ``t``;
    restates utils/graphics_utils.py:38-49 with translate=0, scale=1."""
    Rt = np.zeros((4, 4), dtype=np.float64)
    Rt[:3, :3] = R.transpose()
    Rt[:3, 3] = t
    Rt[3, 3] = 1.0
    return Rt.astype(np.float32)


@dataclass
class Camera:
    """The per-view constants the rasterizer settings need (scene/cameras.py:17-48)."""

    image_width: int
    image_height: int
    FoVx: float
    FoVy: float
    world_view_transform: np.ndarray  # [4,4] = W2C.T
    full_proj_transform: np.ndarray  # [4,4]
    camera_center: np.ndarray  # [3]

    @property
    def tanfovx(self) -> float:
        return math.tan(self.FoVx * 0.5)

    @property
    def tanfovy(self) -> float:
        return math.tan(self.FoVy * 0.5)


def make_camera(width: int, height: int, fovx: float = 1.0, yaw: float = 0.0, pitch: float = 0.0,
                distance: float = 5.0) -> Camera:
    """Camera on a sphere of radius ``distance`` around the origin, looking at the origin.
    yaw = pitch = 0 is SURVEY 8(d)'s canonical view: camera at (0,0,-5) looking down +z, i.e.
    W2C = [I | (0,0,5)]."""
    tanfovx = math.tan(0.5 * fovx)
    tanfovy = tanfovx * height / width
    fovy = 2.0 * math.atan(tanfovy)
    cy, sy = math.cos(yaw), math.sin(yaw)
    cp, sp = math.cos(pitch), math.sin(pitch)
    Ry = np.array([[cy, 0, sy], [0, 1, 0], [-sy, 0, cy]], dtype=np.float64)
    Rx = np.array([[1, 0, 0], [0, cp, -sp], [0, sp, cp]], dtype=np.float64)
    R_w2c = Rx @ Ry  # world -> camera rotation
    t = np.array([0.0, 0.0, distance])
    w2c = world_to_view(R_w2c.T, t)  # world_to_view transposes its argument back
    view_T = np.ascontiguousarray(w2c.T)
    proj_T = np.ascontiguousarray(projection_matrix(ZNEAR, ZFAR, fovx, fovy).T)
    full = (view_T.astype(np.float32) @ proj_T.astype(np.float32)).astype(np.float32)
    center = np.linalg.inv(view_T.astype(np.float64))[3, :3].astype(np.float32)
    return Camera(width, height, fovx, fovy, view_T.astype(np.float32), full, center)


@dataclass
class GaussianScene:
    """Inputs of one rasterizer call, already activated the way scene/gaussian_model.py's getters
    hand them over (exp'd scales, normalised rotations, sigmoid'ed opacity)."""

    means3D: np.ndarray  # [P,3]
    scales: np.ndarray  # [P,3]
    rotations: np.ndarray  # [P,4]  (r,x,y,z), unit norm
    opacities: np.ndarray  # [P,1]
    shs: np.ndarray  # [P,M,3]
    semantics: np.ndarray  # [P,S]
    sh_degree: int = 3
    meta: dict = field(default_factory=dict)

    @property
    def P(self) -> int:
        return int(self.means3D.shape[0])

    @property
    def S(self) -> int:
        return int(self.semantics.shape[1])


def make_scene(P: int, S: int = 16, sh_degree: int = 3, seed: int = 0,
               extent=(2.0, 1.5, 1.0), log_scale_mean: float = -3.5, log_scale_std: float = 0.7) -> GaussianScene:
    """SURVEY.md 8(d) generator.  extent=(2,1.5,1) mu=-3.5 is BASELINE config 1;
    extent=(4,2.64,1) with the calibrated mu of ``HEADLINE`` is the 1M-Gaussian target."""
    def rng(i):
        return np.random.default_rng(seed + i)

    ex = np.asarray(extent, dtype=np.float32)
    xyz = (rng(0).uniform(-1.0, 1.0, size=(P, 3)).astype(np.float32)) * ex
    scales = np.exp(rng(1).normal(log_scale_mean, log_scale_std, size=(P, 3))).astype(np.float32)
    q = rng(2).normal(size=(P, 4)).astype(np.float32)
    q /= np.linalg.norm(q, axis=1, keepdims=True)
    opac = (0.1 + 0.8 * np.abs(rng(3).uniform(-1.0, 1.0, size=(P, 1)))).astype(np.float32)
    M = 16
    shs = np.empty((P, M, 3), dtype=np.float32)
    shs[:, :1, :] = rng(4).normal(size=(P, 1, 3)).astype(np.float32)
    shs[:, 1:, :] = 0.2 * rng(5).normal(size=(P, M - 1, 3)).astype(np.float32)
    sem = rng(6).normal(size=(P, S)).astype(np.float32)
    return GaussianScene(xyz, scales, q.astype(np.float32), opac, shs, sem, sh_degree,
                         meta=dict(P=P, S=S, seed=seed, extent=tuple(float(e) for e in ex),
                                   log_scale_mean=log_scale_mean, log_scale_std=log_scale_std))


# The headline workload of BASELINE.json: 1M Gaussians @1600x1056, RGB (SH deg 3) + 16-d feature.
# log_scale_mean is calibrated once (tests/golden/calibration.json) so that N/P is about 8.
HEADLINE = dict(P=1_000_000, S=16, W=1600, H=1056, extent=(4.0, 2.64, 1.0), log_scale_mean=-4.3,
                log_scale_std=0.7, fovx=1.0)


def make_headline_scene(P: int | None = None, S: int | None = None, seed: int = 0) -> GaussianScene:
    h = HEADLINE
    return make_scene(P or h["P"], S or h["S"], 3, seed, h["extent"], h["log_scale_mean"], h["log_scale_std"])
