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
                distance: float = 5.0, target=(0.0, 0.0, 0.0)) -> Camera:
    """Camera on a sphere of radius ``distance`` around ``target`` (default: the origin), looking at it.
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
    t = np.array([0.0, 0.0, distance]) - R_w2c @ np.asarray(target, dtype=np.float64)  # x_cam = R (x - target) + (0, 0, d)
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
               extent=(2.0, 1.5, 1.0), log_scale_mean: float = -3.5, log_scale_std: float = 0.7,
               kind: str = "uniform") -> GaussianScene:
    """SURVEY.md 8(d) generator.  extent=(2,1.5,1) mu=-3.5 is BASELINE config 1;
    extent=(4,2.64,1) with the calibrated mu of ``HEADLINE`` is the 1M-Gaussian target.

    kind="clustered": the ADVERSARIAL workload (make_clustered_scene) -- what a reconstructed scene looks like to the
    rasterizer rather than a uniform box: clustered density, heavy-tailed anisotropic sizes, opaque foreground."""
    if kind == "clustered":
        return make_clustered_scene(P, S=S, sh_degree=sh_degree, seed=seed, extent=extent, log_scale_mean=log_scale_mean,
                                    log_scale_std=log_scale_std)
    if kind != "uniform":
        raise ValueError("kind must be 'uniform' or 'clustered'")

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
                                   log_scale_mean=log_scale_mean, log_scale_std=log_scale_std, kind="uniform"))


def make_clustered_scene(P: int, S: int = 16, sh_degree: int = 3, seed: int = 0, extent=(4.0, 2.64, 1.0),
                         log_scale_mean: float = -5.2, log_scale_std: float = 1.1, n_clusters: int = 48,
                         needle_frac: float = 0.02, giant_frac: float = 2e-5, occluder_frac: float = 0.12,
                         background_frac: float = 0.10, max_scale: float = 3.0, max_aspect: float = 300.0) -> GaussianScene:
    """A scene with the statistics of a RECONSTRUCTION (BASELINE configs 2 / 3 / 5 name MipNeRF360 scenes whose PLYs are not
    in this image) instead of SURVEY 8(d)'s uniform box -- everything the uniform scene is kind to is made hard here:

      * positions: a mixture of `n_clusters` anisotropic Gaussian blobs with heavy-tailed (Zipf-like) weights and log-normal
        sizes, over a thin uniform background -- some tiles see hundreds of times the Gaussians of others;
      * scales: log-normal with a wide sigma (1.1 instead of 0.7) per axis, so most Gaussians are anisotropic; `needle_frac`
        of them are needles (one axis x 25, the others x 0.25: tens to hundreds of pixels long, a pixel wide; aspect ratios
        are capped at 300 : 1 and the longest axis at 3 scene units) and
        `giant_frac` (at least 3) are frame-filling blobs (0.5 - 1.5 scene units): tile rectangles far beyond the 64-tile
        ellipse masks;
      * opacity: bimodal as in trained scenes -- a translucent mode (0.02 - 0.3; part of it below the 1/255 visibility floor
        after multiplication with the falloff) and an opaque mode (0.85 - 0.999: the 0.99 clamp is hit);
      * `occluder_frac` of the Gaussians form dense, opaque FOREGROUND sheets between the camera and everything else (the
        canonical camera sits at z = -5 looking down +z): lists behind them run thousands deep and almost none of it
        contributes -- the worst case for emit / tile sort, the best for the saturation front;
      * the rest as in make_scene (own RNG stream per array: a prefix of a scene is the smaller scene only for
        kind="uniform"; here the cluster assignment depends on P through the stream length, which is fine -- the scene is
        frozen by (P, seed)).
    tests/golden/calibration.json freezes the statistics of the two sizes the tests and the bench use."""
    def rng(i):
        return np.random.default_rng(seed + 1000 + i)

    ex = np.asarray(extent, dtype=np.float32)
    r0 = rng(0)
    # ---- which population a Gaussian belongs to
    u = r0.uniform(size=P)
    is_occ = u < occluder_frac
    is_bg = (u >= occluder_frac) & (u < occluder_frac + background_frac)
    # ---- clusters: Zipf-like weights, log-normal radii, anisotropic axes
    rc = rng(7)
    w = 1.0 / np.arange(1, n_clusters + 1) ** 0.9
    w /= w.sum()
    centres = rc.uniform(-1.0, 1.0, size=(n_clusters, 3)) * ex * np.array([0.95, 0.9, 1.0])
    radii = np.exp(rc.normal(-1.6, 0.7, size=(n_clusters, 1))) * np.exp(rc.normal(0.0, 0.5, size=(n_clusters, 3)))
    which = r0.choice(n_clusters, size=P, p=w)
    xyz = centres[which] + r0.normal(size=(P, 3)) * radii[which]
    # thin uniform background (a little beyond the box: some Gaussians fall outside the frustum)
    nb = int(is_bg.sum())
    xyz[is_bg] = r0.uniform(-1.15, 1.15, size=(nb, 3)) * ex
    # foreground sheets: three slightly curved, tilted layers at z = -2.6 .. -1.4 (camera at z = -5), each covering a part
    # of the frame
    no = int(is_occ.sum())
    sheet = r0.integers(0, 3, size=no)
    su, sv = r0.uniform(-1.0, 1.0, size=no), r0.uniform(-1.0, 1.0, size=no)
    cx = np.array([-0.9, 0.7, 0.1])[sheet]
    cy = np.array([0.3, -0.5, 0.6])[sheet]
    hw = np.array([0.9, 0.7, 1.3])[sheet]
    hh = np.array([0.8, 0.5, 0.3])[sheet]
    z0 = np.array([-2.6, -2.0, -1.4])[sheet]
    xo = cx + hw * su
    yo = cy + hh * sv
    zo = z0 + 0.25 * su * su - 0.15 * sv + 0.01 * r0.normal(size=no)
    xyz[is_occ] = np.stack([xo, yo, zo], axis=1)
    xyz = xyz.astype(np.float32)
    # ---- scales: wide log-normal, needles, giants; the occluders are small and flat (surfels)
    r1 = rng(1)
    log_s = r1.normal(log_scale_mean, log_scale_std, size=(P, 3))
    kind_u = r1.uniform(size=P)
    needle = kind_u < needle_frac
    axis = r1.integers(0, 3, size=P)
    boost = np.full((P, 3), np.log(0.25))
    boost[np.arange(P), axis] = np.log(25.0)
    log_s[needle] += boost[needle]
    n_giant = max(3, int(round(giant_frac * P)))
    giant_idx = r1.choice(P, size=min(n_giant, P), replace=False)
    log_s[giant_idx] = np.log(r1.uniform(0.5, 1.5, size=(len(giant_idx), 3)))
    occ_ls = r1.normal(log_scale_mean + 0.5, 0.35, size=(no, 3))
    occ_ls[:, 2] -= 1.5  # flat across the sheet
    keep_giant = np.zeros(P, bool)
    keep_giant[giant_idx] = True
    sel = is_occ & ~keep_giant & ~needle
    log_s[sel] = occ_ls[(np.cumsum(is_occ) - 1)[sel]]
    # what a trained scene can hold: the longest axis at most `max_scale` scene units, the aspect ratio at most `max_aspect`
    # (beyond ~1e3 : 1 the reference's own fp32 cov2D -> cov3D -> scale / rotation backward returns noise: two legal builds of
    # it disagree by the size of the gradient itself, and nothing can be pinned on such a Gaussian)
    log_s = np.minimum(log_s, np.log(max_scale))
    log_s = np.maximum(log_s, log_s.max(axis=1, keepdims=True) - np.log(max_aspect))
    scales = np.exp(log_s).astype(np.float32)
    # ---- rotations: random, except the occluders (aligned with their sheet up to a small tilt)
    q = rng(2).normal(size=(P, 4)).astype(np.float32)
    q[is_occ] = np.array([1.0, 0.0, 0.0, 0.0], np.float32) + 0.08 * q[is_occ]
    q /= np.linalg.norm(q, axis=1, keepdims=True)
    # ---- opacity: bimodal
    r3 = rng(3)
    opaque = r3.uniform(size=P) < 0.55
    opac = np.where(opaque, 0.85 + 0.149 * r3.uniform(size=P) ** 0.5, 0.02 + 0.28 * r3.uniform(size=P) ** 2.0)
    opac[is_occ] = 0.9 + 0.099 * r3.uniform(size=no)
    opac[giant_idx] = 0.05 + 0.3 * r3.uniform(size=len(giant_idx))  # (translucent haze: they touch every list without ending it)
    opac = opac.reshape(P, 1).astype(np.float32)
    M = 16
    shs = np.empty((P, M, 3), dtype=np.float32)
    shs[:, :1, :] = rng(4).normal(size=(P, 1, 3)).astype(np.float32)
    shs[:, 1:, :] = 0.2 * rng(5).normal(size=(P, M - 1, 3)).astype(np.float32)
    sem = rng(6).normal(size=(P, S)).astype(np.float32)
    return GaussianScene(xyz, scales, q.astype(np.float32), opac, shs, sem, sh_degree,
                         meta=dict(P=P, S=S, seed=seed, extent=tuple(float(e) for e in ex), log_scale_mean=log_scale_mean,
                                   log_scale_std=log_scale_std, kind="clustered", n_clusters=n_clusters,
                                   needle_frac=needle_frac, giants=int(len(giant_idx)), occluder_frac=occluder_frac))


# The headline workload of BASELINE.json: 1M Gaussians @1600x1056, RGB (SH deg 3) + 16-d feature.
# log_scale_mean is calibrated once (tests/golden/calibration.json) so that N/P is about 8.
HEADLINE = dict(P=1_000_000, S=16, W=1600, H=1056, extent=(4.0, 2.64, 1.0), log_scale_mean=-4.3,
                log_scale_std=0.7, fovx=1.0)


# The adversarial second workload: same size and image as the headline, the statistics of a
# reconstruction; and BASELINE config 5's shape -- a 512 x 512 close-up (gui/main_edit.py:551-553) of a >= 3 M scene.
CLUSTERED = dict(P=1_000_000, S=16, W=1600, H=1056, extent=(4.0, 2.64, 1.0), log_scale_mean=-5.2, log_scale_std=1.1, fovx=1.0)
CLOSEUP = dict(P=3_000_000, S=16, W=512, H=512, extent=(4.0, 2.64, 1.0), log_scale_mean=-4.9, log_scale_std=1.1, fovx=0.6,
               distance=3.2, yaw=0.3, pitch=-0.12)


ORBIT = dict(views=16, distance=3.0, target_radius=2.6, visit_stride=7)


def make_orbit_cameras(width: int, height: int, fovx: float = 1.0, n: int = ORBIT["views"], distance: float = ORBIT["distance"],
                       target_radius: float = ORBIT["target_radius"], visit_stride: int = ORBIT["visit_stride"]):
    """A capture orbit instead of one direction: ``n`` cameras whose look-at points travel once around the scene (an ellipse of
    radius ``target_radius`` in x, with a wobble in y and z) while the viewing direction turns with them, each ``distance`` from
    its target -- the way the training views of a reconstruction surround their scene (scene/dataset_readers.py).  Returned in
    VISIT order: step k renders orbit position (k * visit_stride) mod n, so consecutive steps look at different parts of the
    scene from different sides (train.py:118-124 pops its cameras at random).  On the headline box consecutive views share
    ~30 % of their visible Gaussians (bench.py measures and reports the figure on the device)."""
    cams = []
    for k in range(n):
        i = (k * visit_stride) % n
        th = 2.0 * math.pi * i / n
        target = (target_radius * math.cos(th), 0.6 * math.sin(2.0 * th), 0.3 * math.sin(th))
        cams.append(make_camera(width, height, fovx=fovx, yaw=th + 0.4 * math.sin(3.0 * th), pitch=0.15 * math.sin(2.0 * th + 1.0),
                                distance=distance, target=target))
    return cams


def make_workload(name: str, P: int | None = None, S: int | None = None, seed: int = 0):
    """(scene, canonical camera, spec) of a named workload: "headline" (SURVEY 8(d)'s uniform box), "clustered" (the
    adversarial scene at the headline's size and image) or "closeup" (BASELINE config 5's shape: a 512 x 512 close-up of a
    3 M clustered scene)."""
    spec = {"headline": HEADLINE, "clustered": CLUSTERED, "closeup": CLOSEUP}[name]
    kind = "uniform" if name == "headline" else "clustered"
    sc = make_scene(P or spec["P"], S or spec["S"], 3, seed, spec["extent"], spec["log_scale_mean"], spec["log_scale_std"], kind=kind)
    cam = make_camera(spec["W"], spec["H"], fovx=spec["fovx"], yaw=spec.get("yaw", 0.0), pitch=spec.get("pitch", 0.0),
                      distance=spec.get("distance", 5.0))
    return sc, cam, spec


def make_headline_scene(P: int | None = None, S: int | None = None, seed: int = 0) -> GaussianScene:
    h = HEADLINE
    return make_scene(P or h["P"], S or h["S"], 3, seed, h["extent"], h["log_scale_mean"], h["log_scale_std"])
