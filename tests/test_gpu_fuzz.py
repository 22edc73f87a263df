"""Seeded sweep of small random configurations (sizes, channel counts, SH degree, scale of the Gaussians,
camera pose, background) through the same checks as tests/test_gpu_parity.py: integer stages bit-exact
against the oracle's lists (cull_variant 0), images within 1e-4 and gradients within 1e-3 with the default
(culled) lists.  Meant to catch edge cases no hand-picked case hits: ragged tiles, lists longer or shorter
than a staging batch, Gaussians larger than the image, empty tiles, S not a multiple of 4."""
import numpy as np
import pytest

from goi_hyperplane_amd.scene import make_camera, make_scene
from tests.golden.make_golden import upstream_grads
from tests.test_gpu_parity import check_backward, check_forward, dev, run_hip  # noqa: F401  (dev is a fixture)

pytestmark = pytest.mark.gpu


def configs(n=None, seed=None):
    # GOI_FUZZ_N / GOI_FUZZ_SEED widen the sweep for a soak run (default: the 60 committed configurations)
    import os
    n = int(os.environ.get("GOI_FUZZ_N", 60)) if n is None else n
    seed = int(os.environ.get("GOI_FUZZ_SEED", 2024)) if seed is None else seed
    rng = np.random.default_rng(seed)
    out = []
    for k in range(n):
        P = int(rng.choice([1, 2, 7, 63, 64, 65, 255, 257, 600, 1500, 3000]))
        W = int(rng.integers(17, 220))
        H = int(rng.integers(17, 160))
        S = int(rng.choice([1, 2, 3, 4, 7, 10, 16, 17, 24, 32]))
        deg = int(rng.integers(0, 4))
        mu = float(rng.uniform(-4.2, -0.8))
        yaw, pitch = float(rng.uniform(-0.5, 0.5)), float(rng.uniform(-0.3, 0.3))
        out.append((k, P, S, W, H, mu, deg, yaw, pitch))
    return out


@pytest.mark.parametrize("k,P,S,W,H,mu,deg,yaw,pitch", configs())
def test_random_configuration(oracle_mod, dev, k, P, S, W, H, mu, deg, yaw, pitch):  # noqa: F811
    check_configuration(oracle_mod, dev, k, P, S, W, H, mu, deg, yaw, pitch)


def check_configuration(oracle_mod, dev, k, P, S, W, H, mu, deg, yaw, pitch):  # noqa: F811
    sc = make_scene(P, S=S, sh_degree=deg, seed=100 + k, log_scale_mean=mu)
    cam = make_camera(W, H, yaw=yaw, pitch=pitch)
    bg = np.random.default_rng(k).random(3).astype(np.float32)
    grads = upstream_grads(S, H, W, seed=k)
    o = oracle_mod.from_scene(sc, cam, bg=bg)
    f = o.forward()
    res = run_hip(sc, cam, bg, dev, grads=grads, debug_views=True)
    tag = f"fuzz{k}_P{P}_S{S}_{W}x{H}_mu{mu:.2f}_deg{deg}"
    st = o.state()
    assert res["N"] == f.num_rendered, tag
    v = res["views"]
    assert (v["tiles_touched"].astype(np.uint32) == st["tiles_touched"]).all(), tag
    assert (v["ranges"].astype(np.uint32) == st["ranges"]).all(), tag
    assert (v["point_list"].astype(np.uint32) == st["point_list"]).all(), tag
    ok = f.fragile.reshape(-1) == 0
    if ok.mean() > 0.98:
        assert (v["n_contrib"].astype(np.uint32)[ok] == st["n_contrib"][ok]).all(), tag
        try:
            check_forward(res, f, tag)
        except AssertionError:
            # the forward gate is 1e-4 ABSOLUTE on every map; a draw on which it fails outside the fragile pixels is accepted
            # only if the reference's other fp32 ordering (the FMA-contracted twin) is within 1e-4 of this build there
            # (seen once in 8000 soaked configurations: a depth map at depths of 5-10, 1.06e-4)
            o2f = oracle_mod.from_scene(sc, cam, bg=bg, variant="fma")
            check_forward(res, f, tag + "_either_build", twin=o2f.forward())
        try:
            check_backward(res["grads"], o.backward(*grads), tag)
        except AssertionError:
            # A guard (alpha < 1/255, T < 1e-4, power > 0, a clamp) is a step function: two correct builds one ulp
            # apart can take different sides for one (pixel, Gaussian) pair, and an ill-conditioned Gaussian amplifies
            # that into > 1e-3 of a gradient tensor's scale.  The FMA-contracted twin of the oracle is the committed
            # measure of that noise floor: agreeing with EITHER build is agreement.
            o2 = oracle_mod.from_scene(sc, cam, bg=bg, variant="fma")
            o2.forward()
            try:
                check_backward(res["grads"], o2.backward(*grads), tag + "_vs_fma_twin")
            except AssertionError:
                # Neither build: then a pair may have taken the other side of a guard in THIS implementation (its
                # alpha differs from the oracle's by up to ~1e-5 relative: polynomial evaluation, v_exp_f32).  That
                # shows in the forward as an output jump of ~1/255 of the pair's weight at a pixel the oracle itself
                # marks as fragile (a pair within 1e-4 of a guard).  Only then, and only if every such pixel is fragile, the
                # gradients are compared at 1e-2: a flipped far-tail pair of a needle-shaped Gaussian carries a large
                # gradient although its alpha is 1/255.
                # (the jump is looked for in every output: on a saturated pixel a flipped pair moves alpha = 1 - T by
                # 1/255 of T_final ~ 1e-4 only, but colour and features by 1/255 of the pair's own weight)
                jump = np.zeros(W * H, bool)
                for key, ref in (("render", f.color), ("semantics", f.semantic), ("depth", f.depth), ("alpha", f.alpha)):
                    dd = np.abs(np.asarray(res[key]).reshape(ref.shape[0], -1) - ref.reshape(ref.shape[0], -1))
                    jump |= (dd > 2e-4).any(axis=0)
                flipped = jump
                if flipped.any():
                    assert not (flipped & ok).any(), tag + ": an output jump at a pixel that is not fragile"
                    check_backward(res["grads"], o.backward(*grads), tag + "_guard_flip", tol=1e-2)
                else:
                    # no flip: plain rounding noise, amplified by an ill-conditioned Gaussian (a needle several scene
                    # units long: its cov2D -> cov3D -> rotation chain cancels).  The two oracle builds measure that
                    # amplification: the tolerance is widened to three times their own disagreement, never beyond 1e-2.
                    ga, gb = o.backward(*grads), o2.backward(*grads)
                    floor = max(float(np.abs(np.asarray(ga[n]) - np.asarray(gb[n])).max() / (np.abs(ga[n]).max() + 1e-20))
                                for n in ("means3D", "opacity", "semantics", "sh", "scales", "rotations"))
                    assert floor > 1e-4, tag + ": gradients differ although the oracle builds agree"
                    check_backward(res["grads"], ga, tag + f"_noise_floor_{floor:.1e}", tol=min(1e-2, 3 * floor))
    else:  # pathological draw (e.g. one huge Gaussian grazing every guard): only the exact stages are meaningful
        assert (res["radii"] == f.radii).all(), tag


# Configurations a soak run found where this build ALONE was outside the gradient tolerance (the oracle agreed with
# float64 to 1e-5): kept as strict cases -- plain 1e-3 against the oracle, no fallback criteria.
#   (k, n, seed) = (201, 1500, 8102): ONE Gaussian; its opacity gradient is a sum over pixels that cancels to 1/600 of
#   its terms, and the two-plane split-bf16 flush (products to ~1e-5) left 2.1e-3 there; the moments now get a third
#   plane.
SOAK_REGRESSIONS = [(201, 1500, 8102)]


@pytest.mark.gpu
@pytest.mark.parametrize("k,n,seed", SOAK_REGRESSIONS)
def test_soak_regression_is_within_the_plain_tolerance(oracle_mod, dev, k, n, seed):  # noqa: F811
    cfg = [c for c in configs(n, seed) if c[0] == k][0]
    _k, P, S, W, H, mu, deg, yaw, pitch = cfg
    sc = make_scene(P, S=S, sh_degree=deg, seed=100 + k, log_scale_mean=mu)
    cam = make_camera(W, H, yaw=yaw, pitch=pitch)
    bg = np.random.default_rng(k).random(3).astype(np.float32)
    grads = upstream_grads(S, H, W, seed=k)
    o = oracle_mod.from_scene(sc, cam, bg=bg)
    f = o.forward()
    res = run_hip(sc, cam, bg, dev, grads=grads)
    tag = f"soak{seed}_{k}_P{P}_S{S}_{W}x{H}"
    check_forward(res, f, tag)
    check_backward(res["grads"], o.backward(*grads), tag)


# The KNOWN OUTLIER of the round-5 soak (profiles/r05_soaks.txt): configuration 1185 of GOI_FUZZ_N=1500 GOI_FUZZ_SEED=66001
# (P = 257, S = 17, 173 x 157) is outside EVERY criterion of check_configuration: ONE of its 1 028 rotation-gradient elements is
# 3.11e-3 of the tensor's scale from the oracle where the oracle's own plain and FMA-contracted builds differ by 9.1e-4 (three
# times that, 2.74e-3, is what the sweep allows).  Float64 yardstick (tools/soak_f64.py 1500 66001 2e-3): |hip - f64| 1.28e-2,
# |oracle - f64| 9.67e-3 -- the reference's own fp32 order is 1e-2 from float64 on that element (an ill-conditioned needle,
# DESIGN.md section 2), the oracle is the closer one, and the exact-fp32 flush (bwd_variant 2) gives 2.48e-3.  1 of 2 701 soaked
# configurations in round 5 (rounds 1-4: 4 of 18 200).  The criteria are not widened for it: it sits here as a strict xfail so
# that the outlier is visible in every `-m gpu` run -- and so that a build on which it starts to pass says so.
KNOWN_OUTLIERS = [(1185, 1500, 66001)]


@pytest.mark.gpu
@pytest.mark.xfail(strict=True, reason="known ill-conditioned-needle outlier of the round-5 soak: 3.1e-3 on one dL/drotation "
                                       "element (tolerance 2.74e-3 = 3 x the oracle builds' own spread); float64 figures above")
@pytest.mark.parametrize("k,n,seed", KNOWN_OUTLIERS)
def test_known_soak_outlier_stays_visible(oracle_mod, dev, k, n, seed):  # noqa: F811
    cfg = [c for c in configs(n, seed) if c[0] == k][0]
    check_configuration(oracle_mod, dev, *cfg)
