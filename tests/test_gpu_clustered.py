"""The ADVERSARIAL second workload: a scene with the statistics of a reconstruction -- clustered
density, heavy-tailed anisotropic sizes (needles, frame-filling blobs), bimodal opacity, opaque foreground sheets with tile
lists thousands deep behind them (goi_hyperplane_amd.scene.make_clustered_scene) -- against the CPU oracle:

  * at the headline's size and image (1 M Gaussians, 1600 x 1056, S = 16), dense random upstream gradients on all four
    outputs;
  * at BASELINE config 5's shape: a 512 x 512 close-up (gui/main_edit.py:551-553) of a 3 M scene, dense dL/dcolour, ZERO
    dL/dsemantics (the edit loop's guidance loss touches the image only).

Same gate as tests/test_gpu_parity.py::test_metric_configuration_matches_oracle: forward 1e-4 absolute outside the oracle's
fragile pixels, radii equal, gradients 1e-3 of each tensor's scale; a tensor that fails the plain gate passes only through
the noise-floor criterion of the fuzz test (the oracle's own two builds disagree there), recorded with the statistics.
Also checked on the way: the frame went through the default path (speculative forward, ellipse tile lists, member-mask
backward), nothing overflowed, and what the capacity policy / scratch sizes came to is recorded."""
import json
import os

import numpy as np
import pytest
import torch

from tests.test_gpu_parity import BWD_TOL, FWD_TOL, PARITY_STATS, dev  # noqa: F401  (dev: fixture)

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


@pytest.mark.parametrize("name,sem_grad", [("clustered", True), ("closeup", False)])
def test_clustered_workloads_match_oracle(oracle_mod, dev, name, sem_grad):  # noqa: F811
    from goi_hyperplane_amd import _C, _lib, rasterizer
    from goi_hyperplane_amd.render import GaussianSet, PipelineParams, TorchCamera, render
    from goi_hyperplane_amd.scene import make_workload
    from oracle import compare
    sc, cam, spec = make_workload(name)
    P, S, W, H = spec["P"], spec["S"], spec["W"], spec["H"]
    bg = np.array([0.0, 0.0, 0.0], np.float32) if name == "clustered" else np.array([1.0, 1.0, 1.0], np.float32)  # (edit loop: white)
    rng = np.random.default_rng(99)
    gc, gs, gd, ga = (rng.standard_normal((c, H, W)).astype(np.float32) / (W * H) for c in (3, S, 1, 1))
    if not sem_grad:
        gs, gd, ga = np.zeros_like(gs), np.zeros_like(gd), np.zeros_like(ga)

    pc = GaussianSet.from_scene(sc, dev)
    tcam, tbg, pipe = TorchCamera(cam, dev), torch.tensor(bg, device=dev), PipelineParams()
    spec0 = rasterizer.speculation_stats()
    for _ in range(3):
        render(tcam, pc, pipe, tbg)  # (exact frames: they teach the capacity policy)
    out = render(tcam, pc, pipe, tbg)
    n_lazy = rasterizer.last_num_rendered()
    if _C._FWD["mode"] == "speculative":
        assert isinstance(n_lazy, _C.LazyCount), "the workload must be checked on the default (speculative) forward"
    ups = [torch.tensor(g_, device=dev) for g_ in (gc, gs, gd, ga)]
    torch.autograd.backward((out["render"], out["semantics"], out["depth"], out["alpha"]), ups)
    res = {k: out[k].detach().cpu().numpy() for k in ("render", "semantics", "depth", "alpha", "radii")}
    g_hip = dict(means3D=pc._xyz.grad, opacity=pc._opacity.grad, semantics=pc._semantics.grad, sh=pc._features.grad,
                 scales=pc._scaling.grad, rotations=pc._rotation.grad, means2D=out["viewspace_points"].grad)
    g_hip = {k: v.detach().cpu().numpy() for k, v in g_hip.items()}
    n_hip = int(n_lazy)
    spec1 = rasterizer.speculation_stats()
    assert spec1["overflows"] == spec0["overflows"], "the frame overflowed its speculative capacity"
    capacity = getattr(n_lazy, "capacity", None)
    del out, pc
    torch.cuda.empty_cache()

    o = oracle_mod.from_scene(sc, cam, bg=bg, threads=os.cpu_count() or 1)
    f = o.forward()
    g_orc = o.backward(gc, gs, gd, ga)
    st = o.state()
    lens = (st["ranges"][:, 1].astype(np.int64) - st["ranges"][:, 0].astype(np.int64))
    tag = f"workload_{name}_P{P}_{W}x{H}_S{S}"
    assert 0 < n_hip <= f.num_rendered
    # the workload is what it claims to be: deep lists, big rectangles, a mostly opaque frame
    assert lens.max() > 3000 and f.num_rendered > 5 * H * W // 256, (int(lens.max()), int(f.num_rendered))
    assert int((st["tiles_touched"] > 64).sum()) > 1000  # rectangles beyond the 64-tile ellipse masks (TMASK_FULL)
    # needles: the reference's OWN fp32 evaluation of the pair exponent is ill-conditioned on much of this frame (its two legal
    # compilations differ by up to 0.04 in colour: oracle/goi_oracle.cpp header).  Every pixel that is not guard-fragile is
    # compared with the plain build AND with the variant that evaluates that one statement exactly, and must be within 1e-4
    # of one of them.
    nt = os.cpu_count() or 1
    ox = oracle_mod.from_scene(sc, cam, bg=bg, threads=nt, variant="f64power")
    fx = ox.forward()
    fw = compare.forward_stats(res, f, f_exact=fx)
    # gradients: within 1e-3 of ANY of three legal evaluations of the reference (plain, exact exponent, FMA-contracted);
    # Gaussians on which those disagree among themselves by more than 1e-3 (needles: the reference's fp32 cov2D -> cov3D ->
    # scale / rotation backward cancels catastrophically) are undecided: nothing is pinned on their geometry gradients
    # beyond "finite and not blown up", and there must be few of them
    o2 = oracle_mod.from_scene(sc, cam, bg=bg, threads=nt, variant="fma")
    o2.forward()
    builds = [g_orc, ox.backward(gc, gs, gd, ga), o2.backward(gc, gs, gd, ga)]
    bw = compare.backward_stats_arbitrated(g_hip, builds)
    PARITY_STATS[tag] = dict(bw, forward=fw, num_rendered_oracle=int(f.num_rendered), num_rendered_listed=n_hip,
                             capacity=capacity, longest_tile_list=int(lens.max()), mean_tile_list=float(lens.mean()),
                             visible=int((f.radii > 0).sum()), alpha_mean=float(f.alpha.mean()),
                             rectangles_over_64_tiles=int((st["tiles_touched"] > 64).sum()),
                             plain_gate=compare.backward_stats(g_hip, g_orc))
    assert fw["radii_equal"], f"{tag}: radii differ"
    assert fw["fragile_frac"] < 0.06, fw  # (what is compared with neither result: the guard-fragile pixels)
    for k in ("render", "semantics", "depth", "alpha"):
        assert fw[k]["max"] < FWD_TOL, f"{tag}: {k} {fw[k]}"
    if not sem_grad:
        assert float(np.abs(g_hip["semantics"]).max()) == 0.0  # zero dL/dsemantics upstream: exactly zero downstream
    assert bw["undecided_rows"] <= 2e-3 * P, bw["undecided_rows"]
    for k, s_ in bw.items():
        if not isinstance(s_, dict):
            continue
        assert s_["finite"] and s_["max"] < BWD_TOL and s_["n_over"] == 0 and s_["undecided_blown_up"] == 0, f"{tag}: grad {k} {s_}"
        assert s_["p9999"] < 2e-4, f"{tag}: grad {k} {s_}"
