"""goi_raster_blend_stats (csrc/blend_stats.hip): the lane-utilisation counters against a brute-force numpy replay of the
blend (CR/forward.cu:330-372) on the workspaces of the same frame."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _replay(v, W, H):
    """(pixel, Gaussian) contributions, member pairs per 8x8 quadrant / 4x4 / 2x2 block, from the frame's own lists"""
    gx, gy = (W + 15) // 16, (H + 15) // 16
    m2, co, pl, ranges = v["means2D"].astype(np.float64), v["conic_opacity"].astype(np.float64), v["point_list"], v["ranges"]
    live = mem88 = mem44 = mem22 = 0
    for t in range(gx * gy):
        r0, r1 = int(ranges[t][0]), int(ranges[t][1])
        if r1 <= r0:
            continue
        ids = pl[r0:r1]
        tx, ty = t % gx, t // gx
        px = tx * 16 + np.arange(16)[None, :].repeat(16, 0)
        py = ty * 16 + np.arange(16)[:, None].repeat(16, 1)
        inside = (px < W) & (py < H)
        dx = m2[ids, 0][:, None, None] - px[None]
        dy = m2[ids, 1][:, None, None] - py[None]
        a, b, c, o = (co[ids, i][:, None, None] for i in range(4))
        power = -0.5 * (a * dx * dx + c * dy * dy) - b * dx * dy
        alpha = np.minimum(0.99, o * np.exp(np.minimum(power, 0)))
        valid = (power <= 0) & (alpha >= 1 / 255) & inside[None]
        cp = np.cumprod(np.where(valid, 1 - alpha, 1.0), axis=0)
        stopper = valid & (cp < 1e-4)
        stop = np.where(stopper.any(0), stopper.argmax(0), len(ids))
        contrib = valid & (np.arange(len(ids))[:, None, None] < stop[None])
        live += int(contrib.sum())
        c4 = contrib.reshape(len(ids), 2, 8, 2, 8)
        mem88 += int(c4.any(axis=(2, 4)).sum())
        mem44 += int(contrib.reshape(len(ids), 4, 4, 4, 4).any(axis=(2, 4)).sum())
        mem22 += int(contrib.reshape(len(ids), 8, 2, 8, 2).any(axis=(2, 4)).sum())
    return live, mem88, mem44, mem22


@pytest.mark.parametrize("P,W,H,mu", [(3000, 123, 77, -2.6), (800, 64, 48, -1.2), (20000, 400, 300, -3.5)])
def test_blend_stats_match_a_numpy_replay(P, W, H, mu):
    from goi_hyperplane_amd import _C
    from goi_hyperplane_amd.render import GaussianSet, TorchCamera
    from goi_hyperplane_amd.scene import make_camera, make_scene
    dev = torch.device("cuda:0")
    sc = make_scene(P, S=16, sh_degree=2, seed=3, log_scale_mean=mu)
    cam = make_camera(W, H, yaw=0.2, pitch=-0.1)
    pc, tcam = GaussianSet.from_scene(sc, dev), TorchCamera(cam, dev)
    args = (torch.zeros(3, device=dev), pc._xyz.detach(), torch.Tensor([]), pc._semantics.detach(), pc._opacity.detach(),
            pc._scaling.detach(), pc._rotation.detach(), 1.0, torch.Tensor([]), tcam.world_view_transform,
            tcam.full_proj_transform, cam.tanfovx, cam.tanfovy, H, W, pc._features.detach(), sc.sh_degree, tcam.camera_center,
            False, False)
    n, *_rest, geom, binning, img = _C.rasterize_gaussians(*args)
    st = _C.blend_stats(P, W, H, n, geom, binning, img)
    views = {k: x.cpu().numpy() for k, x in _C.debug_views(P, W, H, n, geom, binning, img).items()}
    live, mem88, mem44, mem22 = _replay(views, W, H)
    assert st["dead_member_pairs"] == 0
    assert st["pixels"] == W * H
    assert st["sum_n_contrib"] == int(views["n_contrib"].astype(np.int64).sum())
    # the replay evaluates alpha in the reference's direct form in float64, the kernels through the folded polynomial in fp32: a
    # pair sitting on a guard can fall either way (the parity tests' "fragile" pixels) -- a fraction of a per cent
    for got, want in ((st["live_lanes"], live), (st["member_pairs"], mem88), (st["pairs_4x4"], mem44), (st["pairs_2x2"], mem22)):
        assert abs(got - want) <= 0.01 * want + 2, (got, want)
    assert st["member_pairs"] <= st["forward_pairs"] <= st["positions"]
    assert st["member_pairs"] <= st["pairs_8x4"] <= st["pairs_4x4"] <= st["pairs_2x2"] <= st["live_lanes"]
    assert 0.0 < st["lane_utilisation_backward"] <= 1.0
