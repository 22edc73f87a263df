#!/usr/bin/env python3
"""How evenly does the band mapping (blend_common.h quad_slot: XCD x owns the x-th contiguous eighth of the quadrants)
spread the blend work over the 8 XCDs?  Cost proxy of a quadrant = the list positions its wave walks = max n_contrib of
its 64 pixels (what render_fwd_k leaves in qcost[]).  Prints per-band sums and a list-scheduling estimate of each
XCD's finish time (512 wave slots per XCD for the backward: 32 CUs x 16)."""
import heapq
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from goi_hyperplane_amd import _C  # noqa: E402
from goi_hyperplane_amd.render import GaussianSet, TorchCamera  # noqa: E402
from goi_hyperplane_amd.scene import HEADLINE, make_camera, make_headline_scene  # noqa: E402
from goi_hyperplane_amd.render import PipelineParams  # noqa: E402,F401

h = HEADLINE
dev = torch.device("cuda:0")
sc = make_headline_scene()
if "--skew" in sys.argv:  # a dense object low in the frame over a sparse background: what captured scenes look like
    u = (sc.means3D[:, 1] / h["extent"][1] + 1.0) * 0.5
    sc.means3D[:, 1] = (2.0 * u ** 2.5 - 1.0) * h["extent"][1]
    v = sc.means3D[:, 0] / h["extent"][0]
    sc.means3D[:, 0] = np.sign(v) * np.abs(v) ** 1.8 * h["extent"][0]
pc = GaussianSet.from_scene(sc, dev)
_C.set_forward_mode(speculative=False)
for yaw in (0.0, -0.1, 0.12):
    cam = make_camera(h["W"], h["H"], fovx=h["fovx"], yaw=yaw)
    tcam = TorchCamera(cam, dev)
    rs_args = (torch.zeros(3, device=dev), pc._xyz.detach(), torch.Tensor([]), pc._semantics.detach(),
               pc._opacity.detach(), pc._scaling.detach(), pc._rotation.detach(), 1.0, torch.Tensor([]),
               tcam.world_view_transform, tcam.full_proj_transform, cam.tanfovx, cam.tanfovy, cam.image_height,
               cam.image_width, pc._features.detach(), sc.sh_degree, tcam.camera_center, False, False)
    n_r, *_rest, geom, binning, img = _C.rasterize_gaussians(*rs_args)
    views = _C.debug_views(sc.P, cam.image_width, cam.image_height, n_r, geom, binning, img)
    nc = views["n_contrib"].cpu().numpy().reshape(h["H"], h["W"]).astype(np.int64)
    gx, gy = (h["W"] + 15) // 16, (h["H"] + 15) // 16
    pad = np.zeros((gy * 16, gx * 16), np.int64)
    pad[:h["H"], :h["W"]] = nc
    q = pad.reshape(gy, 2, 8, gx, 2, 8).max(axis=(2, 5))  # [ty, qy, tx, qx]
    cost = q.transpose(0, 2, 1, 3).reshape(-1)  # quadrant index = tile * 4 + (qy * 2 + qx), tiles row-major
    n = cost.size
    per = (n + 7) // 8
    sums = np.array([cost[b * per:(b + 1) * per].sum() for b in range(8)], np.float64)

    def finish(costs, slots=512, lpt=True):
        c = sorted(costs, reverse=True) if lpt else list(costs)
        heap = [0.0] * slots
        for x in c:
            heapq.heapreplace(heap, heap[0] + x + 20.0)  # +20: a wave's fixed cost in list positions
        return max(heap)

    fin = np.array([finish(cost[b * per:(b + 1) * per]) for b in range(8)])
    allf = finish(cost, slots=4096)
    print(f"yaw {yaw:+.2f}: band sums / mean = {np.round(sums / sums.mean(), 3).tolist()}  max/mean = {sums.max() / sums.mean():.3f}")
    print(f"          LPT finish per XCD / ideal-global = {np.round(fin / allf, 3).tolist()}  -> kernel ends at {fin.max() / allf:.3f} of a perfectly balanced launch")
