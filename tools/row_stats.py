#!/usr/bin/env python3
"""How many partial-gradient rows does the backward write, and how many would a per-TILE combine write?
Reads the validity bytes of the backward scratch after one headline backward (ctypes binding, exact forward so that the
scratch is laid out for num_rendered):  rows = set bytes; per-tile rows = instances with at least one set byte."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from goi_hyperplane_amd import _C, _lib, rasterizer  # noqa: E402
from goi_hyperplane_amd.render import GaussianSet, PipelineParams, TorchCamera, render  # noqa: E402
from goi_hyperplane_amd.scene import HEADLINE, make_camera, make_headline_scene  # noqa: E402

dev = torch.device("cuda:0")
_C.set_binding("ctypes")
_C.set_forward_mode(speculative=False)
pc = GaussianSet.from_scene(make_headline_scene(), dev)
cam = TorchCamera(make_camera(HEADLINE["W"], HEADLINE["H"], fovx=HEADLINE["fovx"]), dev)
out = render(cam, pc, PipelineParams(), torch.zeros(3, device=dev))
n = int(rasterizer.last_num_rendered())
(out["render"].sum() + out["semantics"].sum()).backward()
torch.cuda.synchronize()
buf = next(iter(_C._SCRATCH.values()))
S = HEADLINE["S"]
row_bytes = 4 * ((4 * ((S + 3) // 4) + 4 + 6 + 15) // 16 * 16)
base = (buf.data_ptr() + 255) // 256 * 256
flags_at = (base + 4 * n * row_bytes + 255) // 256 * 256 - buf.data_ptr()
flags = buf[flags_at:flags_at + 4 * n].view(n, 4) != 0
rows = int(flags.sum())
inst = int(flags.any(1).sum())
print(f"listed instances N = {n}; (instance, quadrant) rows written = {rows} ({rows * row_bytes / 1e6:.0f} MB at "
      f"{row_bytes} B); instances with at least one row = {inst} ({inst * row_bytes / 1e6:.0f} MB if the four quadrants "
      f"of a tile were combined before HBM); quadrants per contributing instance = {rows / max(inst, 1):.2f}; "
      f"instances that contribute nothing = {1 - inst / n:.1%}")
