#!/usr/bin/env python3
"""How often does the speculative forward overflow under camera motion?  (reports only)

Three frame sequences over one scene, default policy (first three frames exact, capacity = 2 x the largest count seen):
  random : every frame an independent camera -- distance 2.5 .. 9 (num_rendered varies ~10x), yaw/pitch anywhere in +-0.6/0.3:
           the worst case for a high-water policy is the first few frames, before a close-up has been seen;
  path   : a smooth fly-through (what a GUI or a video render produces), zooming in from far to near;
  scale  : the random cameras plus the GUI's scaling-modifier slider (gui/main.py) thrown around between 0.4 and 3 per frame
           (num_rendered grows with its square): the harshest thing an interactive user can do.
Prints frames, overflows, waits and the num_rendered range; every overflow is also checked: the count read -> redo -> the
frame equals the exact one."""
import math
import os
import sys
import warnings

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from goi_hyperplane_amd import _C, rasterizer  # noqa: E402
from goi_hyperplane_amd.render import GaussianSet, PipelineParams, TorchCamera, render  # noqa: E402
from goi_hyperplane_amd.scene import make_camera, make_scene  # noqa: E402

dev = torch.device("cuda:0")
P = int(sys.argv[1]) if len(sys.argv) > 1 else 300_000
n_frames = int(sys.argv[2]) if len(sys.argv) > 2 else 600
pc = GaussianSet.from_scene(make_scene(P, S=16, seed=3, extent=(4.0, 2.64, 1.0), log_scale_mean=-3.9), dev)
bg = torch.zeros(3, device=dev)
rng = np.random.default_rng(0)
seqs = {
    "random": [dict(distance=float(rng.uniform(2.5, 9.0)), yaw=float(rng.uniform(-0.6, 0.6)), pitch=float(rng.uniform(-0.3, 0.3)))
               for _ in range(n_frames)],
    "path": [dict(distance=9.0 - 6.5 * (0.5 - 0.5 * math.cos(math.pi * i / n_frames)), yaw=0.5 * math.sin(6.28 * i / n_frames),
                  pitch=0.2 * math.sin(3.1 * i / n_frames)) for i in range(n_frames)],
}
seqs["scale"] = [dict(c, scale=float(rng.uniform(0.4, 3.0))) for c in seqs["random"]]
warnings.simplefilter("ignore", _C.RasterOverflowWarning)
for name, cams in seqs.items():
    _C.poll_counts(wait=True)
    _C._SPEC.clear()
    _C.set_forward_mode(speculative=True, capacity=None, inference_speculative=True)  # (image-only frames are exact by default)
    s0 = dict(_C.SPECULATION_STATS)
    counts = []
    with torch.no_grad():
        for c in cams:
            c = dict(c)
            scale = c.pop("scale", 1.0)
            out = render(TorchCamera(make_camera(800, 528, **c), dev), pc, PipelineParams(), bg, scaling_modifier=scale)
            counts.append(rasterizer.last_num_rendered())
    torch.cuda.synchronize()
    _C.poll_counts(wait=True)
    s1 = _C.SPECULATION_STATS
    # (the counts were never read in time: overflowed frames were found by the polls, as a fire-and-forget caller would)
    ns = np.array([c._n if isinstance(c, _C.LazyCount) else int(c) for c in counts])
    over = [i for i, c in enumerate(counts) if getattr(c, "overflowed", False)]
    print(f"{name:6s}: frames {len(cams)}, exact {s1['exact_frames'] - s0['exact_frames']}, speculative "
          f"{s1['speculative_frames'] - s0['speculative_frames']}, OVERFLOWS {s1['overflows'] - s0['overflows']} (frames {over[:12]}), "
          f"waits {s1['waits'] - s0['waits']}; num_rendered {ns.min()} .. {ns.max()} (x{ns.max() / max(ns.min(), 1):.1f}), largest "
          f"frame-to-frame growth x{(ns[1:] / np.maximum(ns[:-1], 1)).max():.2f}")
# an overflow, read before use: redone == exact
_C._SPEC.clear()
_C.set_forward_mode(speculative=True, capacity=None, inference_speculative=True, min_history=1)  # (speculate from frame 2)
cam = TorchCamera(make_camera(800, 528, distance=5.0), dev)
with torch.no_grad():
    render(cam, pc, PipelineParams(), bg, scaling_modifier=0.5)       # exact, teaches a small count
    o = render(cam, pc, PipelineParams(), bg, scaling_modifier=3.0)   # speculative: overflows
    n = rasterizer.last_num_rendered()
    int(n)
    _C.set_forward_mode(speculative=False)
    ref = render(cam, pc, PipelineParams(), bg, scaling_modifier=3.0)
print("scale 0.5 -> 3 jump: overflowed", n.overflowed, "redone", n.redone, "equal to the exact frame:",
      bool(torch.equal(o["render"], ref["render"]) and torch.equal(o["semantics"], ref["semantics"])))
