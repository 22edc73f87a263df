"""Calibrates the headline generator's log-scale mean so that num_rendered / P is about 8 at
1600x1056 ('calibrate once, then freeze').  Run on a GPU box."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from goi_hyperplane_amd import _C
from goi_hyperplane_amd.scene import HEADLINE, make_camera, make_scene
from goi_hyperplane_amd.render import TorchCamera

dev = torch.device("cuda:0")
P = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
cam = make_camera(HEADLINE["W"], HEADLINE["H"], fovx=HEADLINE["fovx"])
tc = TorchCamera(cam, dev)
for mu in (-4.6, -4.4, -4.2, -4.1, -4.0, -3.9, -3.8, -3.6):
    sc = make_scene(P, S=16, seed=0, extent=HEADLINE["extent"], log_scale_mean=mu)
    t = lambda a: torch.tensor(a, device=dev)
    n, color, sem, depth, alpha, radii, *_ = _C.rasterize_gaussians(
        torch.zeros(3, device=dev), t(sc.means3D), torch.Tensor([]), t(sc.semantics), t(sc.opacities), t(sc.scales),
        t(sc.rotations), 1.0, torch.Tensor([]), tc.world_view_transform, tc.full_proj_transform, cam.tanfovx, cam.tanfovy,
        cam.image_height, cam.image_width, t(sc.shs), 3, tc.camera_center, False, False)
    print(f"mu {mu:5.2f}  N {n:9d}  N/P {n / P:6.2f}  visible {(radii > 0).sum().item()}  mean alpha {alpha.mean().item():.3f}")
