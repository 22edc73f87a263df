"""A loop of raw-operator forward + FACTORED backward calls on the headline workload (dL_dsh = NULL: preprocess_bwd_k
without its dL/dSH tile): what kstats.sh profiles to see that kernel's floor.  tools only."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from goi_hyperplane_amd import _C
from goi_hyperplane_amd.render import GaussianSet, TorchCamera
from goi_hyperplane_amd.scene import HEADLINE, make_camera, make_headline_scene

h = HEADLINE
dev = torch.device("cuda:0")
sc = make_headline_scene()
pc = GaussianSet.from_scene(sc, dev)
cam = make_camera(h["W"], h["H"], fovx=h["fovx"])
tc = TorchCamera(cam, dev)
args = (torch.zeros(3, device=dev), pc._xyz.detach(), torch.Tensor([]), pc._semantics.detach(), pc._opacity.detach(),
        pc._scaling.detach(), pc._rotation.detach(), 1.0, torch.Tensor([]), tc.world_view_transform, tc.full_proj_transform,
        cam.tanfovx, cam.tanfovy, cam.image_height, cam.image_width, pc._features.detach(), sc.sh_degree, tc.camera_center,
        False, False)
g = torch.Generator(device=dev).manual_seed(1)
ups = [torch.randn(s, device=dev, generator=g) for s in ((3, h["H"], h["W"]), (h["S"], h["H"], h["W"]), (1, h["H"], h["W"]), (1, h["H"], h["W"]))]
factored = "--full" not in sys.argv
fn = _C.rasterize_gaussians_backward_sh_factored if factored else _C.rasterize_gaussians_backward
for i in range(24):
    n, color, sem, depth, alpha, radii, geom, binning, img = _C.rasterize_gaussians(*args)
    fn(args[0], args[1], radii, args[2], args[3], args[5], args[6], args[7], args[8], args[9], args[10], args[11], args[12],
       ups[0], ups[1], ups[2], ups[3], args[15], args[16], args[17], geom, n, binning, img, alpha, False)
torch.cuda.synchronize()
print("done", "factored" if factored else "full")
