"""Measured lane utilisation of the blend kernels (goi_raster_blend_stats) on the named workloads: one forward per workload,
the device counters, the derived ratios.  usage: python tools/blend_stats.py [out.json] [workloads...]  (GPU)"""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from goi_hyperplane_amd import _C
from goi_hyperplane_amd.render import GaussianSet, TorchCamera
from goi_hyperplane_amd.scene import make_workload


def stats_of(name, P=None):
    dev = torch.device("cuda:0")
    sc, cam, spec = make_workload(name, P=P)
    pc, tcam = GaussianSet.from_scene(sc, dev), TorchCamera(cam, dev)
    W, H = spec["W"], spec["H"]
    args = (torch.zeros(3, device=dev), pc._xyz.detach(), torch.Tensor([]), pc._semantics.detach(), pc._opacity.detach(),
            pc._scaling.detach(), pc._rotation.detach(), 1.0, torch.Tensor([]), tcam.world_view_transform,
            tcam.full_proj_transform, cam.tanfovx, cam.tanfovy, H, W, pc._features.detach(), sc.sh_degree, tcam.camera_center,
            False, False)
    n, *_rest, geom, binning, img = _C.rasterize_gaussians(*args)
    n_true = int(n)
    st = _C.blend_stats(sc.P, W, H, n, geom, binning, img)
    st.update(workload=name, P=sc.P, W=W, H=H, listed_instances=n_true)
    return st


if __name__ == "__main__":
    out = sys.argv[1] if len(sys.argv) > 1 else None
    names = sys.argv[2:] or ["headline", "clustered", "closeup", "headline:3000000"]
    res = []
    for nm in names:
        name, _, p = nm.partition(":")
        st = stats_of(name, int(p) if p else None)
        res.append(st)
        print(json.dumps({k: (round(v, 4) if isinstance(v, float) else v) for k, v in st.items()}))
    if out:
        json.dump(res, open(out, "w"), indent=1)
