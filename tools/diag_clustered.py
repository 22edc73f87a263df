"""Diagnosis aid: the clustered workload through the HIP path against the oracle, stage by stage.
usage: python tools/diag_clustered.py [P] [name]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from goi_hyperplane_amd import _C, _lib, rasterizer
from goi_hyperplane_amd.render import GaussianSet, PipelineParams, TorchCamera, render
from goi_hyperplane_amd.scene import make_workload
from oracle import oracle

P = int(sys.argv[1]) if len(sys.argv) > 1 else 100_000
name = sys.argv[2] if len(sys.argv) > 2 else "clustered"
dev = torch.device("cuda:0")
sc, cam, spec = make_workload(name, P=P)
W, H, S = spec["W"], spec["H"], spec["S"]
bg = np.zeros(3, np.float32)
o = oracle.from_scene(sc, cam, bg=bg, threads=os.cpu_count() or 1)
f = o.forward()
st = o.state()
ok = f.fragile.reshape(-1) == 0
print("oracle N", f.num_rendered, "fragile", 1 - ok.mean(), flush=True)
pc = GaussianSet.from_scene(sc, dev)
tcam, tbg = TorchCamera(cam, dev), torch.tensor(bg, device=dev)
_C.set_forward_mode(speculative=False)
gx = (W + 15) // 16
for opts in ({"cull_variant": 0}, {"cull_variant": 1}, {"cull_variant": 2}, {"cull_variant": 2, "fwd_variant": 0}):
    for k, v in opts.items():
        _lib.set_option(k, v)
    with torch.no_grad():
        args = (tbg, pc._xyz, torch.Tensor([]), pc._semantics, pc._opacity, pc._scaling, pc._rotation, 1.0, torch.Tensor([]),
                tcam.world_view_transform, tcam.full_proj_transform, cam.tanfovx, cam.tanfovy, H, W, pc._features, 3,
                tcam.camera_center, False, False)
        n, color, sem, depth, alpha, radii, geom, binning, img = _C.rasterize_gaussians(*args)
        torch.cuda.synchronize()
    d = np.abs(color.cpu().numpy() - f.color).max(axis=0).reshape(-1)
    bad = (d > 1e-4) & ok
    print(opts, "N", int(n), "max err", float(d[ok].max()), "bad pixels", int(bad.sum()), "radii equal", bool((radii.cpu().numpy() == f.radii).all()), flush=True)
    v = _C.debug_views(P, W, H, n, geom, binning, img)
    nc = v["n_contrib"].cpu().numpy().astype(np.uint32)
    if opts == {"cull_variant": 0}:
        pl = v["point_list"].cpu().numpy().astype(np.uint32)
        rg = v["ranges"].cpu().numpy().astype(np.uint32)
        print("  ranges equal", bool((rg == st["ranges"]).all()), "point_list equal", bool((pl == st["point_list"]).all()),
              "tiles_touched equal", bool((v["tiles_touched"].cpu().numpy().astype(np.uint32) == st["tiles_touched"]).all()),
              "n_contrib equal (non-fragile)", bool((nc[ok] == st["n_contrib"][ok]).all()))
        if not (pl == st["point_list"]).all():
            i = int(np.nonzero(pl != st["point_list"])[0][0])
            print("  first list mismatch at", i, pl[i:i + 8], st["point_list"][i:i + 8])
    if bad.any():
        idx = np.nonzero(bad)[0]
        py, px = idx // W, idx % W
        tiles = (py // 16) * gx + px // 16
        lens = (st["ranges"][:, 1].astype(np.int64) - st["ranges"][:, 0])
        print("  bad pixels: list length of their tiles min/median/max", lens[tiles].min(), np.median(lens[tiles]), lens[tiles].max(),
              "| n_contrib (oracle) min/median/max", st["n_contrib"][idx].min(), np.median(st["n_contrib"][idx]), st["n_contrib"][idx].max())
        print("  all tiles list length median/max", np.median(lens), lens.max(), "distinct bad tiles", len(np.unique(tiles)))
        j = idx[np.argmax(d[idx])]
        print("  worst pixel", j % W, j // W, "err", d[j], "hip", color.cpu().numpy()[:, j // W, j % W], "oracle", f.color[:, j // W, j % W],
              "alpha hip/orc", float(alpha.cpu().numpy().reshape(-1)[j]), float(f.alpha.reshape(-1)[j]), "n_contrib hip/orc", nc[j], st["n_contrib"][j])
    for k in opts:
        _lib.set_option(k, {"cull_variant": 2, "fwd_variant": 1}[k])
