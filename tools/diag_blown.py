"""Which undecided rows of the clustered workload does oracle/compare.py call "blown up", and what do the three builds of the
oracle and this build hold there?  usage: python tools/diag_blown.py   (test tooling; the oracle is the checker)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from goi_hyperplane_amd import rasterizer
from goi_hyperplane_amd.render import GaussianSet, PipelineParams, TorchCamera, render
from goi_hyperplane_amd.scene import make_workload
from oracle import oracle as oracle_mod

dev = torch.device("cuda:0")
sc, cam, spec = make_workload("clustered")
P, S, W, H = spec["P"], spec["S"], spec["W"], spec["H"]
bg = np.zeros(3, np.float32)
rng = np.random.default_rng(99)
gc, gs, gd, ga = (rng.standard_normal((c, H, W)).astype(np.float32) / (W * H) for c in (3, S, 1, 1))
pc = GaussianSet.from_scene(sc, dev)
tcam, tbg, pipe = TorchCamera(cam, dev), torch.tensor(bg, device=dev), PipelineParams()
for _ in range(3):
    render(tcam, pc, pipe, tbg)
out = render(tcam, pc, pipe, tbg)
int(rasterizer.last_num_rendered())
torch.autograd.backward((out["render"], out["semantics"], out["depth"], out["alpha"]), [torch.tensor(g, device=dev) for g in (gc, gs, gd, ga)])
g_hip = dict(means3D=pc._xyz.grad, scales=pc._scaling.grad, rotations=pc._rotation.grad)
g_hip = {k: v.detach().cpu().numpy().astype(np.float64) for k, v in g_hip.items()}
nt = os.cpu_count() or 1
builds = []
for variant in (None, "f64power", "fma"):
    o = oracle_mod.from_scene(sc, cam, bg=bg, threads=nt, **({} if variant is None else {"variant": variant}))
    f = o.forward()
    builds.append(o.backward(gc, gs, gd, ga))
    if variant is None:
        tiles = o.state()["tiles_touched"]
tol = 1e-3
und = np.zeros(P, bool)
for name in g_hip:
    b0 = np.asarray(builds[0][name], np.float64).reshape(P, -1)
    scale = np.abs(b0).max()
    for other in builds[1:]:
        und |= (np.abs(np.asarray(other[name], np.float64).reshape(P, -1) - b0).max(axis=1) / scale) > tol / 3
print("undecided rows", int(und.sum()))
for name in g_hip:
    a = g_hip[name].reshape(P, -1)
    bs = [np.asarray(b[name], np.float64).reshape(P, -1) for b in builds]
    scale = np.abs(bs[0]).max()
    big = np.maximum.reduce([np.abs(b) for b in bs])
    blown = (np.abs(a) > 10 * big + tol * scale) & und[:, None]
    rows = np.nonzero(blown.any(axis=1))[0]
    print(name, "scale %.3g" % scale, "blown rows", rows.tolist())
    for r in rows[:5]:
        print("  row", r, "tiles", int(tiles[r]), "scale(log)", np.asarray(sc.log_scales[r]) if hasattr(sc, "log_scales") else "")
        print("   hip  ", (a[r] / scale).round(5).tolist())
        for b, nm in zip(bs, ("plain", "f64pw", "fma  ")):
            print("   " + nm, (b[r] / scale).round(5).tolist())
