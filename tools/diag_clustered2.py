"""Diagnosis aid: clustered workload, HIP vs the three oracle builds (plain, FMA twin, exact exponent), forward and backward.
usage: python tools/diag_clustered2.py [name] [P]"""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from goi_hyperplane_amd import _C, _lib, rasterizer
from goi_hyperplane_amd.render import GaussianSet, PipelineParams, TorchCamera, render
from goi_hyperplane_amd.scene import make_workload
from oracle import compare, oracle

name = sys.argv[1] if len(sys.argv) > 1 else "clustered"
P = int(sys.argv[2]) if len(sys.argv) > 2 else None
dev = torch.device("cuda:0")
sc, cam, spec = make_workload(name, P=P)
P, W, H, S = sc.P, spec["W"], spec["H"], spec["S"]
bg = np.zeros(3, np.float32)
rng = np.random.default_rng(99)
gc, gs, gd, ga = (rng.standard_normal((c, H, W)).astype(np.float32) / (W * H) for c in (3, S, 1, 1))
nt = os.cpu_count() or 1
orc = {}
for v in ("", "f64power", "fma"):
    o = oracle.from_scene(sc, cam, bg=bg, threads=nt, variant=v)
    f = o.forward()
    orc[v] = (o, f, o.backward(gc, gs, gd, ga))
f, fx = orc[""][1], orc["f64power"][1]
print("oracle N", f.num_rendered, flush=True)
for cull in (2, 0):
    _lib.set_option("cull_variant", cull)
    pc = GaussianSet.from_scene(sc, dev)
    tcam, tbg = TorchCamera(cam, dev), torch.tensor(bg, device=dev)
    _C.set_forward_mode(speculative=False)
    out = render(tcam, pc, PipelineParams(), tbg)
    torch.autograd.backward((out["render"], out["semantics"], out["depth"], out["alpha"]), [torch.tensor(g_, device=dev) for g_ in (gc, gs, gd, ga)])
    res = {k: out[k].detach().cpu().numpy() for k in ("render", "semantics", "depth", "alpha", "radii")}
    g_hip = dict(means3D=pc._xyz.grad, opacity=pc._opacity.grad, semantics=pc._semantics.grad, sh=pc._features.grad,
                 scales=pc._scaling.grad, rotations=pc._rotation.grad, means2D=out["viewspace_points"].grad)
    g_hip = {k: v.detach().cpu().numpy() for k, v in g_hip.items()}
    fw = compare.forward_stats(res, f, f_exact=fx)
    print(f"cull {cull}: forward", {k: fw[k] for k in fw if not isinstance(fw[k], dict)}, {k: (fw[k]["max"], fw[k]["n_over"]) for k in fw if isinstance(fw[k], dict)}, flush=True)
    # the pixels over tolerance: which flags, what the three builds say
    flags = f.fragile.reshape(-1)
    ok = flags == 0
    ill = ((flags & 2) != 0) & ((flags & 1) == 0) & ((fx.fragile.reshape(-1) & 1) == 0)
    ref = np.where(ill[None], fx.color.reshape(3, -1), f.color.reshape(3, -1))
    d = np.abs(res["render"].reshape(3, -1) - ref).max(axis=0)
    bad = np.nonzero((d > 1e-4) & (ok | ill))[0]
    print("  colour pixels over tol:", len(bad))
    for j in bad[:12]:
        print("   px", j % W, j // W, "flags", int(flags[j]), "exact-flags", int(fx.fragile.reshape(-1)[j]), "err", float(d[j]),
              "hip", res["render"].reshape(3, -1)[:, j], "plain", f.color.reshape(3, -1)[:, j], "exact", fx.color.reshape(3, -1)[:, j],
              "fma", orc["fma"][1].color.reshape(3, -1)[:, j], "alpha hip/plain/exact", float(res["alpha"].reshape(-1)[j]), float(f.alpha.reshape(-1)[j]), float(fx.alpha.reshape(-1)[j]))
    for v in ("", "f64power", "fma"):
        bw = compare.backward_stats(g_hip, orc[v][2])
        print(f"  backward vs {v or 'plain'}:", {k: (float('%.3g' % s_["max"]), s_["n_over"], float('%.2g' % s_["p9999"])) for k, s_ in bw.items()}, flush=True)
    print("  oracle plain vs exact:", {k: float('%.3g' % s_["max"]) for k, s_ in compare.backward_stats(orc[""][2], orc["f64power"][2]).items()})
    print("  oracle plain vs fma:  ", {k: float('%.3g' % s_["max"]) for k, s_ in compare.backward_stats(orc[""][2], orc["fma"][2]).items()})
    # where are the worst gradient elements: which Gaussians
    for name_ in ("means3D", "scales"):
        a = g_hip[name_]; b = np.asarray(orc["f64power"][2][name_]).reshape(a.shape)
        scale = np.abs(b).max()
        e = np.abs(a - b).max(axis=1) / scale
        worst = np.argsort(-e)[:6]
        for g in worst:
            print(f"   {name_} worst g={g}: err {e[g]:.3g} hip {a[g]} exact {b[g]} plain {np.asarray(orc[''][2][name_]).reshape(a.shape)[g]} scales {sc.scales[g]} opac {sc.opacities[g]} radius {f.radii[g]}")
    del pc, out
_lib.set_option("cull_variant", 2)
