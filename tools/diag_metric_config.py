"""Diagnosis of an element-wise outlier in tests/test_gpu_parity.py::test_metric_configuration_matches_oracle: prints, for
every gradient tensor, the elements whose |hip - oracle| exceeds thr x the tensor's scale, with the Gaussian they belong
to (scales, opacity, depth, radius) and what the other yardsticks say there: the oracle's FMA-contracted twin and the
exact-fp32 flush (bwd_variant 2).   usage: python tools/diag_metric_config.py [P] [yaw] [thr]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from goi_hyperplane_amd import _lib
from goi_hyperplane_amd.render import GaussianSet, PipelineParams, TorchCamera, render
from goi_hyperplane_amd.scene import HEADLINE, make_camera, make_scene
from oracle import oracle
from tests.golden.make_golden import upstream_grads

P = int(sys.argv[1]) if len(sys.argv) > 1 else 3_000_000
yaw = float(sys.argv[2]) if len(sys.argv) > 2 else -0.05
thr = float(sys.argv[3]) if len(sys.argv) > 3 else 5e-4
h = HEADLINE
W, H, S = h["W"], h["H"], h["S"]
dev = torch.device("cuda:0")
sc = make_scene(P, S=S, sh_degree=3, seed=0 if P == h["P"] else 1, extent=h["extent"], log_scale_mean=h["log_scale_mean"],
                log_scale_std=h["log_scale_std"])
cam = make_camera(W, H, fovx=h["fovx"], yaw=yaw)
bg = np.array([0.05, 0.1, 0.2], np.float32)
gc, gs, gd, ga = [g / (W * H) for g in upstream_grads(S, H, W, seed=11)]
tcam, tbg = TorchCamera(cam, dev), torch.tensor(bg, device=dev)


def hip(variant):
    _lib.set_option("bwd_variant", variant)
    pc = GaussianSet.from_scene(sc, dev)
    out = render(tcam, pc, PipelineParams(), tbg)
    loss = ((out["render"] * torch.tensor(gc, device=dev)).sum() + (out["semantics"] * torch.tensor(gs, device=dev)).sum()
            + (out["depth"] * torch.tensor(gd, device=dev)).sum() + (out["alpha"] * torch.tensor(ga, device=dev)).sum())
    loss.backward()
    g = dict(means3D=pc._xyz.grad, opacity=pc._opacity.grad, semantics=pc._semantics.grad, sh=pc._features.grad,
             scales=pc._scaling.grad, rotations=pc._rotation.grad, means2D=out["viewspace_points"].grad)
    _lib.set_option("bwd_variant", 0)
    return {k: v.detach().cpu().numpy() for k, v in g.items()}, out["radii"].cpu().numpy()


g0, radii = hip(0)
g2, _ = hip(2)
orc = {}
for variant in ("", "fma"):
    o = oracle.from_scene(sc, cam, bg=bg, threads=os.cpu_count() or 1, variant=variant)
    o.forward()
    orc[variant] = o.backward(gc, gs, gd, ga)
for name in g0:
    b = np.asarray(orc[""][name]).reshape(g0[name].shape)
    bf = np.asarray(orc["fma"][name]).reshape(g0[name].shape)
    scale = np.abs(b).max() + 1e-20
    d = np.abs(g0[name].astype(np.float64) - b) / scale
    print(f"{name:10s} scale {scale:.3e} max {d.max():.3e} | twin-vs-oracle max {np.abs(bf - b.astype(np.float64)).max() / scale:.3e} "
          f"| fp32-flush-vs-oracle max {np.abs(g2[name].astype(np.float64) - b).max() / scale:.3e} | n>{thr:g}: {(d > thr).sum()}")
    for idx in np.argwhere(d > thr)[:6]:
        gi = int(idx[0])
        it = tuple(idx)
        print(f"    elem {it}: hip {g0[name][it]:+.6e} fp32flush {g2[name][it]:+.6e} oracle {b[it]:+.6e} twin {bf[it]:+.6e} | "
              f"gaussian {gi}: scales {sc.scales[gi]}, opacity {float(sc.opacities[gi]):.3f}, radius {int(radii[gi])}, "
              f"z {float((np.append(sc.means3D[gi], 1.0) @ cam.world_view_transform)[2]):.3f}")
