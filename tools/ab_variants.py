"""Within-process interleaved A/B of kernel variants (goi_raster_set_option) on the headline workload.
usage: python tools/ab_variants.py fwd_variant 0 1 [--bwd]   (prints per-variant median stage ms)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from goi_hyperplane_amd import _lib
from goi_hyperplane_amd.render import GaussianSet, PipelineParams, TorchCamera, render
from goi_hyperplane_amd.scene import HEADLINE, make_camera, make_scene

opt = sys.argv[1]
values = [int(v) for v in sys.argv[2:] if not v.startswith("--")]
do_bwd = "--bwd" in sys.argv
P = int(os.environ.get("GOI_AB_P", HEADLINE["P"]))
dev = torch.device("cuda:0")
sc = make_scene(P, S=16, seed=0, extent=HEADLINE["extent"], log_scale_mean=HEADLINE["log_scale_mean"])
pc = GaussianSet.from_scene(sc, dev)
cams = [TorchCamera(make_camera(HEADLINE["W"], HEADLINE["H"], yaw=0.02 * (i - 4)), dev) for i in range(8)]
bg = torch.zeros(3, device=dev)
pipe = PipelineParams()
inv = 1.0 / (HEADLINE["W"] * HEADLINE["H"])
res = {v: {} for v in values}
ref = {}
for rnd in range(6):
    for v in values:
        _lib.set_option(opt, v)
        _lib.profile_collect()
        _lib.profile_enable(True)
        for i, cam in enumerate(cams):
            for p in pc.parameters():
                p.grad = None
            if do_bwd:
                out = render(cam, pc, pipe, bg)
                ((out["render"].sum() + out["semantics"].sum()) * inv).backward()
            else:
                with torch.no_grad():
                    out = render(cam, pc, pipe, bg)
            if rnd == 0 and i == 0:
                key = (out["render"].double().sum().item(), out["semantics"].double().sum().item(),
                       pc._semantics.grad.double().abs().sum().item() if do_bwd else 0.0)
                ref[v] = key
        torch.cuda.synchronize()
        _lib.profile_enable(False)
        st = _lib.profile_collect()
        if rnd > 0:
            for k, (ms, n) in st.items():
                if n:
                    res[v].setdefault(k, []).append(ms / n)
for v in values:
    print("variant", v, "checksums", ref[v], {k: round(float(np.median(x)), 4) for k, x in res[v].items()})
