#!/usr/bin/env python3
"""Generates the committed golden fixtures under tests/golden/.  Run in the AUTHORING container
only (it imports the reference's Python helpers from /root/reference; the GPU box has neither).

    python tests/golden/make_golden.py            # writes ref_python_pins.npz + oracle_*.npz

Two kinds of fixture:

1. ref_python_pins.npz -- inputs and outputs of the REFERENCE's own importable Python functions
   for the pieces of the hot path that exist in Python as well as in CUDA:
     utils/sh_utils.py:eval_sh            (convert_SHs_python path, gaussian_renderer/__init__.py:73-78)
     utils/general_utils.py:build_scaling_rotation / strip_symmetric
                                          (compute_cov3D_python path, scene/gaussian_model.py:33-37)
     utils/graphics_utils.py:getWorld2View2 / getProjectionMatrix and the four matrix lines of
     scene/cameras.py:45-48.
   These pin the oracle's SH->RGB, cov3D and the camera conventions of goi_hyperplane_amd/scene.py.
   general_utils hard-codes device="cuda" in torch.zeros(); the generator maps that to the CPU
   for the duration of the call (no reference code is modified or copied).

2. oracle_<name>.npz -- full forward + backward outputs of oracle/liboracle.so on small seeded
   scenes.  They pin the ORACLE ITSELF against accidental change and give the GPU parity tests a
   fixture that does not depend on rebuilding the oracle.  They are NOT reference outputs (the
   CUDA reference cannot run here) -- see DESIGN.md "parity status".
"""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
REF = "/root/reference"


def ref_python_pins():
    sys.path.insert(0, REF)
    from utils.sh_utils import eval_sh  # noqa
    from utils import graphics_utils as gu  # noqa
    from utils import general_utils as ge  # noqa

    rng = np.random.default_rng(1234)
    out = {}
    # ---- SH -> RGB, degrees 0..3
    P = 96
    means = rng.uniform(-2, 2, size=(P, 3)).astype(np.float32)
    campos = np.array([0.3, -0.2, -5.0], np.float32)
    shs = np.concatenate([rng.normal(size=(P, 1, 3)), 0.4 * rng.normal(size=(P, 15, 3))], 1).astype(np.float32)
    out["sh_means"], out["sh_campos"], out["sh_shs"] = means, campos, shs
    for deg in range(4):
        shs_view = torch.tensor(shs).transpose(1, 2).reshape(-1, 3, 16)
        dir_pp = torch.tensor(means) - torch.tensor(campos)[None].repeat(P, 1)
        dir_n = dir_pp / dir_pp.norm(dim=1, keepdim=True)
        sh2rgb = eval_sh(deg, shs_view, dir_n)
        out[f"sh_rgb_deg{deg}"] = torch.clamp_min(sh2rgb + 0.5, 0.0).numpy()

    # ---- cov3D from scaling + rotation (python path), device="cuda" redirected to CPU
    scales = np.exp(rng.normal(-2.0, 0.7, size=(P, 3))).astype(np.float32)
    rots = rng.normal(size=(P, 4)).astype(np.float32)
    real_zeros = torch.zeros

    def cpu_zeros(*a, **k):
        k.pop("device", None)
        return real_zeros(*a, **k)

    for mod in (1.0, 0.7):
        torch.zeros = cpu_zeros
        try:
            L = ge.build_scaling_rotation(mod * torch.tensor(scales), torch.tensor(rots))
            cov = ge.strip_symmetric(L @ L.transpose(1, 2))
        finally:
            torch.zeros = real_zeros
        out[f"cov3D_mod{mod}"] = cov.numpy()
    out["cov_scales"], out["cov_rots"] = scales, rots

    # ---- camera matrices
    cams = []
    for i in range(4):
        ang = 0.4 * i
        Rw2c = np.array([[np.cos(ang), 0, np.sin(ang)], [0, 1, 0], [-np.sin(ang), 0, np.cos(ang)]])
        R = Rw2c.T  # readers store the transposed rotation
        T = np.array([0.1 * i, -0.05 * i, 5.0 + 0.3 * i])
        fovx, fovy = 1.0 + 0.1 * i, 0.7 + 0.05 * i
        wvt = torch.tensor(gu.getWorld2View2(R, T, np.array([0.0, 0.0, 0.0]), 1.0)).transpose(0, 1)
        proj = gu.getProjectionMatrix(znear=0.01, zfar=100.0, fovX=fovx, fovY=fovy).transpose(0, 1)
        full = (wvt.unsqueeze(0).bmm(proj.unsqueeze(0))).squeeze(0)
        center = wvt.inverse()[3, :3]
        cams.append(dict(R=R, T=T, fovx=fovx, fovy=fovy, wvt=wvt.numpy(), proj=proj.numpy(), full=full.numpy(),
                         center=center.numpy()))
    for k in cams[0]:
        out["cam_" + k] = np.stack([np.asarray(c[k], dtype=np.float64 if k in ("R", "T") else np.float32) for c in cams])
    np.savez_compressed(os.path.join(HERE, "ref_python_pins.npz"), **out)
    print("wrote ref_python_pins.npz", {k: v.shape for k, v in out.items()})


ORACLE_CASES = {
    # name: (P, S, W, H, log_scale_mean, kwargs)
    "s10_sh3": dict(P=1500, S=10, W=128, H=96, mu=-2.8, deg=3),
    "s16_sh3": dict(P=1500, S=16, W=128, H=96, mu=-2.8, deg=3),
    "s16_ragged": dict(P=1200, S=16, W=123, H=77, mu=-2.6, deg=2),
}


def upstream_grads(S, H, W, seed=99):
    """Seeded upstream gradients (dL/dcolor, dL/dsemantic, dL/ddepth, dL/dalpha) shared by the
    generator and the tests, so they need not be stored."""
    rng = np.random.default_rng(seed)
    gc = rng.normal(size=(3, H, W)).astype(np.float32)
    gs = rng.normal(size=(S, H, W)).astype(np.float32)
    gd = rng.normal(size=(1, H, W)).astype(np.float32)
    ga = rng.normal(size=(1, H, W)).astype(np.float32)
    return gc, gs, gd, ga


def oracle_goldens():
    from goi_hyperplane_amd.scene import make_camera, make_scene
    from oracle import oracle

    oracle.build(force=True)
    for name, c in ORACLE_CASES.items():
        sc = make_scene(c["P"], S=c["S"], sh_degree=c["deg"], seed=7, log_scale_mean=c["mu"])
        cam = make_camera(c["W"], c["H"], yaw=0.15, pitch=-0.1)
        bg = np.array([0.1, 0.2, 0.3], np.float32)
        o = oracle.from_scene(sc, cam, bg=bg)
        f = o.forward()
        gc, gs, gd, ga = upstream_grads(c["S"], c["H"], c["W"])
        g = o.backward(gc, gs, gd, ga)
        st = o.state()
        np.savez_compressed(
            os.path.join(HERE, f"oracle_{name}.npz"),
            cfg=np.array([c["P"], c["S"], c["W"], c["H"], c["deg"]]), mu=c["mu"], bg=bg,
            color=f.color, semantic=f.semantic, depth=f.depth, alpha=f.alpha, radii=f.radii, fragile=f.fragile,
            num_rendered=f.num_rendered, n_contrib=st["n_contrib"], point_list=st["point_list"], ranges=st["ranges"],
            **{"grad_" + k: v for k, v in g.items()})  # upstream grads: regenerate with upstream_grads()
        print("wrote", name, "N", f.num_rendered, "fragile", int(f.fragile.sum()))


if __name__ == "__main__":
    ref_python_pins()
    oracle_goldens()
