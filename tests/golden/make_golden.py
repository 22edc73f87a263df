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


def semantic_pins():
    """Semantic head: outputs of the REFERENCE's SemanticModel class
    (scene/semantic_model.py, loaded by file path because the `scene` package imports clip) and a
    checkpoint written by its own save(); the gui/main.py:364-386 decode and the train.py:142-163
    losses are evaluated here with that model (those two files cannot be imported: dearpygui / cv2)."""
    import importlib.util
    spec = importlib.util.spec_from_file_location("ref_semantic_model", os.path.join(REF, "scene", "semantic_model.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    torch.manual_seed(7)
    S, TAB, APE, HW = 10, 300, 256, 24 * 16
    mlp = mod.SemanticModel(dim_in=S, dim_out=TAB, num_layer=1, use_bias=True, device="cpu")
    with torch.no_grad():
        mlp.layers[0].bias.normal_(0, 0.1)
    mlp.save(os.path.join(HERE, "ref_semantic_mlp.pt"))
    feats = torch.randn(HW, S)
    lut = torch.rand(TAB, APE) * 0.03 + 0.01 * torch.randn(TAB, APE)
    svm_w = torch.randn(1, APE) * 0.5
    svm_b = torch.tensor([0.3])
    with torch.no_grad():
        dec = mlp(feats)
        sem_logit = torch.softmax(dec * 10, dim=-1).argmax(dim=-1)            # gui/main.py:366
        sem_feature = lut[sem_logit]                                          # :367
        normed = sem_feature / sem_feature.norm(dim=-1, keepdim=True)         # :370
        logit = torch.nn.functional.linear(normed / 0.3438, svm_w, svm_b).squeeze()  # networks.py:56-57
        sim = logit.sigmoid()                                                 # :374
        bg = sim < 0.5                                                        # :379
        sim_out = sim.clone()
        sim_out[bg] = 0                                                       # :382
    # training losses, train.py:142-163, iteration < 1000
    gtl = torch.randn(HW, APE)
    lutp = lut.clone().requires_grad_(True)
    f = feats.clone().requires_grad_(True)
    sem_label = torch.softmax(mlp(f), dim=-1)
    g = gtl / gtl.norm(dim=1, keepdim=True)
    lut1 = lutp / lutp.norm(dim=1, keepdim=True)
    simm = g @ lut1.T
    sim_val = simm.max(dim=1, keepdim=True)[0]
    label = (simm == sim_val).float().detach()
    lab = torch.nn.MSELoss()(sem_label, label) * 50
    sl = (1 - sim_val.mean())
    recc = 1 - torch.nn.functional.cosine_similarity(lutp[sem_label.argmax(-1)], g, dim=-1).mean()
    b = torch.softmax(simm * 1, dim=1) * torch.log_softmax(simm * 1, dim=1)
    sl1 = -1.0 * b.sum(dim=-1).mean()
    loss = lab + sl + 0.3 * sl1 + recc
    loss.backward()
    np.savez_compressed(os.path.join(HERE, "ref_semantic_pins.npz"), feats=feats.numpy(), lut=lut.numpy(),
                        svm_w=svm_w.numpy(), svm_b=svm_b.numpy(), dec=dec.numpy(), idx=sem_logit.numpy(),
                        sim=sim_out.numpy(), bg=bg.numpy(), gtl=gtl.numpy(), loss=loss.item(),
                        terms=np.array([lab.item(), sl.item(), sl1.item(), recc.item()]),
                        grad_feats=f.grad.numpy(), grad_lut=lutp.grad.numpy())
    print("wrote ref_semantic_pins.npz, ref_semantic_mlp.pt; loss", loss.item())


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


def knn_golden():
    """distCUDA2 has no reference-side vectors (CUDA-only implementation, no tests); the fixture is an
    independent exact answer: float64 k-d tree, k = 4 including the point itself."""
    from scipy.spatial import cKDTree

    rng = np.random.default_rng(11)
    c = rng.standard_normal((12, 3)) * 3
    pts = (c[rng.integers(0, 12, 3000)] + 0.2 * rng.standard_normal((3000, 3))).astype(np.float32)
    d, _ = cKDTree(pts.astype(np.float64)).query(pts.astype(np.float64), k=4)
    np.savez_compressed(os.path.join(HERE, "knn_kdtree64.npz"), points=pts, mean_dist2=(d[:, 1:] ** 2).mean(1))
    print("wrote knn_kdtree64")


def kmeans_pins():
    """Runs the reference's kmeans (train.py:36-56; the function only -- the module itself imports the
    CUDA extensions) on a small seeded input and stores input, seed and resulting centres."""
    import ast

    src = open(os.path.join(REF, "train.py")).read()
    fn = next(n for n in ast.parse(src).body if isinstance(n, ast.FunctionDef) and n.name == "kmeans")
    ns = {"torch": torch}
    exec(compile(ast.Module(body=[fn], type_ignores=[]), "train.py:kmeans", "exec"), ns)
    g = torch.Generator().manual_seed(5)
    base = torch.randn(24, 32, generator=g)
    x = base[torch.randint(0, 24, (900,), generator=g)] + 0.15 * torch.randn(900, 32, generator=g)
    out = {}
    for name, k in (("k16", 16), ("k40_dead", 40)):
        xin = x.clone()
        torch.manual_seed(1234)
        c = ns["kmeans"](xin, k)
        out[name + "_centers"] = c.numpy()
        out[name + "_x_after"] = xin.numpy()
    np.savez_compressed(os.path.join(HERE, "ref_kmeans_pins.npz"), x=x.numpy(), seed=1234, **out)
    print("wrote ref_kmeans_pins")


if __name__ == "__main__":
    knn_golden()
    kmeans_pins()
    ref_python_pins()
    semantic_pins()
    oracle_goldens()
