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


def _ref_ast(relpath):
    import ast
    return ast.parse(open(os.path.join(REF, relpath)).read())


def _exec_stmts(stmts, ns, label):
    """Executes the reference's OWN statements (AST nodes sliced out of its source, nothing re-typed) in `ns`."""
    import ast
    mod = ast.Module(body=list(stmts), type_ignores=[])
    ast.fix_missing_locations(mod)
    exec(compile(mod, label, "exec"), ns)
    return ns


def _method(tree, cls, name):
    import ast
    c = next(n for n in tree.body if isinstance(n, ast.ClassDef) and n.name == cls)
    return next(n for n in c.body if isinstance(n, ast.FunctionDef) and n.name == name)


class _cuda_is_cpu:
    """The reference hard-codes .cuda() / .to("cuda") / device="cuda"; for the duration of a call those land on the CPU
    (same device redirection as the torch.zeros one in ref_python_pins; no reference code is modified or copied)."""

    def __enter__(self):
        self.saved = (torch.Tensor.cuda, torch.Tensor.to, torch.tensor, torch.zeros)
        real_to, real_tensor, real_zeros = torch.Tensor.to, torch.tensor, torch.zeros

        def to(t, *a, **k):
            a = tuple(x for x in a if not (isinstance(x, str) and x.startswith("cuda")))
            if isinstance(k.get("device"), str) and k["device"].startswith("cuda"):
                k.pop("device")
            return real_to(t, *a, **k) if (a or k) else t

        def strip(fn):
            def f(*a, **k):
                if isinstance(k.get("device"), str) and k["device"].startswith("cuda"):
                    k.pop("device")
                return fn(*a, **k)
            return f
        torch.Tensor.cuda = lambda t, *a, **k: t
        torch.Tensor.to = to
        torch.tensor = strip(real_tensor)
        torch.zeros = strip(real_zeros)
        return self

    def __exit__(self, *exc):
        torch.Tensor.cuda, torch.Tensor.to, torch.tensor, torch.zeros = self.saved
        return False


def semantic_pins():
    """Semantic head, pinned to the reference's OWN statements -- nothing below re-types a line of it:

      * the model is the reference's SemanticModel class (scene/semantic_model.py, loaded by file path because the
        `scene` package imports clip) and the checkpoint is written by its own save();
      * the GUI decode is GUI.compute_similarity (gui/main.py:362-384) -- the method's AST node, compiled as it stands and
        called with a stand-in `self` that only carries the attributes the method reads (renderer.MLP, renderer.LUT,
        res_finetuned, resMLP = the reference's own LinearSVM class from networks.py:12-59, also taken by AST because the
        module imports cv2-dependent helpers);
      * the training losses are the statements of training()'s loop body from `sem_feature = sem_feature.permute(...)` to
        `sem_loss = lab + sl + 0.3 * sl1 + recc` (train.py:141-163), sliced out of the function's AST and executed on
        seeded inputs; the gradients are autograd's through those statements.
    gui/main.py and train.py cannot be imported here (dearpygui / cv2 / the CUDA extensions)."""
    import ast
    import importlib.util
    import types
    spec = importlib.util.spec_from_file_location("ref_semantic_model", os.path.join(REF, "scene", "semantic_model.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    torch.manual_seed(7)
    S, TAB, APE, H, W = 10, 300, 256, 24, 16
    HW = H * W
    mlp = mod.SemanticModel(dim_in=S, dim_out=TAB, num_layer=1, use_bias=True, device="cpu")
    with torch.no_grad():
        mlp.layers[0].bias.normal_(0, 0.1)
    mlp.save(os.path.join(HERE, "ref_semantic_mlp.pt"))
    feats = torch.randn(HW, S)
    lut = torch.rand(TAB, APE) * 0.03 + 0.01 * torch.randn(TAB, APE)
    svm_w = torch.randn(1, APE) * 0.5
    svm_b = torch.tensor([0.3])

    # ---- decode: the reference's GUI.compute_similarity and LinearSVM, as they stand
    net = _ref_ast("networks.py")
    net_ns = {"torch": torch, "nn": torch.nn, "optim": torch.optim, "F": torch.nn.functional}
    _exec_stmts([n for n in net.body if isinstance(n, ast.FunctionDef) and n.name == "inverse_sigmoid"]
                + [n for n in net.body if isinstance(n, ast.ClassDef) and n.name == "LinearSVM"], net_ns, "networks.py")
    svm = net_ns["LinearSVM"]()
    svm.weight_set(svm_w)
    with torch.no_grad():
        svm.linear.bias.copy_(svm_b)
    gui_ns = _exec_stmts([_method(_ref_ast("gui/main.py"), "GUI", "compute_similarity")], {"torch": torch},
                         "gui/main.py:GUI.compute_similarity")
    gui_self = types.SimpleNamespace(renderer=types.SimpleNamespace(MLP=mlp, LUT=lut), res_finetuned=True, resMLP=svm,
                                     vlm=None, clip_feature_thresh=0.5)
    bg = torch.zeros(HW, dtype=torch.bool)
    with _cuda_is_cpu():
        sim_out = gui_ns["compute_similarity"](gui_self, feats, bg)
    with torch.no_grad():
        dec = mlp(feats)
        sem_logit = torch.softmax(dec * 10, dim=-1).argmax(dim=-1)  # (only stored for the tests' index comparison)

    # ---- training losses: train.py's own loop-body statements
    tr = _ref_ast("train.py")
    fn = next(n for n in tr.body if isinstance(n, ast.FunctionDef) and n.name == "training")
    loop = next(n for n in ast.walk(fn) if isinstance(n, ast.For) and getattr(n.target, "id", "") == "iteration")

    def assigns(node, name):
        return isinstance(node, ast.Assign) and any(getattr(t, "id", None) == name for t in node.targets)
    first = next(i for i, n in enumerate(loop.body) if assigns(n, "sem_feature") and "permute" in ast.unparse(n))
    last = next(i for i, n in enumerate(loop.body) if assigns(n, "sem_loss"))
    stmts = loop.body[first:last + 1]
    lines = (stmts[0].lineno, stmts[-1].end_lineno)
    gtl = torch.randn(HW, APE)
    lutp = lut.clone().requires_grad_(True)
    f = feats.clone().requires_grad_(True)
    out = {}
    for tag, iteration in (("", 10), ("_t2", 1500)):  # t = 1 below iteration 1000, 2 afterwards (train.py:156)
        lutp.grad = f.grad = None
        ns = {"torch": torch, "softmax": torch.nn.functional.softmax, "log_softmax": torch.nn.functional.log_softmax,
              "cosine_similarity": torch.nn.functional.cosine_similarity, "iteration": iteration, "semantic_MLP": mlp,
              "lut": lutp, "dataset": types.SimpleNamespace(sem_dim=S, ape_dim=APE),
              "sem_feature": f.T.reshape(S, H, W),  # the rasterizer's [S,H,W] map: permute(1,2,0).reshape gives `f` back
              "viewpoint_cam": types.SimpleNamespace(semantic={"ape": gtl.T.reshape(APE, H, W).clone()})}
        with _cuda_is_cpu():
            _exec_stmts(stmts, ns, f"train.py:{lines[0]}-{lines[1]}")
        ns["sem_loss"].backward()
        out["loss" + tag] = ns["sem_loss"].item()
        out["terms" + tag] = np.array([ns[k].item() for k in ("lab", "sl", "sl1", "recc")])
        out["grad_feats" + tag] = f.grad.numpy().copy()
        out["grad_lut" + tag] = lutp.grad.numpy().copy()
    np.savez_compressed(os.path.join(HERE, "ref_semantic_pins.npz"), feats=feats.numpy(), lut=lut.numpy(),
                        svm_w=svm_w.numpy(), svm_b=svm_b.numpy(), dec=dec.numpy(), idx=sem_logit.numpy(),
                        sim=sim_out.numpy(), bg=bg.numpy(), gtl=gtl.numpy(),
                        ref_lines=np.array(lines), **out)
    print("wrote ref_semantic_pins.npz, ref_semantic_mlp.pt; loss", out["loss"], out["loss_t2"], "train.py lines", lines)


def format_pins():
    """On-disk formats, pinned to the reference's OWN statements of scene/gaussian_model.py (its
    module cannot be imported: plyfile, simple_knn): construct_list_of_attributes (:255-269) as it stands; save_ply
    (:271-289) up to the numpy structured array `elements` it hands to plyfile (the mkdir and the two plyfile lines are
    left out: plyfile is not in this image); load_ply (:307-352) from behind PlyData.read on an object that exposes that
    very `elements` array the way plyfile does (elements[0][name], elements[0].properties[i].name); capture (:53-68)."""
    import types
    tree = _ref_ast("scene/gaussian_model.py")
    g = torch.Generator().manual_seed(3)
    P, S = 7, 10
    r = lambda *s: torch.randn(*s, generator=g)  # noqa: E731
    raw = dict(xyz=r(P, 3), features_dc=r(P, 1, 3), features_rest=r(P, 15, 3), semantics=r(P, S), opacity=r(P, 1),
               scaling=r(P, 3), rotation=r(P, 4))

    class Stub:
        max_sh_degree = 3
        semantic_dim = S
        get_xyz = property(lambda self: self._xyz)
    me = Stub()
    me._xyz, me._features_dc, me._features_rest = raw["xyz"], raw["features_dc"], raw["features_rest"]
    me._semantics, me._opacity, me._scaling, me._rotation = raw["semantics"], raw["opacity"], raw["scaling"], raw["rotation"]
    ns = _exec_stmts([_method(tree, "GaussianModel", "construct_list_of_attributes")], {}, "gaussian_model.py:construct_list")
    Stub.construct_list_of_attributes = ns["construct_list_of_attributes"]
    names = me.construct_list_of_attributes()
    sp = _method(tree, "GaussianModel", "save_ply")
    body = sp.body[1:-2]  # without mkdir_p(...) and the two plyfile lines
    import ast
    assert "mkdir_p" in ast.unparse(sp.body[0]) and "PlyElement" in ast.unparse(sp.body[-2]) and "write" in ast.unparse(sp.body[-1])
    ns = _exec_stmts(body, {"np": np, "torch": torch, "self": me, "path": "unused"}, "gaussian_model.py:save_ply")
    elements = ns["elements"]
    assert list(elements.dtype.names) == names and elements.dtype.itemsize == 4 * len(names)
    out = {"names": np.array(names), "elements_bytes": np.frombuffer(elements.tobytes(), np.uint8), "P": P}
    out.update({"in_" + k: v.numpy() for k, v in raw.items()})

    lp = _method(tree, "GaussianModel", "load_ply")
    assert "PlyData.read" in ast.unparse(lp.body[0])

    class El:
        def __init__(self, arr):
            self.arr = arr
            self.properties = [types.SimpleNamespace(name=n) for n in arr.dtype.names]

        def __getitem__(self, k):
            return self.arr[k]
    for tag, sem_dim in (("", S), ("_mismatch", 16)):
        you = Stub()
        you.semantic_dim = sem_dim
        with _cuda_is_cpu():
            _exec_stmts(lp.body[1:], {"np": np, "torch": torch, "nn": torch.nn, "self": you,
                                      "plydata": types.SimpleNamespace(elements=[El(elements)])}, "gaussian_model.py:load_ply")
        for k in ("_xyz", "_features_dc", "_features_rest", "_opacity", "_scaling", "_rotation", "_semantics"):
            out["load" + tag + k] = getattr(you, k).detach().numpy()
        assert you.active_sh_degree == 3

    cap = Stub()
    for k in ("active_sh_degree", "_xyz", "_features_dc", "_features_rest", "_semantics", "_scaling", "_rotation", "_opacity",
              "max_radii2D", "xyz_gradient_accum", "denom", "spatial_lr_scale"):
        setattr(cap, k, k.lstrip("_"))
    cap.optimizer = types.SimpleNamespace(state_dict=lambda: "optimizer_state")
    ns = _exec_stmts([_method(tree, "GaussianModel", "capture")], {}, "gaussian_model.py:capture")
    out["capture_order"] = np.array(ns["capture"](cap))
    np.savez_compressed(os.path.join(HERE, "ref_format_pins.npz"), **out)
    print("wrote ref_format_pins.npz:", len(names), "attributes,", elements.nbytes, "payload bytes; capture:", list(out["capture_order"]))


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
    format_pins()
    oracle_goldens()
