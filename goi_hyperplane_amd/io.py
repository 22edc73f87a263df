"""On-disk formats around the hot path, byte-compatible with the reference so
that scenes, code books and checkpoints move in both directions:

  * point_cloud.ply        scene/gaussian_model.py:255-358 (save_ply / load_ply): binary little-endian
                           PLY, one `vertex` element, float32 properties x y z nx ny nz f_dc_* f_rest_*
                           sem_* opacity scale_* rot_* holding the RAW (pre-activation) parameters
  * semantic_MLP.pt        scene/semantic_model.py:52-63 ({"args", "state_dict"})  -> semantic.SemanticModel
  * LUT.pt                 train.py:189 (torch.save of the [tab_len, ape_dim] Parameter)
  * chkpnt<iter>.pth       train.py:202 + scene/gaussian_model.py:54-90: ((13-tuple), iteration)
  * k-means code-book init train.py:36-56

The reference reads and writes PLY through the `plyfile` package; this module speaks the format
directly with numpy (header grammar of the PLY 1.0 spec: ascii / binary_little_endian /
binary_big_endian, scalar properties).
"""
from __future__ import annotations

import os

import numpy as np
import torch

_PLY_TYPES = {"char": "i1", "int8": "i1", "uchar": "u1", "uint8": "u1", "short": "i2", "int16": "i2",
              "ushort": "u2", "uint16": "u2", "int": "i4", "int32": "i4", "uint": "u4", "uint32": "u4",
              "float": "f4", "float32": "f4", "double": "f8", "float64": "f8"}


def ply_attribute_names(n_dc: int, n_rest: int, n_sem: int, n_scale: int = 3, n_rot: int = 4) -> list:
    """construct_list_of_attributes (scene/gaussian_model.py:255-270)."""
    names = ["x", "y", "z", "nx", "ny", "nz"]
    names += [f"f_dc_{i}" for i in range(n_dc)]
    names += [f"f_rest_{i}" for i in range(n_rest)]
    names += [f"sem_{i}" for i in range(n_sem)]
    names.append("opacity")
    names += [f"scale_{i}" for i in range(n_scale)]
    names += [f"rot_{i}" for i in range(n_rot)]
    return names


def _np(t):
    return t.detach().cpu().numpy() if isinstance(t, torch.Tensor) else np.asarray(t)


def save_ply(path, xyz, features_dc, features_rest, semantics, opacity, scaling, rotation) -> None:
    """save_ply (scene/gaussian_model.py:272-291).  features_dc [P,1,3], features_rest [P,K,3] in the
    reference's parameter layout; they are written channel-major (transpose(1,2).flatten)."""
    xyz = _np(xyz).astype(np.float32)
    P = xyz.shape[0]
    f_dc = np.ascontiguousarray(np.transpose(_np(features_dc), (0, 2, 1))).reshape(P, -1)
    f_rest = np.ascontiguousarray(np.transpose(_np(features_rest), (0, 2, 1))).reshape(P, -1)
    cols = [xyz, np.zeros_like(xyz), f_dc, f_rest, _np(semantics).reshape(P, -1), _np(opacity).reshape(P, -1),
            _np(scaling).reshape(P, -1), _np(rotation).reshape(P, -1)]
    table = np.ascontiguousarray(np.concatenate([c.astype(np.float32) for c in cols], axis=1), dtype="<f4")
    names = ply_attribute_names(f_dc.shape[1], f_rest.shape[1], cols[4].shape[1], cols[6].shape[1], cols[7].shape[1])
    assert table.shape[1] == len(names)
    d = os.path.dirname(path)
    if d:
        os.makedirs(d, exist_ok=True)  # mkdir_p(os.path.dirname(path))
    header = "ply\nformat binary_little_endian 1.0\nelement vertex %d\n" % P
    header += "".join(f"property float {n}\n" for n in names) + "end_header\n"
    with open(path, "wb") as f:
        f.write(header.encode("ascii"))
        f.write(table.tobytes())


def read_ply_vertex(path) -> np.ndarray:
    """The first element of a PLY file as a numpy structured array (what plydata.elements[0] gives)."""
    with open(path, "rb") as f:
        if f.readline().strip() != b"ply":
            raise ValueError(f"{path}: not a PLY file")
        fmt, count, props, in_first = None, None, [], False
        n_elements = 0
        while True:
            line = f.readline()
            if not line:
                raise ValueError(f"{path}: truncated PLY header")
            tok = line.decode("ascii", "replace").split()
            if not tok or tok[0] in ("comment", "obj_info"):
                continue
            if tok[0] == "format":
                fmt = tok[1]
            elif tok[0] == "element":
                n_elements += 1
                in_first = n_elements == 1
                if in_first:
                    count = int(tok[2])
            elif tok[0] == "property" and in_first:
                if tok[1] == "list":
                    raise ValueError(f"{path}: list properties are not supported in the vertex element")
                props.append((tok[2], _PLY_TYPES[tok[1]]))
            elif tok[0] == "end_header":
                break
        if fmt is None or count is None:
            raise ValueError(f"{path}: malformed PLY header")
        if fmt == "ascii":
            rows = np.loadtxt(f, max_rows=count, ndmin=2, dtype=np.float64)
            out = np.empty(count, dtype=[(n, t) for n, t in props])
            for i, (n, _t) in enumerate(props):
                out[n] = rows[:, i]
            return out
        order = {"binary_little_endian": "<", "binary_big_endian": ">"}[fmt]
        dt = np.dtype([(n, order + t) for n, t in props])
        data = np.frombuffer(f.read(count * dt.itemsize), dtype=dt, count=count)
        return data


REFERENCE_SEM_DIM = 10  # arguments/__init__.py:39 (sem_dim), cuda_rasterizer/config.h:18 (SEM_CHANNELS)


def load_ply(path, max_sh_degree: int = 3, semantic_dim: int | None = None) -> dict:
    """load_ply (scene/gaussian_model.py:308-358): raw parameter arrays in the reference's layout
    (features_dc [P,1,3], features_rest [P,(D+1)^2-1,3]) as float32 numpy arrays.

    semantic_dim=None (default): the semantic features have the width the FILE has (its sem_* columns; a file without
    any gets REFERENCE_SEM_DIM zero columns).  An explicit semantic_dim follows the reference's rule -- a file whose
    number of sem_* columns differs yields ZEROS of the file's width (gaussian_model.py:332-336) -- and warns, because
    decoding or training on those zeros is silent garbage (e.g. a default reference run saves 10 columns)."""
    v = read_ply_vertex(path)
    names = v.dtype.names
    P = v.shape[0]
    xyz = np.stack([v["x"], v["y"], v["z"]], axis=1)
    opacity = np.asarray(v["opacity"])[..., None]
    features_dc = np.zeros((P, 3, 1))
    for c in range(3):
        features_dc[:, c, 0] = v[f"f_dc_{c}"]

    def numbered(prefix):
        return sorted([n for n in names if n.startswith(prefix)], key=lambda x: int(x.split("_")[-1]))

    extra = numbered("f_rest_")
    assert len(extra) == 3 * (max_sh_degree + 1) ** 2 - 3
    features_extra = np.stack([v[n] for n in extra], axis=1).reshape(P, 3, (max_sh_degree + 1) ** 2 - 1) if extra \
        else np.zeros((P, 3, 0))
    sem_names = numbered("sem_")
    want = len(sem_names) if semantic_dim is None else int(semantic_dim)
    sems = np.zeros((P, len(sem_names) or want or REFERENCE_SEM_DIM))
    if len(sem_names) == want:
        for i, n in enumerate(sem_names):
            sems[:, i] = v[n]
    elif sem_names:
        import warnings
        warnings.warn(f"{path}: the file has {len(sem_names)} sem_* columns but semantic_dim={want} was requested; "
                      f"following the reference, the semantic features are ZERO [P,{len(sem_names)}]. Pass "
                      f"semantic_dim=None (or {len(sem_names)}) to load them.", stacklevel=2)
    scales = np.stack([v[n] for n in numbered("scale_")], axis=1)
    rots = np.stack([v[n] for n in numbered("rot")], axis=1)
    f32 = lambda a: np.ascontiguousarray(a, dtype=np.float32)  # noqa: E731
    return {"xyz": f32(xyz), "features_dc": f32(np.transpose(features_dc, (0, 2, 1))),
            "features_rest": f32(np.transpose(features_extra, (0, 2, 1))), "semantics": f32(sems),
            "opacity": f32(opacity), "scaling": f32(scales), "rotation": f32(rots)}


def activate(raw: dict) -> dict:
    """Raw parameters -> what the rasterizer consumes (scene/gaussian_model.py:39-53,90-117:
    exp scaling, sigmoid opacity, normalised rotation, dc + rest concatenated)."""
    t = {k: torch.as_tensor(v) for k, v in raw.items()}
    return {"means3D": t["xyz"], "scales": torch.exp(t["scaling"]),
            "rotations": torch.nn.functional.normalize(t["rotation"]), "opacities": torch.sigmoid(t["opacity"]),
            "shs": torch.cat((t["features_dc"], t["features_rest"]), dim=1), "semantics": t["semantics"]}


# ---- code book -----------------------------------------------------------------------------------
def save_codebook(directory, semantic_mlp, lut) -> None:
    """train.py:187-189: semantic_MLP.pt + LUT.pt next to point_cloud.ply."""
    os.makedirs(directory, exist_ok=True)
    semantic_mlp.save(os.path.join(directory, "semantic_MLP.pt"))
    torch.save(lut, os.path.join(directory, "LUT.pt"))


def load_codebook(directory, map_location=None):
    """gui/gs_renderer.py:214-228."""
    from .semantic import SemanticModel
    mlp = SemanticModel.load(os.path.join(directory, "semantic_MLP.pt"), map_location=map_location)
    lut = torch.load(os.path.join(directory, "LUT.pt"), map_location=map_location)
    return mlp, lut


def kmeans(x: torch.Tensor, ncluster: int, niter: int = 10) -> torch.Tensor:
    """train.py:36-56: spherical k-means used to initialise the code book.  Same RNG consumption
    (one randperm for the seeds, one per iteration for dead clusters) and the same in-place
    normalisation of `x`; the per-cluster means are one index_add instead of `ncluster` masked means."""
    N, D = x.size()
    x /= x.norm(dim=1, keepdim=True)
    centers = x[torch.randperm(N)[:ncluster]]
    for _ in range(niter):
        centers = centers / centers.norm(dim=1, keepdim=True)
        assignments = (x @ centers.T).argmax(1)
        sums = torch.zeros((ncluster, D), dtype=x.dtype, device=x.device).index_add_(0, assignments, x)
        counts = torch.bincount(assignments, minlength=ncluster).to(x.dtype)
        centers = sums / counts[:, None]  # empty cluster -> 0/0 = nan, like the mean of an empty selection
        nanix = torch.any(torch.isnan(centers), dim=1)
        ndead = int(nanix.sum().item())
        centers[nanix] = x[torch.randperm(N)[:ndead]]
    return centers


# ---- checkpoint ------------------------------------------------------------------------------------
CHECKPOINT_FIELDS = ("active_sh_degree", "xyz", "features_dc", "features_rest", "semantics", "scaling", "rotation",
                     "opacity", "max_radii2D", "xyz_gradient_accum", "denom", "optimizer_state", "spatial_lr_scale")


def save_checkpoint(path, model: dict, iteration: int) -> None:
    """train.py:202: torch.save((gaussians.capture(), iteration), path); capture() is the 13-tuple of
    scene/gaussian_model.py:54-69 in CHECKPOINT_FIELDS order."""
    torch.save((tuple(model[k] for k in CHECKPOINT_FIELDS), iteration), path)


def load_checkpoint(path, map_location=None):
    """-> (dict keyed by CHECKPOINT_FIELDS, iteration)  (train.py:72-73, gaussian_model.py:71-88)."""
    model_params, iteration = torch.load(path, map_location=map_location, weights_only=False)
    if len(model_params) != len(CHECKPOINT_FIELDS):
        raise ValueError(f"{path}: expected a {len(CHECKPOINT_FIELDS)}-tuple, got {len(model_params)}")
    return dict(zip(CHECKPOINT_FIELDS, model_params)), iteration
