"""On-disk formats: PLY with sem_* columns, code-book files, checkpoint tuple,
k-means init.  kmeans is pinned to outputs of the reference's own function (tests/golden/
make_golden.py: kmeans_pins); the PLY layout to the header grammar plyfile emits for the reference's
`elements` array (scene/gaussian_model.py:283-291)."""
import os

import numpy as np
import pytest
import torch

from goi_hyperplane_amd import io as gio
from goi_hyperplane_amd.semantic import SemanticModel

GOLD = os.path.join(os.path.dirname(__file__), "golden")


def raw_model(P=37, S=16, seed=0):
    g = torch.Generator().manual_seed(seed)
    r = lambda *s: torch.randn(*s, generator=g)  # noqa: E731
    return dict(xyz=r(P, 3), features_dc=r(P, 1, 3), features_rest=r(P, 15, 3), semantics=r(P, S), opacity=r(P, 1),
                scaling=r(P, 3), rotation=r(P, 4))


def test_ply_header_and_payload_are_the_reference_layout(tmp_path):
    m = raw_model()
    path = tmp_path / "point_cloud" / "iteration_7" / "point_cloud.ply"
    gio.save_ply(str(path), **m)
    blob = path.read_bytes()
    head, payload = blob.split(b"end_header\n", 1)
    lines = head.decode().splitlines()
    assert lines[:3] == ["ply", "format binary_little_endian 1.0", "element vertex 37"]
    names = [ln.split()[2] for ln in lines[3:]]
    assert all(ln.startswith("property float ") for ln in lines[3:])
    assert names == (["x", "y", "z", "nx", "ny", "nz", "f_dc_0", "f_dc_1", "f_dc_2"] + [f"f_rest_{i}" for i in range(45)]
                     + [f"sem_{i}" for i in range(16)] + ["opacity", "scale_0", "scale_1", "scale_2"]
                     + [f"rot_{i}" for i in range(4)])
    table = np.frombuffer(payload, dtype="<f4").reshape(37, len(names))
    assert np.array_equal(table[:, 0:3], m["xyz"].numpy()) and not table[:, 3:6].any()
    # f_rest is channel-major: column c*15 + k holds features_rest[:, k, c]
    assert np.array_equal(table[:, 9 + 1 * 15 + 4], m["features_rest"][:, 4, 1].numpy())
    assert np.array_equal(table[:, 54:70], m["semantics"].numpy())


def test_ply_round_trip_and_semantic_dim_rule(tmp_path):
    m = raw_model()
    path = str(tmp_path / "pc.ply")
    gio.save_ply(path, **m)
    back = gio.load_ply(path, max_sh_degree=3, semantic_dim=16)
    for k, v in m.items():
        assert np.array_equal(back[k], v.numpy()), k
    with pytest.warns(UserWarning, match="16 sem_\\* columns but semantic_dim=10"):
        other = gio.load_ply(path, max_sh_degree=3, semantic_dim=10)  # mismatch -> zeros with the FILE's width, loudly
    assert other["semantics"].shape == (37, 16) and not other["semantics"].any()
    auto = gio.load_ply(path, max_sh_degree=3)  # default: whatever the file holds (a default reference run
    assert np.array_equal(auto["semantics"], m["semantics"].numpy())  # saves 10 columns; they must not turn into zeros)
    act = gio.activate(back)
    assert act["shs"].shape == (37, 16, 3) and torch.all(act["scales"] > 0)
    assert torch.allclose(act["rotations"].norm(dim=1), torch.ones(37))


def test_reads_ascii_and_foreign_property_types(tmp_path):
    names = gio.ply_attribute_names(3, 45, 0)
    path = tmp_path / "plain.ply"
    rows = np.arange(3 * len(names), dtype=np.float64).reshape(3, -1) / 7
    with open(path, "w") as f:
        f.write("ply\nformat ascii 1.0\ncomment made by hand\nelement vertex 3\n")
        f.write("".join(f"property {'double' if n == 'x' else 'float'} {n}\n" for n in names))
        f.write("element face 0\nproperty list uchar int vertex_indices\nend_header\n")
        for r in rows:
            f.write(" ".join(repr(float(x)) for x in r) + "\n")
    back = gio.load_ply(str(path), semantic_dim=16)
    assert np.allclose(back["xyz"], rows[:, :3]) and back["semantics"].shape == (3, 16) and not back["semantics"].any()


def test_codebook_files_round_trip(tmp_path):
    mlp = SemanticModel(dim_in=16, dim_out=300, num_layer=1, use_bias=True, device="cpu")
    lut = torch.nn.Parameter(torch.rand(300, 256) * 0.03)
    gio.save_codebook(str(tmp_path), mlp, lut)
    assert sorted(os.listdir(tmp_path)) == ["LUT.pt", "semantic_MLP.pt"]
    raw = torch.load(tmp_path / "semantic_MLP.pt")
    assert set(raw) == {"args", "state_dict"} and raw["args"]["dim_out"] == 300
    mlp2, lut2 = gio.load_codebook(str(tmp_path))
    assert torch.equal(mlp2.layers[0].weight, mlp.layers[0].weight) and torch.equal(lut2, lut)


def test_kmeans_matches_the_reference_function():
    z = np.load(os.path.join(GOLD, "ref_kmeans_pins.npz"))
    for name, k in (("k16", 16), ("k40_dead", 40)):
        x = torch.from_numpy(z["x"].copy())
        torch.manual_seed(int(z["seed"]))
        c = gio.kmeans(x, k)
        assert np.array_equal(x.numpy(), z[name + "_x_after"])  # the in-place normalisation of the input
        ref = z[name + "_centers"]
        assert c.shape == ref.shape
        # same seeds, same assignments; the means are summed in a different order
        assert np.allclose(c.numpy(), ref, rtol=0, atol=2e-6), name


def test_checkpoint_tuple_round_trip(tmp_path):
    from goi_hyperplane_amd.optim import FusedAdam
    m = raw_model(P=11)
    params = {k: torch.nn.Parameter(v) for k, v in m.items()}
    opt = FusedAdam([{"params": [params["semantics"]], "lr": 5e-3, "name": "semantics"}], lr=0.0, eps=1e-15)
    model = dict(active_sh_degree=3, max_radii2D=torch.zeros(11), xyz_gradient_accum=torch.zeros(11, 1),
                 denom=torch.zeros(11, 1), optimizer_state=opt.state_dict(), spatial_lr_scale=1.5, **params)
    path = str(tmp_path / "chkpnt30000.pth")
    gio.save_checkpoint(path, model, 30000)
    (tup, it) = torch.load(path, weights_only=False)  # what the reference's restore() unpacks
    assert it == 30000 and len(tup) == 13 and tup[0] == 3 and tup[12] == 1.5
    assert torch.equal(tup[4], params["semantics"]) and tup[11]["param_groups"][0]["name"] == "semantics"
    back, it2 = gio.load_checkpoint(path)
    assert it2 == 30000 and torch.equal(back["rotation"], params["rotation"])


def _format_pins():
    return np.load(os.path.join(GOLD, "ref_format_pins.npz"))


def test_ply_writer_emits_the_reference_writers_own_records(tmp_path):
    """tests/golden/ref_format_pins.npz holds what the REFERENCE's statements produce (make_golden.py:format_pins, AST-exec
    of scene/gaussian_model.py): construct_list_of_attributes' list and the structured array `elements` save_ply builds and
    hands to plyfile.  This module's writer must put exactly those records behind a header that declares exactly those
    properties (plyfile writes a float32 structured array as its raw little-endian records)."""
    z = _format_pins()
    names = [str(n) for n in z["names"]]
    m = {k[3:]: torch.from_numpy(z[k]) for k in z.files if k.startswith("in_")}
    assert gio.ply_attribute_names(3, 45, m["semantics"].shape[1]) == names
    path = tmp_path / "pc.ply"
    gio.save_ply(str(path), **m)
    head, payload = path.read_bytes().split(b"end_header\n", 1)
    lines = head.decode().splitlines()
    assert lines[:3] == ["ply", "format binary_little_endian 1.0", f"element vertex {int(z['P'])}"]
    assert [ln.split() for ln in lines[3:]] == [["property", "float", n] for n in names]
    assert payload == z["elements_bytes"].tobytes()


@pytest.mark.parametrize("tag,sem_dim", [("", 10), ("_mismatch", 16)])
def test_ply_reader_returns_what_the_reference_reader_builds(tmp_path, tag, sem_dim):
    """... and reading that file back gives the parameter tensors the reference's load_ply statements build from the same
    records (feature layouts after its transposes; its rule for a semantic width that differs from the file's: zeros of
    the FILE's width)."""
    z = _format_pins()
    path = tmp_path / "pc.ply"
    path.write_bytes(("ply\nformat binary_little_endian 1.0\nelement vertex %d\n" % int(z["P"])
                      + "".join(f"property float {n}\n" for n in z["names"]) + "end_header\n").encode()
                     + z["elements_bytes"].tobytes())
    import warnings
    with warnings.catch_warnings():
        warnings.simplefilter("ignore", UserWarning)
        got = gio.load_ply(str(path), max_sh_degree=3, semantic_dim=sem_dim)
    for ours, theirs in (("xyz", "_xyz"), ("features_dc", "_features_dc"), ("features_rest", "_features_rest"),
                         ("opacity", "_opacity"), ("scaling", "_scaling"), ("rotation", "_rotation"),
                         ("semantics", "_semantics")):
        ref = z["load" + tag + theirs]
        assert got[ours].shape == ref.shape and np.array_equal(got[ours], ref), (ours, got[ours].shape, ref.shape)
    if tag:
        assert not got["semantics"].any()


def test_checkpoint_field_order_is_the_reference_capture_order():
    assert list(gio.CHECKPOINT_FIELDS) == [str(n) for n in _format_pins()["capture_order"]]
