"""The speculative forward's capacity policy (host logic only: goi_hyperplane_amd/_C.py): which frames are comparable with
the ones it has learnt from.  No GPU, no library call."""
import torch

from goi_hyperplane_amd import _C


def test_tiles():
    assert _C._tiles(1056, 1600) == 66 * 100 and _C._tiles(17, 17) == 4 and _C._tiles(16, 16) == 1


def test_a_frame_is_another_scene_when_count_or_image_changes_by_more_than_two():
    st = {"P": 1000, "tiles": 100, "high_water": 5000, "seen": 7}
    assert not _C._other_scene(st, 1000, 100)
    assert not _C._other_scene(st, 2000, 200) and not _C._other_scene(st, 500, 50)
    assert _C._other_scene(st, 2001, 100) and _C._other_scene(st, 499, 100)
    assert _C._other_scene(st, 1000, 201) and _C._other_scene(st, 1000, 49)
    assert not _C._other_scene(st, 1000, 0)  # image size unknown: not checked
    assert not _C._other_scene({"P": 1000, "high_water": 1, "seen": 1}, 1000, 300)  # nothing recorded yet


def test_note_count_starts_over_and_pick_capacity_follows():
    dev = torch.device("cuda", 7)  # (an index nobody else uses; nothing touches the device)
    _C._SPEC.pop(7, None)
    saved = dict(_C._FWD)
    try:
        _C._FWD.update(mode="speculative", capacity=None, min_history=3, headroom=2.0)
        for _ in range(3):
            _C._note_count(dev, 1000, 5000, 100)
        cap = _C._pick_capacity(dev, 1000, False, False, 100)
        assert cap is not None and cap >= 2 * 5000
        assert _C._pick_capacity(dev, 1000, False, False, 400) is None  # a 4x larger image: exact, and ...
        _C._note_count(dev, 1000, 40000, 400)                           # ... its count starts the history over
        st = _C._SPEC[7]
        assert st["seen"] == 1 and st["high_water"] == 40000 and st["tiles"] == 400
        assert _C._pick_capacity(dev, 1000, False, False, 400) is None  # (two more exact frames to go)
    finally:
        _C._FWD.clear()
        _C._FWD.update(saved)
        _C._SPEC.pop(7, None)


def test_render_result_shows_the_lazy_visibility_key_every_way_one_can_look():
    """render() returns the reference's dictionary (gaussian_renderer/__init__.py:99-105); `visibility_filter` = radii > 0
    is formed on first use, and no way of looking at the dictionary can tell."""
    import torch
    from goi_hyperplane_amd.render import RenderResult
    mk = lambda: RenderResult({"render": 1, "radii": torch.tensor([0, 3, 0, 7])})  # noqa: E731
    want = torch.tensor([False, True, False, True])
    assert torch.equal(mk()["visibility_filter"], want)
    assert torch.equal(mk().get("visibility_filter"), want)
    assert "visibility_filter" in mk() and "nope" not in mk()
    assert set(mk().keys()) == {"render", "radii", "visibility_filter"} == set(mk()) == {k for k, _ in mk().items()}
    assert len(mk()) == 3 and len(list(mk().values())) == 3
    assert set(dict(mk())) == {"render", "radii", "visibility_filter"} and set(mk().copy()) == set(dict(mk()))
    assert "visibility_filter" in repr(mk())
    d = mk()
    a = d["visibility_filter"]
    assert d["visibility_filter"] is a  # formed once
    try:
        d["missing"]
    except KeyError:
        pass
    else:
        raise AssertionError("a key that is not there must raise")


def test_screenspace_placeholder_is_a_leaf_of_zeros_that_costs_no_fill():
    import torch
    from goi_hyperplane_amd.render import _screenspace_placeholder
    xyz = torch.randn(7, 3)
    p = _screenspace_placeholder(xyz)
    assert p.is_leaf and p.requires_grad and p.shape == (7, 3) and p.stride(0) == 0 and float(p.abs().max()) == 0.0
    (p * torch.arange(21.).view(7, 3)).sum().backward()
    assert torch.equal(p.grad, torch.arange(21.).view(7, 3))
    q = _screenspace_placeholder(xyz)
    assert q.grad is None and q.data_ptr() == p.data_ptr()  # the next frame's placeholder: a new leaf over the same zero row


def test_clustered_scene_statistics_are_frozen():
    """tests/golden/calibration.json pins what the generator produces for the two workloads (a change of the generator
    must be a deliberate re-calibration): cheap checks on the arrays themselves, no rendering."""
    import json
    import os
    import numpy as np
    from goi_hyperplane_amd.scene import CLUSTERED, make_scene
    GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
    with open(os.path.join(GOLD, "calibration.json")) as fh:
        cal = json.load(fh)["clustered"]
    c = CLUSTERED
    sc = make_scene(100_000, S=4, seed=0, extent=c["extent"], log_scale_mean=c["log_scale_mean"],
                    log_scale_std=c["log_scale_std"], kind="clustered")
    got = dict(mean_log_scale=float(np.log(sc.scales).mean()), opaque_frac=float((sc.opacities > 0.8).mean()),
               z_front_frac=float((sc.means3D[:, 2] < -1.2).mean()), max_scale=float(sc.scales.max()))
    for k, v in cal["array_stats_P100k"].items():
        assert abs(got[k] - v) <= 1e-4 * max(1.0, abs(v)), (k, got[k], v)
