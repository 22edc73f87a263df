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
