"""distCUDA2 on the CPU: the oracle's restatement of the reference's Morton/box
search (oracle/knn_oracle.cpp, after submodules/simple-knn/simple_knn.cu:45-221) against two
independent exact searches, plus the drop-in import surface."""
import os

import numpy as np
import pytest
import torch
from scipy.spatial import cKDTree

from oracle import oracle

GOLD = os.path.join(os.path.dirname(__file__), "golden")


def clouds():
    rng = np.random.default_rng(7)
    out = {}
    out["gauss_aniso"] = (rng.standard_normal((6000, 3)) * [3, 1, 0.2] + [1, -2, 5]).astype(np.float32)
    c = rng.standard_normal((40, 3)) * 4
    out["clusters"] = (c[rng.integers(0, 40, 5000)] + 0.05 * rng.standard_normal((5000, 3))).astype(np.float32)
    g = np.stack(np.meshgrid(np.arange(16), np.arange(16), np.arange(12), indexing="ij"), -1).reshape(-1, 3)
    out["lattice_ties"] = (g * 0.25).astype(np.float32)  # every point has >= 3 equidistant neighbours
    d = rng.standard_normal((1500, 3)).astype(np.float32)
    out["duplicates"] = np.concatenate([d, d[:700], d[:100]])  # coincident points: distance 0 counts
    out["positive_octant"] = (rng.random((3000, 3)) * 2 + 10).astype(np.float32)  # origin-anchored Morton box
    out["box_edge_1025"] = rng.standard_normal((1025, 3)).astype(np.float32)
    out["collinear"] = np.stack([np.linspace(-3, 3, 2100), np.zeros(2100), np.zeros(2100)], 1).astype(np.float32)
    return out


@pytest.mark.parametrize("name", list(clouds()))
@pytest.mark.parametrize("fma", [False, True])
def test_restated_search_equals_exhaustive_search(name, fma):
    pts = clouds()[name]
    a = oracle.knn_mean_dist2(pts, fma=fma)
    b = oracle.knn_mean_dist2(pts, fma=fma, brute=True)
    assert np.array_equal(a, b), f"{name}: pruned search differs from the O(P^2) definition"


@pytest.mark.parametrize("name", ["gauss_aniso", "clusters", "positive_octant", "duplicates"])
def test_oracle_against_float64_kdtree(name):
    pts = clouds()[name]
    p64 = pts.astype(np.float64)
    d, _ = cKDTree(p64).query(p64, k=4)
    ref = (d[:, 1:] ** 2).mean(1)
    got = oracle.knn_mean_dist2(pts).astype(np.float64)
    # coordinates are exact in both; only the fp32 rounding of 3 products, 2 sums and the mean differs
    scale = (np.abs(p64).max() ** 2) * 2 ** -22
    assert np.all(np.abs(got - ref) <= 4e-7 * ref + scale * 1e-1)


def test_small_counts_follow_the_reference_formula():
    fmax = np.float32(np.finfo(np.float32).max)
    one = oracle.knn_mean_dist2(np.zeros((1, 3), np.float32))
    assert np.isinf(one[0])  # (FLT_MAX + FLT_MAX + FLT_MAX) / 3 overflows in fp32
    three = oracle.knn_mean_dist2(np.array([[0, 0, 0], [1, 0, 0], [0, 2, 0]], np.float32))
    assert np.array_equal(three, np.array([(np.float32(1) + np.float32(4) + fmax) / np.float32(3)] * 0 +
                                          [((np.float32(1) + np.float32(4)) + fmax) / np.float32(3),
                                           ((np.float32(1) + np.float32(5)) + fmax) / np.float32(3),
                                           ((np.float32(4) + np.float32(5)) + fmax) / np.float32(3)], np.float32))
    four = oracle.knn_mean_dist2(np.array([[0, 0, 0], [1, 0, 0], [0, 2, 0], [0, 0, 3]], np.float32))
    assert np.allclose(four, [(1 + 4 + 9) / 3, (1 + 5 + 10) / 3, (4 + 5 + 13) / 3, (9 + 10 + 13) / 3])


def test_golden_fixture():
    z = np.load(os.path.join(GOLD, "knn_kdtree64.npz"))
    got = oracle.knn_mean_dist2(z["points"])
    assert np.all(np.abs(got - z["mean_dist2"]) <= 1e-6 * z["mean_dist2"])


def test_drop_in_import_and_loud_failure_off_gpu():
    from simple_knn._C import distCUDA2  # the reference's import line (scene/gaussian_model.py:9)
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        distCUDA2(torch.zeros(8, 3))
