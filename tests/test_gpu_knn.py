"""distCUDA2 on the GPU (csrc/knn.hip through goi_knn_dist2) against the CPU oracle.
Bar: bit-exact against the oracle in the contraction form the kernel spells out; within 1 ulp-scale
relative error of the un-contracted form (the reference's nvcc build may use either)."""
import numpy as np
import pytest
import torch

from oracle import oracle
from tests.test_knn_cpu import clouds

pytestmark = pytest.mark.gpu


def gpu_knn(pts):
    from simple_knn._C import distCUDA2
    return distCUDA2(torch.from_numpy(np.ascontiguousarray(pts)).float().cuda()).cpu().numpy()


@pytest.mark.parametrize("name", list(clouds()))
def test_matches_oracle_bit_exact(name):
    pts = clouds()[name]
    got = gpu_knn(pts)
    assert np.array_equal(got, oracle.knn_mean_dist2(pts, fma=True)), name
    plain = oracle.knn_mean_dist2(pts, fma=False)
    assert np.all(np.abs(got - plain) <= 4e-7 * np.maximum(plain, np.float32(1e-30)) + np.float32(1e-12))


@pytest.mark.parametrize("P", [0, 1, 2, 3, 4, 5, 63, 64, 255, 256, 257, 511, 513, 1024, 1025])
def test_ragged_sizes(P):
    rng = np.random.default_rng(P)
    pts = rng.standard_normal((P, 3)).astype(np.float32)
    got = gpu_knn(pts)
    assert got.shape == (P,)
    assert np.array_equal(got, oracle.knn_mean_dist2(pts, fma=True))


def test_input_order_invariance_and_noncontiguous_input():
    from simple_knn._C import distCUDA2
    rng = np.random.default_rng(3)
    pts = rng.standard_normal((20000, 3)).astype(np.float32)
    perm = rng.permutation(len(pts))
    a = gpu_knn(pts)
    b = gpu_knn(pts[perm])
    assert np.array_equal(a[perm], b)
    wide = torch.from_numpy(np.concatenate([pts, pts], 1)).cuda()[:, :3]  # strided view: .contiguous() path
    assert np.array_equal(distCUDA2(wide).cpu().numpy(), a)


def test_scene_scale_cloud_against_oracle():
    """SfM-sized input (the call site feeds the COLMAP cloud, scene/gaussian_model.py:147): 1M points."""
    rng = np.random.default_rng(5)
    c = rng.standard_normal((2000, 3)) * 5
    pts = (c[rng.integers(0, 2000, 1_000_000)] + 0.3 * rng.standard_normal((1_000_000, 3))).astype(np.float32)
    got = gpu_knn(pts)
    assert np.isfinite(got).all() and (got >= 0).all()
    sub = rng.choice(len(pts), 200_000, replace=False)  # exact k-NN of a subset is NOT the subset of the k-NN:
    ref = oracle.knn_mean_dist2(pts, fma=True)          # so run the whole cloud through the pruned oracle
    assert np.array_equal(got[sub], ref[sub]) and np.array_equal(got, ref)


def test_rejects_wrong_dtype():
    from simple_knn._C import distCUDA2
    with pytest.raises(TypeError):
        distCUDA2(torch.zeros(10, 3, dtype=torch.float64, device="cuda"))
