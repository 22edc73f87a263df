"""FusedAdam on the GPU (csrc/adam.hip through goi_adam_step): bit-exact against the numpy oracle,
a few ulp from torch.optim.Adam running on the same device, masked step, ragged sizes."""
import numpy as np
import pytest
import torch

from oracle import oracle
from tests.test_adam_cpu import GROUPS, LRS, make_params, reference_groups

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("P", [1, 3, 255, 1021, 4096])
def test_bit_exact_against_oracle_over_steps(P):
    from goi_hyperplane_amd.optim import FusedAdam
    params = make_params(P, device="cuda")
    opt = FusedAdam(reference_groups(params), lr=0.0, eps=1e-15)
    mine = {k: (v.detach().cpu().numpy().copy(), np.zeros(v.shape, np.float32), np.zeros(v.shape, np.float32))
            for k, v in params.items()}
    rng = np.random.default_rng(P)
    for step in range(1, 6):
        mask = (rng.random(P) < 0.3) if step % 2 == 0 else None
        for k, v in params.items():
            gr = (rng.standard_normal(v.shape) * (10.0 ** rng.integers(-6, 1))).astype(np.float32)
            v.grad = torch.from_numpy(gr).cuda()
            p, m, s = mine[k]
            mine[k] = oracle.adam_step(p, gr, m, s, step, LRS[k], eps=1e-15, nograd_rows=mask)
        opt.step(nograd_mask=None if mask is None else torch.from_numpy(mask).cuda())
        for k, v in params.items():
            st = opt.state[v]
            assert np.array_equal(v.detach().cpu().numpy(), mine[k][0]), (k, step, "param")
            assert np.array_equal(st["exp_avg"].cpu().numpy(), mine[k][1]), (k, step, "exp_avg")
            assert np.array_equal(st["exp_avg_sq"].cpu().numpy(), mine[k][2]), (k, step, "exp_avg_sq")


def test_tracks_torch_adam_on_device():
    from goi_hyperplane_amd.optim import FusedAdam
    P = 20000
    a, b = make_params(P, device="cuda"), make_params(P, device="cuda")
    fused = FusedAdam(reference_groups(a), lr=0.0, eps=1e-15)
    ref = torch.optim.Adam(reference_groups(b), lr=0.0, eps=1e-15)
    g = torch.Generator(device="cuda").manual_seed(3)
    for step in range(6):
        for k in GROUPS:
            gr = torch.randn(a[k].shape, device="cuda", generator=g) * 1e-2
            a[k].grad, b[k].grad = gr.clone(), gr.clone()
        fused.step()
        ref.step()
    for k in GROUPS:
        x, y = a[k].detach(), b[k].detach()
        assert torch.all((x - y).abs() <= 4e-7 * (y.abs() + y.abs().max())), k
        sx, sy = fused.state[a[k]], ref.state[b[k]]
        assert torch.all((sx["exp_avg"] - sy["exp_avg"]).abs() <= 4e-7 * (sy["exp_avg"].abs() + sy["exp_avg"].abs().max()))
        assert torch.all((sx["exp_avg_sq"] - sy["exp_avg_sq"]).abs() <= 4e-7 * (sy["exp_avg_sq"].abs() + sy["exp_avg_sq"].abs().max()))


def test_params_without_grad_are_skipped_and_headline_size_runs():
    from goi_hyperplane_amd.optim import FusedAdam
    P = 1_000_000
    params = make_params(P, device="cuda")
    opt = FusedAdam(reference_groups(params), lr=0.0, eps=1e-15)
    before = params["xyz"].detach().clone()
    params["semantics"].grad = torch.ones_like(params["semantics"])  # the reference's default: semantics only
    opt.step()
    assert torch.equal(params["xyz"].detach(), before) and len(opt.state[params["xyz"]]) == 0
    # first step with g = 1: m = 0.1, v = 0.001 -> p -= lr * 1 (up to rounding)
    assert torch.allclose(params["semantics"].detach() + LRS["semantics"],
                          make_params(P, device="cuda")["semantics"].detach(), atol=1e-6)
