"""Fused Adam on the CPU: the numpy oracle against torch.optim.Adam itself --
the class the reference instantiates (scene/gaussian_model.py:180) -- and the checkpoint format."""
import numpy as np
import pytest
import torch

from oracle import oracle

GROUPS = {"xyz": (3,), "f_dc": (1, 3), "f_rest": (15, 3), "semantics": (16,), "opacity": (1,), "scaling": (3,),
          "rotation": (4,)}
LRS = {"xyz": 1.6e-4, "f_dc": 2.5e-3, "f_rest": 1.25e-4, "semantics": 5e-3, "opacity": 5e-2, "scaling": 5e-3,
       "rotation": 1e-3}


def make_params(P, seed=0, device="cpu"):
    g = torch.Generator().manual_seed(seed)
    return {k: torch.nn.Parameter(torch.randn((P,) + shp, generator=g).to(device)) for k, shp in GROUPS.items()}


def reference_groups(params):
    return [{"params": [params[k]], "lr": LRS[k], "name": k} for k in GROUPS]


def test_oracle_follows_torch_adam_over_steps():
    P = 257
    params = make_params(P)
    opt = torch.optim.Adam(reference_groups(params), lr=0.0, eps=1e-15)
    mine = {k: (v.detach().numpy().copy(), np.zeros(v.shape, np.float32), np.zeros(v.shape, np.float32))
            for k, v in params.items()}
    rng = np.random.default_rng(0)
    for step in range(1, 8):
        for k, v in params.items():
            gr = (rng.standard_normal(v.shape) * (10.0 ** rng.integers(-6, 1))).astype(np.float32)
            if step == 3:
                gr[: P // 2] = 0  # rows that were not visible in this view
            v.grad = torch.from_numpy(gr.copy())
            p, m, s = mine[k]
            mine[k] = oracle.adam_step(p, gr, m, s, step, LRS[k], eps=1e-15)
        opt.step()
        for k, v in params.items():
            st = opt.state[v]
            for got, ref in ((mine[k][0], v.detach().numpy()), (mine[k][1], st["exp_avg"].numpy()),
                             (mine[k][2], st["exp_avg_sq"].numpy())):
                # same fp32 operations; the CPU kernels fuse a*b+c (lerp) and order alpha*t1/t2 differently:
                # a few ulp of the operands' magnitude (cancellation in g - m makes that an ABSOLUTE bound)
                assert np.all(np.abs(got - ref) <= 4e-7 * (np.abs(ref) + np.abs(ref).max())), (k, step)


def test_masked_rows_behave_like_zero_gradient():
    P = 64
    rng = np.random.default_rng(1)
    p, g = rng.standard_normal((P, 3)).astype(np.float32), rng.standard_normal((P, 3)).astype(np.float32)
    m, v = rng.standard_normal((P, 3)).astype(np.float32) * 0.1, rng.random((P, 3)).astype(np.float32)
    mask = rng.random(P) < 0.4
    a = oracle.adam_step(p, g, m, v, 5, 1e-2, nograd_rows=mask)
    g0 = g.copy()
    g0[mask] = 0
    b = oracle.adam_step(p, g0, m, v, 5, 1e-2)
    for x, y in zip(a, b):
        assert np.array_equal(x, y)
    assert not np.array_equal(a[0][mask], p[mask])  # momentum still moves masked rows, as in the reference


def test_state_dict_round_trips_with_torch_adam():
    """capture()/restore() of the reference pickle optimizer.state_dict(): both classes must accept
    each other's."""
    from goi_hyperplane_amd.optim import FusedAdam
    params = make_params(10)
    ref = torch.optim.Adam(reference_groups(params), lr=0.0, eps=1e-15)
    for v in params.values():
        v.grad = torch.ones_like(v)
    ref.step()
    sd = ref.state_dict()
    fused = FusedAdam(reference_groups(params), lr=0.0, eps=1e-15)
    fused.load_state_dict(sd)
    assert [g["name"] for g in fused.param_groups] == list(GROUPS)
    st = fused.state[params["xyz"]]
    assert float(st["step"]) == 1.0 and st["exp_avg"].shape == params["xyz"].shape
    back = torch.optim.Adam(reference_groups(params), lr=0.0, eps=1e-15)
    back.load_state_dict(fused.state_dict())
    assert set(sd["param_groups"][0]) == set(fused.state_dict()["param_groups"][0])


def test_loud_failure_off_gpu():
    from goi_hyperplane_amd.optim import FusedAdam
    params = make_params(4)
    opt = FusedAdam(reference_groups(params), lr=0.0, eps=1e-15)
    for v in params.values():
        v.grad = torch.ones_like(v)
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        opt.step()
    with pytest.raises(NotImplementedError):
        FusedAdam(reference_groups(params), weight_decay=0.1)
