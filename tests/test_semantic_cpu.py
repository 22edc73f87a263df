"""CPU pins of the semantic head against vectors produced by the REFERENCE's own code: its
SemanticModel class and checkpoint format, its GUI.compute_similarity method (gui/main.py:362-384) with its own LinearSVM
(networks.py:12-59), and the loss statements of its training loop (train.py:142-163) -- all executed from the reference's
AST by tests/golden/make_golden.py:semantic_pins, nothing re-typed."""
import os

import numpy as np
import torch

from goi_hyperplane_amd.semantic import (LinearSVM, SemanticModel, codebook_losses, compute_similarity_reference,
                                         svm_score_fn)

GOLD = os.path.join(os.path.dirname(__file__), "golden")


def _load():
    pins = np.load(os.path.join(GOLD, "ref_semantic_pins.npz"))
    mlp = SemanticModel.load(os.path.join(GOLD, "ref_semantic_mlp.pt"), map_location="cpu")  # written by the reference class
    return pins, mlp


def _svm(pins):
    svm = LinearSVM()
    with torch.no_grad():
        svm.linear.weight.copy_(torch.tensor(pins["svm_w"]))
        svm.linear.bias.copy_(torch.tensor(pins["svm_b"]))
    return svm


def test_reference_checkpoint_loads_and_matches():
    pins, mlp = _load()
    assert mlp.args["dim_in"] == 10 and mlp.args["dim_out"] == 300 and mlp.num_layer == 1
    with torch.no_grad():
        dec = mlp(torch.tensor(pins["feats"]))
    np.testing.assert_allclose(dec.numpy(), pins["dec"], rtol=1e-5, atol=1e-6)


def test_decode_restatement_matches_reference_lines():
    pins, mlp = _load()
    bg = torch.zeros(pins["feats"].shape[0], dtype=torch.bool)
    sim, idx = compute_similarity_reference(torch.tensor(pins["feats"]), mlp, torch.tensor(pins["lut"]),
                                            svm_score_fn(_svm(pins)), 0.5, out_bg_mask=bg)
    assert (idx.numpy() == pins["idx"]).all()
    assert (bg.numpy() == pins["bg"]).all()
    np.testing.assert_allclose(sim.numpy(), pins["sim"], rtol=1e-5, atol=1e-6)


import pytest


@pytest.mark.parametrize("tag,iteration", [("", 10), ("_t2", 1500)])  # anneal factor t = 1 / 2 (train.py:156)
def test_training_losses_match_reference_lines(tag, iteration):
    pins, mlp = _load()
    assert tuple(pins["ref_lines"]) == (142, 163)  # the statements the vectors were produced by
    S, H, W = 10, 24, 16
    f = torch.tensor(pins["feats"]).T.reshape(S, H, W).clone().requires_grad_(True)
    lut = torch.tensor(pins["lut"]).clone().requires_grad_(True)
    gtl = torch.tensor(pins["gtl"]).T.reshape(256, H, W)
    loss, terms = codebook_losses(f, mlp, lut, gtl, iteration=iteration)
    loss.backward()
    assert abs(loss.item() - float(pins["loss" + tag])) < 1e-5
    np.testing.assert_allclose([terms[k].item() for k in ("lab", "sl", "sl1", "recc")], pins["terms" + tag], rtol=1e-5,
                               atol=1e-6)
    gf = f.grad.reshape(S, -1).T.numpy()
    np.testing.assert_allclose(gf, pins["grad_feats" + tag], rtol=1e-4, atol=1e-8)
    np.testing.assert_allclose(lut.grad.numpy(), pins["grad_lut" + tag], rtol=1e-4, atol=1e-8)
