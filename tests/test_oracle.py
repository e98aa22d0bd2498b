"""The CPU oracle against the reference's own fixtures and against itself.

Pinned by the reference: the 5-node CitationGraph fixture and its assertions
(tests/dtdg/test_types_torch.py:12-14,36-56), the anchor known answer
(tests/core/test_utils.py:47-53), and the decoder goldens produced by the reference's own
torch decoders (tests/golden/make_golden.py).  The aggregation has no reference fixture
(parity unpinned): three independent restatements must agree instead."""
import os

import numpy as np
import pytest
import torch

from oracle import oracle as orc

GOLD = os.path.join(os.path.dirname(__file__), "golden")


def test_anchor_known_answer():
    fx = np.load(os.path.join(GOLD, "types_fixture.npz"))
    a = fx["anchor_in"]
    assert np.array_equal(orc.get_anchor_position(a, np.unique(a)), fx["anchor_out"])
    assert orc.is_sorted(np.arange(10)) and not orc.is_sorted(np.array([3, 1, 2]))


def test_citation_fixture_matches_reference_assertions():
    fx = np.load(os.path.join(GOLD, "types_fixture.npz"))
    edges, nodes, times = fx["edges"], fx["nodes"], fx["times"]
    for i, t in enumerate(fx["expect_t"]):
        src, dst, n, feat, _ = orc.citation_dispatch(edges, nodes, len(times), int(t))
        assert len(src) == fx["expect_num_edges"][i]
        assert n == fx["expect_num_nodes"][i]
        assert feat[:, -1].sum() == fx["expect_citation_sum"][i]
        # with one stored feature only the citation column survives the reference's slicing
        assert feat.shape == (n, 1)


def test_csr_is_stable_and_degrees_count_multi_edges():
    src = np.array([3, 1, 3, 0, 1, 3, 3], dtype=np.int64)
    dst = np.array([2, 2, 0, 2, 2, 2, 0], dtype=np.int64)
    indptr, indices, eid = orc.coo_to_csr(dst, src, 4)
    assert indptr.tolist() == [0, 2, 2, 7, 7]
    assert indices.tolist() == [3, 3, 3, 1, 0, 1, 3]
    assert eid.tolist() == [2, 6, 0, 1, 3, 4, 5]
    assert orc.degree(dst, 4).tolist() == [2, 0, 5, 0]
    with pytest.raises(ValueError):
        orc.degree(np.array([4]), 4)
    with pytest.raises(ValueError):
        orc.coo_to_csr(np.array([0]), np.array([-1]), 4)


def test_three_restatements_of_the_aggregation_agree():
    n, e, f = 1000, 8000, 16
    rng = np.random.default_rng(0)
    src, dst = orc.powerlaw_snapshot(n, e, t=0)
    x = rng.standard_normal((n, f)).astype(np.float32)
    indptr, indices, eid = orc.coo_to_csr(dst, src, n)
    w = rng.random(e).astype(np.float32)
    for weights in (None, w):
        a = orc.aggregate(indptr, indices, x, edge_w=None if weights is None else weights[eid])
        b = orc.aggregate_add_at(src, dst, n, x, weights)
        c = orc.aggregate_scipy(src, dst, n, n, x, weights)
        # elementwise against the sum of absolute terms (hub rows add hundreds of values,
        # and scipy merges duplicate edges before multiplying)
        scale = orc.aggregate(indptr, indices, np.abs(x),
                              edge_w=None if weights is None else weights[eid]) + 1e-30
        assert (np.abs(a - b) / scale).max() < 1e-5
        assert (np.abs(a - c) / scale).max() < 1e-5


def test_graphconv_oracle_numpy_vs_torch_and_goldens():
    g = np.load(os.path.join(GOLD, "graphconv.npz"))
    src, dst, x = g["small/src"], g["small/dst"], g["small/x"]
    for fo in (4, 6, 9):
        w, b = g[f"small/w{fo}"], g[f"small/b{fo}"]
        for norm in ("both", "right", "left", "none"):
            y = orc.graphconv_forward(src, dst, 7, x, w, b, norm=norm, act="leaky_relu",
                                      allow_zero_in_degree=True)
            np.testing.assert_allclose(y, g[f"small/y_{norm}_{fo}"], rtol=1e-6, atol=1e-6)
            yt = orc.graphconv_torch(src, dst, 7, torch.tensor(x), torch.tensor(w), torch.tensor(b),
                                     norm=norm, act="leaky_relu").numpy()
            np.testing.assert_allclose(y, yt, rtol=1e-5, atol=1e-6)
    with pytest.raises(ValueError):  # nodes 3..6 have no in-edges
        orc.graphconv_forward(src, dst, 7, x, g["small/w4"], g["small/b4"])
    with pytest.raises(ValueError):
        orc.graphconv_forward(src, dst, 7, x, g["small/w4"], g["small/b4"], norm="sym",
                              allow_zero_in_degree=True)


@pytest.mark.parametrize("kind", ["gru", "lstm", "rnn"])
def test_cell_oracle_matches_reference_decoders(kind):
    """numpy cells + MLP reproduce the outputs and memories of the reference's decoders."""
    d = np.load(os.path.join(GOLD, "decoders.npz"))
    p = lambda name: d[f"{kind}/param/{name}"]
    params = [(p(f"base_model.weight_ih_l{k}"), p(f"base_model.weight_hh_l{k}"),
               p(f"base_model.bias_ih_l{k}"), p(f"base_model.bias_hh_l{k}")) for k in range(2)]
    mlp = []
    for i in (0, 4, 8):
        mlp.append({"w": p(f"simple_decoder.mlp.{i}.weight"), "b": p(f"simple_decoder.mlp.{i}.bias")})
        if i < 8:
            mlp.append({"gamma": p(f"simple_decoder.mlp.{i+1}.weight"),
                        "beta": p(f"simple_decoder.mlp.{i+1}.bias"), "mean": None, "var": None})
    mem_h = d[f"{kind}/mem_h0"].copy()
    mem_c = d[f"{kind}/mem_c0"].copy() if kind == "lstm" else None
    # step 0 only: later steps use parameters the reference has not updated either (no
    # optimizer step in the generator), but BatchNorm running stats do not enter train mode
    for step, n in enumerate((17, 23)):
        x = d[f"{kind}/step{step}/x"]
        if kind == "lstm":
            out, h, c = orc.recurrent_step(kind, x, mem_h[:, :n], params, mem_c[:, :n])
            mem_c[:, :n] = c
        else:
            out, h = orc.recurrent_step(kind, x, mem_h[:, :n], params)
        mem_h[:, :n] = h
        y = orc.mlp_forward(out, mlp, training=True)
        np.testing.assert_allclose(mem_h, d[f"{kind}/step{step}/mem_h"], rtol=1e-5, atol=1e-6)
        np.testing.assert_allclose(y, d[f"{kind}/step{step}/y"], rtol=1e-4, atol=1e-5)
