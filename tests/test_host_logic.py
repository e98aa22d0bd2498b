"""Host-side behaviour of the container and modules (no kernel runs here)."""
import os

import numpy as np
import pytest
import torch

from spiral_amd import _native as nv
from spiral_amd.core.errors import DimensionError, GraphError, ValueNotSortedError
from spiral_amd.core.utils import get_anchor_position, is_sorted
from spiral_amd.dtdg.types import CitationGraph, Snapshot, VEInvariantDTDG
from spiral_amd.graph import SnapshotGraph, aggregate

GOLD = os.path.join(os.path.dirname(__file__), "golden")


def test_utils_known_answers():
    fx = np.load(os.path.join(GOLD, "types_fixture.npz"))
    a = fx["anchor_in"]
    assert np.array_equal(get_anchor_position(a, np.unique(a)), fx["anchor_out"])
    assert is_sorted(np.arange(10))
    shuffled = np.array([4, 1, 3, 0, 2])
    assert not is_sorted(shuffled)


def test_container_validation_errors():
    fx = np.load(os.path.join(GOLD, "types_fixture.npz"))
    edges, nodes, times = fx["edges"], fx["nodes"], fx["times"]
    with pytest.raises(DimensionError):
        VEInvariantDTDG(edges[:, :2], nodes, times, device="cpu")
    with pytest.raises(DimensionError):
        VEInvariantDTDG(edges, nodes[:, :0], times, device="cpu")
    with pytest.raises(ValueNotSortedError):
        VEInvariantDTDG(edges[::-1], nodes, times, device="cpu")
    with pytest.raises(ValueNotSortedError):
        VEInvariantDTDG(edges, nodes[::-1], times, device="cpu")
    with pytest.raises(ValueNotSortedError):
        VEInvariantDTDG(edges, nodes, times[::-1], device="cpu")


def test_prefix_dispatch_counts_on_the_reference_fixture():
    fx = np.load(os.path.join(GOLD, "types_fixture.npz"))
    g = VEInvariantDTDG(fx["edges"], fx["nodes"], fx["times"], device="cpu")
    assert len(g) == 3
    for i, t in enumerate(fx["expect_t"]):
        snap, ids = g.dispatcher(int(t))
        assert isinstance(snap, Snapshot) and int(snap.t) == t
        assert snap.observation.num_edges() == fx["expect_num_edges"][i]
        assert snap.observation.num_nodes() == fx["expect_num_nodes"][i]
        assert snap.num_nodes() == ids.numel() and torch.equal(ids, torch.arange(ids.numel()))
        assert snap.num_node_features() == 1
    assert g.edge_id["3_0"] == 2 and len(g.edge_id) == 4


def test_no_cpu_fallback():
    g = SnapshotGraph(torch.tensor([0, 1]), torch.tensor([1, 0]), 2)
    with pytest.raises(nv.SpmiError):
        g.in_degrees()
    with pytest.raises(nv.SpmiError):
        g.csr()
    fx = np.load(os.path.join(GOLD, "types_fixture.npz"))
    cg = CitationGraph(fx["edges"], fx["nodes"], fx["times"], device="cpu")
    cg.dispatcher(0)  # no degree needed at t = 0
    with pytest.raises(nv.SpmiError):
        cg.dispatcher(1)


def test_graphconv_argument_checks():
    from spiral_amd.core.types import GraphConv

    with pytest.raises(GraphError):
        GraphConv(4, 4, norm="sym")
    conv = GraphConv(4, 3, allow_zero_in_degree=True)
    assert conv.weight.shape == (4, 3) and torch.count_nonzero(conv.bias) == 0
    with pytest.raises(GraphError):
        conv("not a graph", torch.zeros(2, 4))


def test_snapshot_graph_surface():
    g = SnapshotGraph(torch.tensor([2, 3, 4]), torch.tensor([1, 2, 3]))
    assert g.num_nodes() == 5 and g.num_edges() == 3 and g.device == torch.device("cpu")
    assert g.to("cpu") is g
    g.ndata["h"] = torch.zeros(5, 2)
    with pytest.raises(ValueError):
        g.ndata["h"] = torch.zeros(4, 2)
    g2 = g.add_self_loop()
    assert g2.num_edges() == 8 and g2.ndata["h"].shape == (5, 2)
    sn = Snapshot(g, 1)
    assert sn.to("cpu") is sn and sn.num_nodes() == 5
