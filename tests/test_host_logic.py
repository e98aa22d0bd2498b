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


# ---- output head: BatchNorm folding and the packed layout (host logic, no GPU) ------------
def _mlp_with_stats(dims, out, act, seed):
    import torch
    from spiral_amd.dtdg.models.decoder.torch.simpleDecoder import MLP

    torch.manual_seed(seed)
    mlp = MLP(32, dims, output_dim=out, activation=act).eval()
    gen = torch.Generator().manual_seed(seed + 1)
    with torch.no_grad():
        for m in mlp.modules():
            if isinstance(m, torch.nn.BatchNorm1d):
                m.running_mean.copy_(torch.randn(m.num_features, generator=gen))
                m.running_var.copy_(torch.rand(m.num_features, generator=gen) + 0.5)
                m.weight.copy_(torch.randn(m.num_features, generator=gen))
                m.bias.copy_(torch.randn(m.num_features, generator=gen))
    return mlp


def test_folded_head_layers_equal_the_module_in_eval_mode():
    import torch

    for dims, out, act in (([10, 20, 10, 5], 1, "linear"), ([7], 3, "sigmoid"), ([], 4, "tanh"), ([8, 8], 2, "relu")):
        mlp = _mlp_with_stats(dims, out, act, seed=len(dims))
        x = torch.randn(50, 32)
        layers = mlp._folded_layers()
        assert len(layers) == len(dims) + 1
        h = x
        for w, b, relu in layers:
            h = h @ w.t() + (0 if b is None else b)
            if relu:
                h = torch.relu(h)
        fin = {None: lambda t: t, "sigmoid": torch.sigmoid, "relu": torch.relu, "tanh": torch.tanh}[mlp._final_act]
        with torch.no_grad():
            want = mlp.mlp(x)
        assert torch.allclose(fin(h), want, rtol=1e-5, atol=1e-6)
        assert [r for _, _, r in layers] == [True] * len(dims) + [False]
        # cached until a parameter or a running statistic changes in place
        assert mlp._folded_layers() is layers
        with torch.no_grad():
            mlp.layers[0].weight.add_(1.0)
        assert mlp._folded_layers() is not layers


def test_packed_head_layout():
    import torch
    from spiral_amd import ops

    mlp = _mlp_with_stats([10, 20], 3, "linear", seed=5)
    layers = mlp._folded_layers()
    w1t, b1p, out1, relu1, rest, outs, relus, n_rest, width = ops.pack_mlp_head(layers, "cpu")[:9]
    assert (out1, relu1, n_rest, width) == (10, 1, 2, 3)
    assert w1t.shape == (32, 16) and b1p.shape == (16,)            # 10 outputs padded to 16
    assert torch.equal(w1t[:, :10], layers[0][0].t()) and not w1t[:, 10:].any()
    assert torch.equal(b1p[:10], layers[0][1]) and not b1p[10:].any()
    w2, b2, _ = layers[1]
    w3, b3, _ = layers[2]
    assert torch.equal(rest, torch.cat([w2.reshape(-1), b2, w3.reshape(-1), b3]))
    assert ops.mlp_head_supported(128, [10, 20, 10, 5, 1])
    assert not ops.mlp_head_supported(128, [64, 1])                 # wider than the kernel holds
    assert not ops.mlp_head_supported(100, [10, 1])                 # input not a multiple of 32
    assert not ops.mlp_head_supported(1024, [32, 1])                # layer-1 weight beyond LDS budget


def test_node_memory_spare_table_and_swap():
    """The second table of a NodeMemory (used by steps that write every node's new state beside
    the old one): same shape, created once, trades places with the memory on swap_spare(), and
    is dropped by reset_state() and by a move to another device."""
    import torch

    from spiral_amd.dtdg.models.decoder.torch.sequentialDecoder.implicitTimeModels import NodeMemory

    mem = NodeMemory(7, 4, 2)
    old = mem.memory
    sp = mem.spare()
    assert sp.shape == old.shape and sp.dtype == old.dtype and sp.data_ptr() != old.data_ptr()
    assert mem.spare() is sp                                   # allocated once
    sp.fill_(3.0)
    mem.swap_spare()
    assert mem.memory is sp and mem.spare() is old
    assert torch.equal(mem.get_memory(torch.tensor([0, 6])), torch.full((2, 2, 4), 3.0))
    mem.update_memory(torch.zeros(2, 2, 4), torch.tensor([1, 2]))
    assert mem.memory[:, 1:3].abs().sum() == 0 and mem.memory[:, 0].eq(3.0).all()
    mem.reset_state()
    assert mem._spare is None and mem.memory.shape == (2, 7, 4)
    moved = mem.to("meta")
    assert moved._spare is None and moved.memory.device.type == "meta"


def test_node_memory_adopts_layer_matrices_without_aliasing_the_table():
    """set_layers(): the state of every node handed over as one matrix per layer (outputs of
    tracked steps) is read back through layer(k) as it is, and the [L, N, H] table the first
    reader asks for is a copy of its own: writing the table (update_memory, in-place edits)
    never reaches the adopted matrices, which an autograd graph may still hold."""
    import torch

    from spiral_amd.dtdg.models.decoder.torch.sequentialDecoder.implicitTimeModels import NodeMemory

    mem = NodeMemory(5, 3, 2)
    assert mem.rows() == 5 and mem.layer(1).data_ptr() == mem.memory[1].data_ptr()
    a = torch.full((5, 3), 1.0, requires_grad=True) * 2.0       # non-leaf, as a step's output is
    b = torch.full((5, 3), 7.0)
    mem.set_layers([a, b])
    assert mem.rows() == 5 and mem.device == a.device
    assert mem.layer(0).data_ptr() == a.data_ptr() and not mem.layer(0).requires_grad
    assert mem.layer(1).data_ptr() == b.data_ptr()
    assert mem.bounded()                                        # boundedness carries over
    table = mem.memory                                          # first reader: stacked copy
    assert table.shape == (2, 5, 3) and table.data_ptr() not in (a.data_ptr(), b.data_ptr())
    assert mem.memory is table and mem.layer(0).data_ptr() == table[0].data_ptr()
    mem.update_memory(torch.zeros(2, 2, 3), torch.tensor([0, 4]))
    assert float(a.detach().min()) == 2.0 and float(b.min()) == 7.0   # the adopted matrices are untouched
    assert mem.memory[:, 0].abs().sum() == 0 and mem.memory[0, 1].eq(2.0).all()
    mem.set_layers([a, b])
    mem.memory = torch.ones(2, 5, 3)                            # assignment drops the adopted matrices
    assert mem.layer(0).eq(1.0).all()
    mem.set_layers([a, b])
    moved = mem.to("meta")
    assert moved.memory.shape == (2, 5, 3) and moved.device.type == "meta"


def test_mrr_definition():
    """Reciprocal rank of each positive among its own negatives (the metric the reference takes
    from ogb's Evaluator, examples/tf/link_prediction_sliding_window.py:142-146): hand-computed
    cases, ties counted half, and the script's layout of one evaluation pass."""
    import torch

    from spiral_amd.automl.evaluate import link_prediction_mrr, mrr

    pos = torch.tensor([0.9, 0.2, 0.5])
    neg = torch.tensor([[0.1, 0.3, 0.5], [0.7, 0.3, 0.1], [0.5, 0.5, 0.9]])
    out = mrr(pos, neg)
    # ranks: 1 (nothing above); 3 (two above); 1 above + 2 ties -> 0.5 * (1 + 3) + 1 = 3
    assert torch.allclose(out["mrr_list"], torch.tensor([1.0, 1 / 3, 1 / 3]))
    assert out["hits@1_list"].tolist() == [1.0, 0.0, 0.0] and out["hits@3_list"].tolist() == [1.0, 1.0, 1.0]
    # without ties the rank is the position in the descending sort (ogb's argsort form)
    gen = torch.Generator().manual_seed(0)
    p, n = torch.rand(50, generator=gen), torch.rand(50, 20, generator=gen)
    order = torch.argsort(torch.cat([p[:, None], n], 1), dim=1, descending=True)
    want = 1.0 / ((order == 0).nonzero()[:, 1] + 1).float()
    assert torch.allclose(mrr(p, n)["mrr_list"], want)
    # two evaluated nodes, two positives each, three negatives per node
    scores = torch.tensor([0.9, 0.4, 0.8, 0.1, 0.5, 0.3, 0.2, 0.6, 0.7, 0.0])
    got = link_prediction_mrr(scores, n_pos=4, negatives_per_node=3)
    want = (1 / 1 + 1 / 2 + 1 / 1 + 1 / 3) / 4   # node 0 negatives .5 .3 .2, node 1 negatives .6 .7 0
    assert abs(got - want) < 1e-6
    import pytest
    with pytest.raises(ValueError):
        mrr(torch.rand(3), torch.rand(4, 2))
