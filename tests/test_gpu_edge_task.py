"""Edge task (link prediction) on the GPU: the head the reference implements only for its
TensorFlow backend (tf/sequentialDecoder/baseModel.py:32-46,154-164, tf/simpleDecoder.py:90-97,
automl/tf/prepare_dataset.py:37-94), here on the torch path.  Parity against the CPU
oracle (numpy cells + MLP) and a differentiable torch restatement; the reference holds no
numeric fixture for it (parity unpinned)."""
import copy

import numpy as np
import pytest
import torch
import torch.nn.functional as F

from oracle import oracle as orc
from spiral_amd.automl.model_assembler import assembler
from spiral_amd.automl.prepare_dataset import LABEL, TARGET, prepare_edge_task
from spiral_amd.core.errors import DimensionError
from spiral_amd.dtdg.models.decoder.torch.sequentialDecoder.implicitTimeModels import GRU, LSTM
from spiral_amd.dtdg.models.decoder.torch.simpleDecoder import MLP
from spiral_amd.dtdg.models.encoder.implicitTimeEncoder.staticGraphEncoder import GCN
from spiral_amd.dtdg.types import VEInvariantDTDG
from spiral_amd.graph import SnapshotGraph

pytestmark = pytest.mark.gpu


def rel(a, b):
    a, b = np.asarray(a, dtype=np.float64), np.asarray(b, dtype=np.float64)
    return np.abs(a - b).max() / max(np.abs(b).max(), 1e-30)


def _dtdg(dev, seed=0, n=600, e=9000, steps=5, f=8):
    rng = np.random.default_rng(seed)
    node_t = np.sort(rng.integers(0, steps, n))
    nodes = np.hstack([node_t[:, None], rng.standard_normal((n, f))])
    a, b = rng.integers(0, n, e), rng.integers(0, n, e)
    keep = a != b
    s, d = np.maximum(a, b)[keep], np.minimum(a, b)[keep]
    _, first = np.unique(s * n + d, return_index=True)  # the edge-id dict needs distinct pairs
    s, d = s[np.sort(first)], d[np.sort(first)]
    et = node_t[s]
    edges = np.stack([et, s, d], 1)[np.argsort(et, kind="stable")].astype(np.float64)
    return VEInvariantDTDG(edges, nodes, np.arange(steps), device=dev), edges, nodes


def test_remove_edges_successors(dev):
    g = SnapshotGraph(torch.tensor([0, 0, 1, 2, 2], device=dev), torch.tensor([1, 2, 2, 0, 1], device=dev), 3)
    g.edata["h"] = torch.arange(5.0, device=dev)[:, None]
    assert g.successors(0).tolist() == [1, 2] and g.predecessors(2).tolist() == [0, 1]
    assert g.in_degrees().tolist() == [1, 2, 2]
    g.remove_edges(torch.tensor([1, 3]))
    assert g.num_edges() == 3 and g.edges()[0].tolist() == [0, 1, 2] and g.edges()[1].tolist() == [1, 2, 1]
    assert g.edata["h"][:, 0].tolist() == [0.0, 2.0, 4.0]
    assert g.in_degrees().tolist() == [0, 2, 1] and g.successors(0).tolist() == [1]


def test_prepare_edge_task_targets_are_consistent(dev):
    graph, edges, _ = _dtdg(dev)
    np.random.seed(0)
    prepare_edge_task(graph, 2, num_negative=5, start_t=2)
    for t in range(2, len(graph)):
        target, label = graph.time_data[TARGET][t], graph.time_data[LABEL][t]
        pos_e = graph.time_data["positive_edges"][t]
        n_pos = int(label.sum())
        assert target.shape == (len(label), 2) and n_pos == len(pos_e) and n_pos > 0
        src, dst, n, _, _ = orc.dispatch(edges, graph.nodes, len(graph), t)
        have = set(zip(src.tolist(), dst.tolist()))
        for (u, v), y in zip(target.tolist(), label.tolist()):
            assert ((u, v) in have) == (y == 1.0) and 0 <= v < n
        assert np.array_equal(edges[pos_e.numpy(), 1:3].astype(np.int64), target[:n_pos].numpy())


@pytest.mark.parametrize("kind", ["gru", "lstm"])
def test_edge_head_matches_oracle_and_torch_restatement(dev, kind):
    graph, edges, nodes = _dtdg(dev, seed=1)
    n_nodes, f, hid = nodes.shape[0], nodes.shape[1] - 1, 32
    np.random.seed(1)
    prepare_edge_task(graph, 2, num_negative=4, start_t=3)
    t = 3
    snap, _ = graph.dispatcher(t)
    target, label = graph.time_data[TARGET][t], graph.time_data[LABEL][t].float()
    snap.observation.remove_edges(graph.time_data["positive_edges"][t])

    torch.manual_seed(0)
    gnn = GCN(2, f, 32, activation=F.leaky_relu, allow_zero_in_degree=True)
    cls = {"gru": GRU, "lstm": LSTM}[kind]
    head = MLP(2 * hid, [16, 8], dropout=0.0, activation="sigmoid")
    rnn = cls(32, hid, n_nodes, 2, head)
    model = assembler(gnn, rnn)
    ref_rnn, ref_mlp = copy.deepcopy(rnn.base_model), copy.deepcopy(head)
    convs = [(l.weight.detach().clone().requires_grad_(), l.bias.detach().clone().requires_grad_())
             for l in gnn.layers]
    mem_h0 = rnn.memory_h.memory.clone()
    mem_c0 = rnn.memory_c.memory.clone() if kind == "lstm" else None
    model.to(dev)
    rnn.to(dev)
    model.eval()  # BatchNorm running statistics: the oracle MLP is the eval-mode one
    ref_mlp.eval()

    predict = model((snap, target))
    loss = F.binary_cross_entropy(predict.squeeze(-1), label.to(dev))
    loss.backward()

    # differentiable CPU restatement
    src, dst = (x.cpu().numpy() for x in snap.observation.edges())
    n = snap.num_nodes()
    h = snap.node_feature().cpu()
    for w, b in convs:
        h = orc.graphconv_torch(src, dst, n, h, w, b, act="leaky_relu")
    uniq, inv = torch.unique(target.reshape(-1), return_inverse=True)
    state = mem_h0[:, uniq].contiguous()
    if kind == "lstm":
        out, (hn, cn) = ref_rnn(h[uniq][:, None, :], (state, mem_c0[:, uniq].contiguous()))
    else:
        out, hn = ref_rnn(h[uniq][:, None, :], state)
    p_ref = ref_mlp((out[:, 0, :], inv.reshape(-1, 2)))
    loss_ref = F.binary_cross_entropy(p_ref.squeeze(-1), label)
    loss_ref.backward()

    assert predict.shape == (target.shape[0], 1)
    assert rel(predict.detach().cpu(), p_ref.detach()) < 1e-5
    assert abs(loss.item() - loss_ref.item()) / loss_ref.item() < 1e-5
    want_mem = mem_h0.clone()
    want_mem[:, uniq] = hn.detach()
    assert rel(rnn.memory_h.memory.cpu(), want_mem) < 1e-5
    for layer, (w, b) in zip(gnn.layers, convs):
        assert rel(layer.weight.grad.cpu(), w.grad) < 1e-4
    for (name, p), q in zip(rnn.base_model.named_parameters(), ref_rnn.parameters()):
        assert rel(p.grad.cpu(), q.grad) < 1e-4, name

    # numpy oracle of the head on the same embeddings
    params = [tuple(getattr(ref_rnn, f"{nm}_l{k}").detach().numpy() for nm in
                    ("weight_ih", "weight_hh", "bias_ih", "bias_hh")) for k in range(2)]
    mlp_layers = []
    for m in ref_mlp.mlp:
        if isinstance(m, torch.nn.Linear):
            mlp_layers.append({"w": m.weight.detach().numpy(), "b": m.bias.detach().numpy()})
        elif isinstance(m, torch.nn.BatchNorm1d):
            mlp_layers.append({"gamma": m.weight.detach().numpy(), "beta": m.bias.detach().numpy(),
                               "mean": m.running_mean.numpy(), "var": m.running_var.numpy()})
    got = orc.edge_head_forward(kind, h.detach().numpy(), target.numpy(), mem_h0.numpy(), params,
                                mlp_layers, None if mem_c0 is None else mem_c0.numpy())
    assert rel(got[0], predict.detach().cpu()) < 1e-5

    with pytest.raises(DimensionError):
        head((out[:, 0, :].detach(), torch.zeros(2, 2, 2, dtype=torch.long)))


def test_prepare_citation_task_against_the_loop_form(dev):
    """Targets and labels of the citation task (spiro/automl/tf/prepare_dataset.py:97-122) for
    the container of the reference's own fixture (tests/automl/batch_generator_test.py:21-23,
    validating_snapshots=1, minimum_citation=0) and for a random table."""
    from spiral_amd.automl.prepare_dataset import LABEL, TARGET, prepare_citation_task
    from spiral_amd.dtdg.types import CitationGraph

    edges = np.array([[0, 0, 1], [0, 0, 2], [1, 3, 0], [2, 4, 1], [2, 4, 2]])
    nodes = np.array([[0, 0], [0, 1], [0, 2], [1, 3], [2, 4], [3, 9]])
    times = np.array([2006, 2007, 2008, 2009])
    cases = [(edges, nodes, times, dict(start_t=1, validating_snapshots=1, minimum_citation=0))]
    rng = np.random.default_rng(4)
    n, e, steps = 400, 5000, 12
    node_t = np.sort(rng.integers(0, steps, n))
    a, b = rng.integers(0, n, e), rng.integers(0, n, e)
    s, d = np.maximum(a, b), np.minimum(a, b)
    big_e = np.stack([node_t[s], s, d], 1)[np.argsort(node_t[s], kind="stable")]
    big_n = np.hstack([node_t[:, None], rng.integers(0, 5, (n, 3))])
    cases.append((big_e, big_n, np.arange(steps), dict(start_t=2, validating_snapshots=3, minimum_citation=4)))
    for ed, no, ti, kw in cases:
        graph = CitationGraph(ed.astype(np.float64), no.astype(np.float64), ti, device=dev)
        prepare_citation_task(graph, **kw)
        want_t, want_l = orc.prepare_citation_task(ed.astype(np.float64), no.astype(np.float64), len(ti), **kw)
        assert sorted(graph.time_data[TARGET]) == sorted(want_t)
        for t in want_t:
            got_t, got_l = graph.time_data[TARGET][t], graph.time_data[LABEL][t]
            assert got_t.dtype == torch.int32 and got_t.device.type == "cuda"
            np.testing.assert_array_equal(got_t.cpu().numpy(), want_t[t])
            np.testing.assert_array_equal(got_l.cpu().numpy(), want_l[t].astype(np.float32))
