"""The reference's training loop (examples/torch/citation_prediction.py:88-102) on a small
synthetic citation DTDG: GPU path vs a CPU restatement built from the oracle's GraphConv
and torch.nn.GRU/LSTM with the same parameters and node memories.  Losses over three
snapshot steps (SGD, so that rounding differences are not amplified the way Adam's
sign-like first steps amplify them) at 1e-4 relative, first-step loss at 1e-5."""
import copy

import numpy as np
import pytest
import torch
import torch.nn.functional as F

from oracle import oracle as orc
from spiral_amd.automl.model_assembler import assembler
from spiral_amd.dtdg.models.decoder.torch.sequentialDecoder.implicitTimeModels import GRU, LSTM
from spiral_amd.dtdg.models.decoder.torch.simpleDecoder import MLP
from spiral_amd.dtdg.models.encoder.implicitTimeEncoder.staticGraphEncoder import GCN
from spiral_amd.dtdg.types import CitationGraph

pytestmark = pytest.mark.gpu


def _dtdg(seed=0, n=1500, e=12000, steps=7, f=6):
    rng = np.random.default_rng(seed)
    node_t = np.sort(rng.integers(0, steps, n))
    nodes = np.hstack([node_t[:, None], rng.standard_normal((n, f))])
    a, b = rng.integers(0, n, e), rng.integers(0, n, e)
    s, d = np.maximum(a, b), np.minimum(a, b)
    et = node_t[s]
    edges = np.stack([et, s, d], 1)[np.argsort(et, kind="stable")].astype(np.float64)
    return edges, nodes, np.arange(steps)


class CpuReference:
    """encoder = oracle GraphConv x2 (leaky_relu), decoder = torch recurrent module + MLP,
    node memory handled as SequentialDecoder.forward does (detached write-back)."""

    def __init__(self, model, kind, n_nodes):
        self.kind = kind
        self.convs = [(l.weight.detach().cpu().clone().requires_grad_(),
                       l.bias.detach().cpu().clone().requires_grad_()) for l in model.encoder.layers]
        self.rnn = copy.deepcopy(model.decoder.base_model).cpu()
        self.mlp = copy.deepcopy(model.decoder.simple_decoder).cpu()
        self.mem_h = model.decoder.memory_h.memory.detach().cpu().clone()
        self.mem_c = model.decoder.memory_c.memory.detach().cpu().clone() if kind == "lstm" else None
        self.params = [p for wb in self.convs for p in wb] + list(self.rnn.parameters()) + \
            list(self.mlp.parameters())

    def forward(self, src, dst, n, feat):
        h = torch.from_numpy(feat)
        for w, b in self.convs:
            h = orc.graphconv_torch(src, dst, n, h, w, b, act="leaky_relu")
        if self.kind == "lstm":
            out, (hn, cn) = self.rnn(h[:, None, :], (self.mem_h[:, :n].contiguous(),
                                                    self.mem_c[:, :n].contiguous()))
            self.mem_c[:, :n] = cn.detach()
        else:
            out, hn = self.rnn(h[:, None, :], self.mem_h[:, :n].contiguous())
        self.mem_h[:, :n] = hn.detach()
        return self.mlp((out[:, 0, :], None))


@pytest.mark.parametrize("kind", ["gru", "lstm"])
def test_training_loop_matches_cpu_restatement(dev, kind):
    edges, nodes, times = _dtdg()
    steps, n_nodes, f = len(times), nodes.shape[0], nodes.shape[1] - 1
    graph = CitationGraph(edges, nodes, times, device=dev)
    torch.manual_seed(0)
    gnn = GCN(2, f, 16, activation=F.leaky_relu, allow_zero_in_degree=True)
    cls = {"gru": GRU, "lstm": LSTM}[kind]
    rnn = cls(16, 10, n_nodes, 2, MLP(10, [10, 20, 10, 5], dropout=0.0))
    model = assembler(gnn, rnn)
    ref = CpuReference(model, kind, n_nodes)
    model.to(dev)  # the assembler moves the decoder's node memories too
    assert model.decoder.memory_h.memory.device.type == "cuda"
    model.train()
    ref.rnn.train()
    ref.mlp.train()
    opt = torch.optim.SGD(model.parameters(), lr=0.01)
    opt_ref = torch.optim.SGD(ref.params, lr=0.01)
    losses, losses_ref = [], []
    for t in range(2, 5):
        snap, ids = graph.dispatcher(t)
        nxt, _ = graph.dispatcher(t + 1)
        opt.zero_grad()
        predict = model((snap, torch.arange(snap.num_nodes())))
        label = nxt.node_feature()[: snap.num_nodes(), -1].float()
        loss = F.mse_loss(predict.squeeze(), label)
        loss.backward()
        losses.append(loss.item())

        src, dst, n, feat, _ = orc.citation_dispatch(edges, nodes, steps, t)
        _, _, _, feat_next, _ = orc.citation_dispatch(edges, nodes, steps, t + 1)
        opt_ref.zero_grad()
        p_ref = ref.forward(src, dst, n, feat)
        loss_ref = F.mse_loss(p_ref.squeeze(), torch.from_numpy(feat_next[:n, -1]))
        loss_ref.backward()
        if t == 2:  # identical parameters and state: every gradient at the fp32 bar
            mine = [p for l in model.encoder.layers for p in (l.weight, l.bias)] + \
                list(model.decoder.base_model.parameters()) + list(model.decoder.simple_decoder.parameters())
            from tests.test_gpu_reference_loop import _check_first_step_gradients

            _check_first_step_gradients(mine, ref.params, ref.mlp)
        opt.step()
        opt_ref.step()
        losses_ref.append(loss_ref.item())
    assert abs(losses[0] - losses_ref[0]) / abs(losses_ref[0]) < 1e-5
    for a, b in zip(losses, losses_ref):
        assert abs(a - b) / abs(b) < 1e-4, (losses, losses_ref)
    mem = model.decoder.memory_h.memory.cpu()
    assert (mem - ref.mem_h).abs().max() / ref.mem_h.abs().max() < 1e-4
    # gradients of the last step, parameter by parameter
    mine = [p for l in model.encoder.layers for p in (l.weight, l.bias)]
    for p, (w, b) in zip(mine[::2], ref.convs):
        assert (p.grad.cpu() - w.grad).abs().max() / w.grad.abs().max() < 1e-3


def test_eval_mode_leaves_memory_untouched_and_smoke(dev):
    edges, nodes, times = _dtdg(seed=1)
    n_nodes, f = nodes.shape[0], nodes.shape[1] - 1
    graph = CitationGraph(edges, nodes, times, device=dev)
    gnn = GCN(2, f, 16, activation=F.leaky_relu, allow_zero_in_degree=True, dropout=0.2).to(dev)
    rnn = GRU(16, 10, n_nodes, 2, MLP(10, [10, 20, 10, 5])).to(dev)
    model = assembler(gnn, rnn)
    model.eval()
    rnn.eval_mode()
    before = rnn.memory_h.memory.clone()
    snap, ids = graph.dispatcher(3)
    with torch.no_grad():
        y1 = model((snap, ids))
        y2 = model((snap, ids))
    assert torch.equal(before, rnn.memory_h.memory) and torch.equal(y1, y2)
    assert y1.shape == (snap.num_nodes(), 1)
    rnn.training_mode = True
    with torch.no_grad():
        model((snap, ids))
    assert not torch.equal(before, rnn.memory_h.memory)
    rnn.reset_memory_state()
    assert rnn.memory_h.memory.device.type == "cuda"

    import __graft_entry__ as entry

    entry.smoke()
