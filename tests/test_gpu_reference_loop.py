"""The reference's training script as it runs (examples/torch/citation_prediction.py:83-101):
Adam(lr=1e-3) (:85), two epochs (:88) with `reset_memory_state()` at the start of each (:89),
t over range(5, n_snapshot - 2) (:90), labels from `dispatcher(t + 1)` (:94,97), MSE (:97),
backward and optimiser step (:98-101).  The GPU model against a CPU restatement (oracle
GraphConv in torch ops + torch.nn.GRU/LSTM + the MLP) with the same parameters; the node
memories drawn by `reset_memory_state` are copied into the restatement after every reset
(the reference draws them with torch.randn, implicitTimeModels.py:46).

Bars.  First step (identical parameters and state): loss 1e-5 relative, every parameter's
gradient 1e-5 of that gradient's largest entry.  Later steps: Adam divides the gradient by
its own magnitude, so the first updates are +-lr whatever the gradient's size and a rounding
difference in a near-zero gradient entry moves a weight by up to 2 lr; losses are held to
1e-5 relative over both epochs (measured 1.5e-7), the memories to 1e-4 of their largest entry
(measured 7e-8 for the GRU, 3e-6 for the LSTM) and the parameters to 2e-4 absolute (measured
6e-8 / 1.7e-5); the test prints what it measures."""
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


def _dtdg(seed=0, n=2000, e=16000, steps=10, f=6):
    rng = np.random.default_rng(seed)
    node_t = np.sort(rng.integers(0, steps, n))
    nodes = np.hstack([node_t[:, None], rng.standard_normal((n, f))])
    a, b = rng.integers(0, n, e), rng.integers(0, n, e)
    s, d = np.maximum(a, b), np.minimum(a, b)
    et = node_t[s]
    edges = np.stack([et, s, d], 1)[np.argsort(et, kind="stable")].astype(np.float64)
    return edges, nodes, np.arange(steps)


class CpuLoop:
    def __init__(self, model, kind):
        self.kind = kind
        self.convs = [(l.weight.detach().cpu().clone().requires_grad_(),
                       l.bias.detach().cpu().clone().requires_grad_()) for l in model.encoder.layers]
        self.rnn = copy.deepcopy(model.decoder.base_model).cpu()
        self.mlp = copy.deepcopy(model.decoder.simple_decoder).cpu()
        self.mlp.FUSE_EVAL = False
        self.params = [p for wb in self.convs for p in wb] + list(self.rnn.parameters()) + \
            list(self.mlp.parameters())
        self.mem_h = self.mem_c = None

    def take_memory(self, decoder):
        self.mem_h = decoder.memory_h.memory.detach().cpu().clone()
        if self.kind == "lstm":
            self.mem_c = decoder.memory_c.memory.detach().cpu().clone()

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


def _gpu_params(model):
    return [p for l in model.encoder.layers for p in (l.weight, l.bias)] + \
        list(model.decoder.base_model.parameters()) + list(model.decoder.simple_decoder.parameters())


def _zero_gradient_biases(mlp, offset):
    """Positions (in the parameter list) of the Linear biases that feed a train-mode
    BatchNorm1d: the batch mean removes them, so their gradient is exactly zero."""
    out, pos, mods = [], offset, list(mlp.layers)
    for m, nxt in zip(mods, mods[1:] + [None]):
        n_par = len(list(m.parameters()))
        if isinstance(m, torch.nn.Linear) and isinstance(nxt, torch.nn.BatchNorm1d) and m.bias is not None:
            out.append(pos + 1)
        pos += n_par
    return out


def _check_first_step_gradients(mine, theirs, mlp):
    """Every parameter's gradient within 1e-5 of that gradient's largest entry.  Linear biases
    in front of a train-mode BatchNorm1d have a gradient of exactly zero; both sides hold
    rounding noise there (column sums that cancel), which is required to be zero within 1e-5
    of the largest gradient entry of the model instead of being compared digit by digit."""
    gmax = max(float(q.grad.abs().max()) for q in theirs)
    zero = set(_zero_gradient_biases(mlp, len(theirs) - len(list(mlp.parameters()))))
    assert zero, "the head is expected to hold Linear -> BatchNorm1d pairs"
    bad = []
    worst = 0.0
    for i, (p, q) in enumerate(zip(mine, theirs)):
        assert p.grad is not None and q.grad is not None, i
        if i in zero:
            r = max(float(p.grad.abs().max()), float(q.grad.abs().max())) / gmax
        else:
            r = float((p.grad.cpu() - q.grad).abs().max()) / float(q.grad.abs().max())
        worst = max(worst, r)
        if not r <= 1e-5:
            bad.append((i, tuple(q.shape), f"{r:.2e}"))
    assert not bad, bad
    return worst


@pytest.mark.parametrize("kind", ["gru", "lstm"])
def test_two_epochs_of_the_reference_loop_with_adam(dev, kind):
    edges, nodes, times = _dtdg()
    n_snapshot, n_nodes, f = len(times), nodes.shape[0], nodes.shape[1] - 1
    graph = CitationGraph(edges, nodes, times, device=dev)
    torch.manual_seed(0)
    gnn = GCN(2, f, 32, activation=F.leaky_relu, allow_zero_in_degree=True)   # :65 without dropout
    cls = {"gru": GRU, "lstm": LSTM}[kind]
    rnn = cls(32, 10, n_nodes, 2, MLP(10, [10, 20, 10, 5], dropout=0.0))      # :73-80
    model = assembler(gnn, rnn)
    ref = CpuLoop(model, kind)
    model.to(dev)
    model.decoder.to(dev)
    model.train()
    ref.rnn.train()
    ref.mlp.train()
    opt = torch.optim.Adam(_gpu_params(model), lr=1e-3)                       # :85
    opt_ref = torch.optim.Adam(ref.params, lr=1e-3)
    losses, losses_ref, first = [], [], True
    for epoch in range(2):                                                    # :88
        torch.manual_seed(100 + epoch)
        model.decoder.reset_memory_state()                                    # :89
        assert model.decoder.memory_h.memory.device.type == "cuda"
        ref.take_memory(model.decoder)
        for t in range(5, n_snapshot - 2):                                    # :90
            snap, ids = graph.dispatcher(t)
            nxt, _ = graph.dispatcher(t + 1)                                  # :94
            opt.zero_grad()
            predict = model((snap, ids))                                      # :96
            label = nxt.node_feature()[: snap.num_nodes(), -1].float()        # :97
            loss = F.mse_loss(predict.squeeze(), label)
            loss.backward()                                                   # :98

            src, dst, n, feat, _ = orc.citation_dispatch(edges, nodes, n_snapshot, t)
            feat_next = orc.citation_dispatch(edges, nodes, n_snapshot, t + 1)[3]
            opt_ref.zero_grad()
            loss_ref = F.mse_loss(ref.forward(src, dst, n, feat).squeeze(),
                                  torch.from_numpy(feat_next[:n, -1]))
            loss_ref.backward()
            if first:
                first_loss = abs(loss.item() - loss_ref.item()) / abs(loss_ref.item())
                assert first_loss < 1e-5, first_loss
                first_grad = _check_first_step_gradients(_gpu_params(model), ref.params, ref.mlp)
                first = False
            opt.step()                                                        # :101
            opt_ref.step()
            losses.append(loss.item())
            losses_ref.append(loss_ref.item())
    assert len(losses) == 2 * (n_snapshot - 7)
    # After the first step the parameters have been through Adam, which turns rounding differences
    # of near-zero gradient entries into steps of +-lr, so these bars are wider than 1e-5: losses
    # 1e-5 (measured 1.5e-7), memory 1e-4 (measured 7e-8 / 3e-6), parameters 2e-4 absolute
    # (measured 6e-8 / 1.7e-5).  What is measured is printed (profiles/r03_parity_measured.txt).
    loss_err = max(abs(a - b) / abs(b) for a, b in zip(losses, losses_ref))
    assert loss_err < 1e-5, (loss_err, losses, losses_ref)
    mem = model.decoder.memory_h.memory.cpu()
    mem_err = float((mem - ref.mem_h).abs().max() / ref.mem_h.abs().max())
    assert mem_err < 1e-4, mem_err
    # parameters after 2 x 3 Adam steps; the zero-gradient biases are left out: Adam turns their
    # rounding noise into +-lr steps on both sides (they do not reach the output in train mode)
    zero = set(_zero_gradient_biases(ref.mlp, len(ref.params) - len(list(ref.mlp.parameters()))))
    par_err = 0.0
    for i, (p, q) in enumerate(zip(_gpu_params(model), ref.params)):
        if i not in zero:
            par_err = max(par_err, float((p.detach().cpu() - q.detach()).abs().max()))
            assert par_err < 2e-4, (i, par_err)
    print(f"reference loop, {kind}: first-step loss {first_loss:.1e} and gradients {first_grad:.1e} (bar 1e-5); "
          f"losses of all steps {loss_err:.1e} (bar 1e-5), memory {mem_err:.1e} (bar 1e-4), "
          f"parameters {par_err:.1e} absolute (bar 2e-4)")
