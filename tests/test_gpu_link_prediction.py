"""Link prediction as a loop (BASELINE configs[0]): the training loop of the reference's
examples/tf/link_prediction_sliding_window.py:99-146 on the torch path.

    prepare_edge_task -> per snapshot: remove the positive edges, encoder, recurrent step on the
    distinct nodes of the target pairs, pair head, BCE, backward, Adam -> held-out snapshot, MRR

The reference ships this task for its TensorFlow backend only and holds no numeric fixture for
it (parity unpinned); the check is the CPU restatement `oracle.link_prediction_loop_torch`,
which follows the script line by line.  Bars: loss and every gradient of the first step at
1e-5 (relative to the gradient's largest entry); later steps have been through Adam, which
turns rounding differences of near-zero gradient entries into steps of +-lr (8 steps of 1e-3
here), so the loss list is held to 1e-3 (measured 1e-7), the evaluation predictions (sigmoid
outputs, absolute) to 5e-3 (measured 3e-4 to 1.2e-3; the largest for the swapped loss, whose
gradient with respect to a prediction is a constant +-16/m that keeps near-zero entries near
zero) and the MRR to a rank change of two positives (measured values are printed)."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

from oracle import oracle as orc
from spiral_amd.automl.evaluate import link_prediction_mrr
from spiral_amd.automl.model_assembler import assembler
from spiral_amd.automl.prepare_dataset import LABEL, TARGET, prepare_edge_task
from spiral_amd.dtdg.models.decoder.torch.sequentialDecoder.implicitTimeModels import GRU, LSTM
from spiral_amd.dtdg.models.decoder.torch.simpleDecoder import MLP
from spiral_amd.dtdg.models.encoder.implicitTimeEncoder.staticGraphEncoder import GCN
from spiral_amd.dtdg.types import VEInvariantDTDG

pytestmark = pytest.mark.gpu


def rel(a, b):
    a, b = np.asarray(a, dtype=np.float64), np.asarray(b, dtype=np.float64)
    return np.abs(a - b).max() / max(np.abs(b).max(), 1e-30)


def _tables(seed, n, e, steps, f):
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
    return edges, nodes


def swapped_bce(predict, label, eps=1e-7):
    """The loss of the script as written (:109): Keras BinaryCrossentropy called with
    (prediction, label) where it expects (y_true, y_pred)."""
    y = label.clamp(eps, 1 - eps)
    return -(predict * torch.log(y) + (1 - predict) * torch.log(1 - y)).mean()


@pytest.mark.parametrize("kind,loss_order", [("gru", "bce"), ("lstm", "bce"), ("gru", "swapped")])
def test_link_prediction_loop_against_the_cpu_restatement(dev, kind, loss_order):
    n, steps, f, hid, out_d = 4000, 7, 8, 32, 16
    edges, nodes = _tables(3, n, 60_000, steps, f)
    graph = VEInvariantDTDG(edges, nodes, np.arange(steps), device=dev)
    eval_t, start, n_eval_neg = steps - 1, 2, 50
    np.random.seed(5)
    prepare_edge_task(graph, 2, num_negative=n_eval_neg, start_t=eval_t)     # held-out snapshot (:37-42)
    eval_task = (graph.time_data[TARGET][eval_t], graph.time_data[LABEL][eval_t].float(),
                 graph.time_data["positive_edges"][eval_t])
    prepare_edge_task(graph, 2, num_negative=6, start_t=start)               # per epoch in the script (:103)
    train_ts = list(range(start, steps - 1))
    assert len(train_ts) >= 3
    tasks = {t: (graph.time_data[TARGET][t].numpy(), graph.time_data[LABEL][t].numpy(),
                 graph.time_data["positive_edges"][t].numpy()) for t in train_ts}
    tasks[eval_t] = tuple(a.numpy() for a in eval_task)

    torch.manual_seed(0)
    gnn = GCN(2, f, hid, activation=F.leaky_relu, allow_zero_in_degree=True)
    head = MLP(2 * out_d, [16, 8], dropout=0.0, activation="sigmoid")
    rnn = {"gru": GRU, "lstm": LSTM}[kind](hid, out_d, n, 2, head)
    model = assembler(gnn, rnn)
    convs = [(l.weight.detach().numpy().copy(), l.bias.detach().numpy().copy()) for l in gnn.layers]
    rnn_params = [tuple(getattr(rnn.base_model, f"{nm}_l{k}").detach().numpy().copy()
                        for nm in ("weight_ih", "weight_hh", "bias_ih", "bias_hh")) for k in range(2)]
    mlp_layers = []
    for m in head.layers:
        if isinstance(m, torch.nn.Linear):
            mlp_layers.append({"w": m.weight.detach().numpy().copy(), "b": m.bias.detach().numpy().copy()})
        elif isinstance(m, torch.nn.BatchNorm1d):
            mlp_layers.append({"gamma": m.weight.detach().numpy().copy(), "beta": m.bias.detach().numpy().copy(),
                               "mean": m.running_mean.numpy().copy(), "var": m.running_var.numpy().copy()})
    mem_h0 = rnn.memory_h.memory.clone()
    mem_c0 = rnn.memory_c.memory.clone() if kind == "lstm" else None
    want = orc.link_prediction_loop_torch(edges, nodes, steps, tasks, train_ts, eval_t, convs, kind, rnn_params,
                                          mlp_layers, mem_h0.numpy(), None if mem_c0 is None else mem_c0.numpy(),
                                          epochs=2, loss_order=loss_order, eval_negatives_per_node=n_eval_neg)

    model.to(dev)
    rnn.to(dev)
    params = [p for l in gnn.layers for p in (l.weight, l.bias)] + \
             [getattr(rnn.base_model, f"{nm}_l{k}") for k in range(2)
              for nm in ("weight_ih", "weight_hh", "bias_ih", "bias_hh")] + \
             [p for m in head.layers if isinstance(m, (torch.nn.Linear, torch.nn.BatchNorm1d))
              for p in (m.weight, m.bias)]
    opt = torch.optim.Adam(model.parameters(), lr=1e-3, eps=1e-8)
    loss_fn = swapped_bce if loss_order == "swapped" else F.binary_cross_entropy
    losses, first_grads = [], None
    for _ in range(2):
        model.train()
        rnn.train_mode()
        rnn.memory_h.memory.copy_(mem_h0)       # the script re-draws the memory per epoch (:101)
        if mem_c0 is not None:
            rnn.memory_c.memory.copy_(mem_c0)
        for t in train_ts:
            snap, _ = graph.dispatcher(t)
            snap.observation.remove_edges(graph.time_data["positive_edges"][t])
            label = graph.time_data[LABEL][t].float().to(dev)
            opt.zero_grad()
            predict = model((snap, graph.time_data[TARGET][t]))
            loss = loss_fn(predict.squeeze(-1), label)
            loss.backward()
            if first_grads is None:
                first_grads = [p.grad.detach().cpu().numpy().copy() for p in params]
            opt.step()
            losses.append(loss.item())
    model.eval()
    rnn.eval_mode()
    with torch.no_grad():
        snap, _ = graph.dispatcher(eval_t)
        snap.observation.remove_edges(eval_task[2])
        predict = model((snap, eval_task[0])).squeeze(-1)
    n_pos = len(eval_task[2])
    got_mrr = link_prediction_mrr(predict, n_pos, negatives_per_node=n_eval_neg)

    assert abs(losses[0] - want["losses"][0]) / abs(want["losses"][0]) < 1e-5
    # A Linear bias that feeds a train-mode BatchNorm has a gradient of exactly zero: both sides
    # hold rounding noise there, which must be zero within 1e-5 of the model's largest gradient
    # entry instead of agreeing digit by digit.
    mods = [m for m in head.layers if isinstance(m, (torch.nn.Linear, torch.nn.BatchNorm1d))]
    zero_grad = set()
    for i, m in enumerate(mods[:-1]):
        if isinstance(m, torch.nn.Linear) and isinstance(mods[i + 1], torch.nn.BatchNorm1d):
            zero_grad.add(len(params) - 2 * len(mods) + 2 * i + 1)
    top = max(np.abs(w).max() for w in want["first_grads"])
    errs = [np.abs(g - w).max() / top if i in zero_grad else rel(g, w)
            for i, (g, w) in enumerate(zip(first_grads, want["first_grads"]))]
    worst = max(errs)
    assert worst < 1e-5, f"first-step gradients off by {worst:.2e}: {[f'{e:.1e}' for e in errs]}"
    loss_err = max(abs(a - b) / abs(b) for a, b in zip(losses, want["losses"]))
    assert loss_err < 1e-3, f"loss list off by {loss_err:.2e}: {losses} vs {want['losses']}"
    pred_err = np.abs(predict.cpu().numpy() - want["eval_predict"]).max()
    assert pred_err < 5e-3, f"evaluation predictions off by {pred_err:.2e}"
    assert abs(got_mrr - want["mrr"]) < 1e-3 * max(want["mrr"], 1e-3) + 2.0 / n_pos, (got_mrr, want["mrr"])
    assert rel(rnn.memory_h.memory.cpu().numpy(), want["mem_h"]) < 1e-3
    assert 0.0 < got_mrr <= 1.0 and len(losses) == 2 * len(train_ts)
    print(f"{kind}/{loss_order}: first-step grads {worst:.1e}, loss list {loss_err:.1e}, "
          f"eval predictions {pred_err:.1e}, MRR {got_mrr:.4f} vs {want['mrr']:.4f}")
