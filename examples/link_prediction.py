"""Link prediction on a discrete-time dynamic graph: the loop of the reference's
examples/tf/link_prediction_sliding_window.py:99-146 on the torch path of `spiral_amd`.

    prepare_edge_task(graph, 2)             positives: 2 sampled successors of every new node,
                                            negatives: sampled non-neighbours (prepare_dataset.py:56-94)
    per snapshot t:                         remove the positive edges from the snapshot (:105-106),
                                            encoder -> recurrent step on the distinct nodes of the
                                            target pairs -> pair head (sigmoid) -> BCE -> Adam
    held-out last snapshot:                 1000 negatives per evaluated node, MRR (:128-146)

The reference's script decodes with sliding-window attention models that exist only for its
TensorFlow backend; the torch path has the node-memory GRU / LSTM / RNN decoders, which is what
runs here.  `--reference-loss-order` reproduces the script's loss call as written
(`loss_fn(predict, label)` on a Keras loss whose signature is (y_true, y_pred), :109); the
default is the binary cross-entropy of the prediction against the label.

    python examples/link_prediction.py            # synthetic citation graph, needs an MI355X
"""
import argparse
import os
import sys
import time

import numpy as np
import torch
import torch.nn.functional as F

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from spiral_amd.automl.evaluate import link_prediction_mrr  # noqa: E402
from spiral_amd.automl.model_assembler import assembler  # noqa: E402
from spiral_amd.automl.prepare_dataset import LABEL, TARGET, prepare_edge_task  # noqa: E402
from spiral_amd.dtdg.models.decoder.sequentialDecoders import GRU, LSTM, RNN  # noqa: E402
from spiral_amd.dtdg.models.decoder.simpleDecoders import MLP  # noqa: E402
from spiral_amd.dtdg.models.encoder.implicitTimeEncoder.staticGraphEncoder import GAT, GCN, SGCN, GraphSage  # noqa: E402
from spiral_amd.dtdg.types import VEInvariantDTDG  # noqa: E402


def synthetic_tables(n_nodes: int, n_edges: int, steps: int, n_feat: int, seed: int = 0):
    """Tables in the reference's format (spiro/dtdg/dataloader.py:36-71): nodes `[t, feat...]` in
    time order, edges `[t, src, dst]` newer -> older with the source's time, distinct pairs (the
    container's edge-id dictionary is keyed by the pair)."""
    rng = np.random.default_rng(seed)
    node_t = np.sort(rng.integers(0, steps, n_nodes))
    nodes = np.hstack([node_t[:, None], rng.standard_normal((n_nodes, n_feat))])
    a = rng.integers(0, n_nodes, n_edges)
    b = np.minimum((n_nodes * rng.random(n_edges) ** 2.0).astype(np.int64), n_nodes - 1)
    keep = a != b
    s, d = np.maximum(a, b)[keep], np.minimum(a, b)[keep]
    _, first = np.unique(s * n_nodes + d, return_index=True)
    s, d = s[np.sort(first)], d[np.sort(first)]
    et = node_t[s]
    edges = np.stack([et, s, d], 1)[np.argsort(et, kind="stable")].astype(np.float64)
    return edges, nodes


def swapped_bce(predict: torch.Tensor, label: torch.Tensor, eps: float = 1e-7) -> torch.Tensor:
    """Keras BinaryCrossentropy()(y_true=predict, y_pred=label): the script's call as written."""
    y = label.clamp(eps, 1 - eps)
    return -(predict * torch.log(y) + (1 - predict) * torch.log(1 - y)).mean()


def masked_snapshot(graph, t):
    """Snapshot t without the positive edges of its task, its target pairs and its labels."""
    snapshot, _ = graph.dispatcher(t)
    snapshot.observation.remove_edges(graph.time_data["positive_edges"][t])
    return snapshot, graph.time_data[TARGET][t], graph.time_data[LABEL][t].float()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gnn", default="gcn", choices=["gcn", "sgcn", "gat", "sage"])
    ap.add_argument("--rnn", default="gru", choices=["gru", "lstm", "rnn"])
    ap.add_argument("--hidden", type=int, default=32)
    ap.add_argument("--output-dim", type=int, default=10)
    ap.add_argument("--epochs", type=int, default=5)
    ap.add_argument("--nodes", type=int, default=20_000)
    ap.add_argument("--edges", type=int, default=300_000)
    ap.add_argument("--steps", type=int, default=17)
    ap.add_argument("--start", type=int, default=12, help="first training snapshot (the script's `start`)")
    ap.add_argument("--eval-negatives", type=int, default=1000)
    ap.add_argument("--reference-loss-order", action="store_true")
    args = ap.parse_args()
    dev = torch.device("cuda:0")

    edges, nodes = synthetic_tables(args.nodes, args.edges, args.steps, n_feat=16)
    graph = VEInvariantDTDG(edges, nodes, np.arange(args.steps), device=dev)
    n_snapshot = len(graph)
    eval_t = n_snapshot - 1
    np.random.seed(0)
    prepare_edge_task(graph, 2, num_negative=args.eval_negatives, start_t=eval_t)
    eval_snapshot, eval_pairs, _ = masked_snapshot(graph, eval_t)
    n_pos_eval = len(graph.time_data["positive_edges"][eval_t])

    in_dim, act = eval_snapshot.num_node_features(), F.relu
    if args.gnn == "gcn":
        gnn = GCN(2, in_dim, args.hidden, activation=act, allow_zero_in_degree=True, dropout=0.2)
    elif args.gnn == "sgcn":
        gnn = SGCN(2, in_dim, args.hidden, allow_zero_in_degree=True)
    elif args.gnn == "gat":
        gnn = GAT([1], in_dim, args.hidden, activation=act, allow_zero_in_degree=True)
    else:
        gnn = GraphSage("gcn", in_dim, args.hidden, activation=act)
    head = MLP(2 * args.output_dim, [args.hidden, 20, 10, 5], activation="sigmoid")
    cell = {"gru": GRU, "lstm": LSTM, "rnn": RNN}[args.rnn]
    model = assembler(gnn, cell(args.hidden, args.output_dim, eval_snapshot.num_nodes(), 2, head)).to(dev)
    opt = torch.optim.Adam(model.parameters(), lr=1e-3, eps=1e-8)
    loss_fn = swapped_bce if args.reference_loss_order else F.binary_cross_entropy

    best = float("inf")
    for epoch in range(args.epochs):
        t0 = time.perf_counter()
        model.train()
        model.decoder.train_mode()
        model.decoder.reset_memory_state()
        prepare_edge_task(graph, 2, start_t=args.start)        # fresh samples every epoch (:103)
        for t in range(args.start, n_snapshot - 1):
            snapshot, pairs, label = masked_snapshot(graph, t)
            opt.zero_grad()
            predict = model((snapshot, pairs))
            loss = loss_fn(predict.squeeze(-1), label.to(dev))
            loss.backward()
            opt.step()
        model.eval()
        model.decoder.eval_mode()
        with torch.no_grad():
            snapshot, pairs, label = masked_snapshot(graph, n_snapshot - 1)
            test = loss_fn(model((snapshot, pairs)).squeeze(-1), label.to(dev)).item()
            line = f"epoch {epoch}: train loss {loss.item():.4f}  test loss {test:.4f}"
            if test < best:                                     # the script evaluates at a new best (:127)
                best = test
                score = model((eval_snapshot, eval_pairs)).squeeze(-1)
                mrr = link_prediction_mrr(score, n_pos_eval, negatives_per_node=args.eval_negatives)
                line += f"  MRR {mrr:.4f} ({n_pos_eval // 2} nodes, {args.eval_negatives} negatives each)"
        torch.cuda.synchronize()
        print(line + f"  {time.perf_counter() - t0:.2f} s", flush=True)


if __name__ == "__main__":
    main()
