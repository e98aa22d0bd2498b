"""Citation-count prediction on a discrete-time dynamic graph, the pipeline of the reference's
examples/torch/citation_prediction.py with `spiro` replaced by `spiral_amd`: a GCN / SGCN / GAT /
GraphSage encoder per snapshot, a GRU / LSTM / RNN over per-node memory, an MLP head; labels are
the citation counts of the next snapshot; Adam, MSE, one backward pass per snapshot.

    python examples/citation_prediction.py                     # synthetic citation graph
    python examples/citation_prediction.py --dataset ogbn-arxiv  # needs the `ogb` package and its data

Needs an MI355X: there is no CPU path.  The synthetic graph has the reference's table format
(nodes time-ordered, edges newer -> older, `spiro/dtdg/dataloader.py:36-71`)."""
import argparse
import os
import sys
import time

import numpy as np
import torch
import torch.nn.functional as F

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from spiral_amd.core.backends import PYTORCH  # noqa: E402
from spiral_amd.core.config import set_backend  # noqa: E402

set_backend(PYTORCH)
from spiral_amd.automl.model_assembler import assembler  # noqa: E402
from spiral_amd.dtdg.dataloader import ogb_dataset, preprocess_citation_graph, supported_ogb_datasets  # noqa: E402
from spiral_amd.dtdg.models.decoder.sequentialDecoders import GRU, LSTM, RNN  # noqa: E402
from spiral_amd.dtdg.models.decoder.simpleDecoders import MLP  # noqa: E402
from spiral_amd.dtdg.models.encoder.implicitTimeEncoder.staticGraphEncoder import GAT, GCN, SGCN, GraphSage  # noqa: E402


def synthetic_citation_data(n_nodes: int, n_edges: int, years: int, n_feat: int, seed: int = 0):
    """A dictionary of the shape an OGB node-property dataset hands out: `edge_index` (2, |E|),
    `node_feat` (|V|, f) with the label as last column, `node_year` (|V|, 1).  Papers cite
    earlier papers with a preference for already cited ones (a power law over the id)."""
    rng = np.random.default_rng(seed)
    year = np.sort(rng.integers(2000, 2000 + years, n_nodes))
    cited = np.minimum((n_nodes * rng.random(n_edges) ** 3.0).astype(np.int64), n_nodes - 1)
    citing = rng.integers(0, n_nodes, n_edges)
    feat = rng.standard_normal((n_nodes, n_feat)).astype(np.float32)
    label = rng.integers(0, 40, (n_nodes, 1)).astype(np.float32)
    return {"edge_index": np.stack([citing, cited]), "node_feat": np.hstack([feat, label]),
            "node_year": year.reshape(-1, 1)}


def build_model(args, in_dim: int, n_nodes: int):
    act = F.leaky_relu
    if args.gnn == "gcn":
        gnn = GCN(args.gnn_layers, in_dim, args.hidden, activation=act, allow_zero_in_degree=True,
                  dropout=args.dropout)
    elif args.gnn == "sgcn":
        gnn = SGCN(args.gnn_layers, in_dim, args.hidden, allow_zero_in_degree=True)
    elif args.gnn == "gat":
        gnn = GAT([1] * args.gnn_layers, in_dim, args.hidden, activation=act, allow_zero_in_degree=True)
    else:
        gnn = GraphSage("gcn", in_dim, args.hidden, depth=args.gnn_layers, activation=act)
    head = MLP(args.rnn_hidden, [10, 20, 10, 5])
    cell = {"gru": GRU, "lstm": LSTM, "rnn": RNN}[args.rnn]
    return assembler(gnn, cell(args.hidden, args.rnn_hidden, n_nodes, args.rnn_layers, head))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--dataset", default="synthetic", choices=["synthetic"] + supported_ogb_datasets())
    ap.add_argument("--gnn", default="gcn", choices=["gcn", "sgcn", "gat", "sage"])
    ap.add_argument("--rnn", default="gru", choices=["gru", "lstm", "rnn"])
    ap.add_argument("--gnn-layers", type=int, default=2)
    ap.add_argument("--rnn-layers", type=int, default=2)
    ap.add_argument("--hidden", type=int, default=32)
    ap.add_argument("--rnn-hidden", type=int, default=10)
    ap.add_argument("--dropout", type=float, default=0.1)
    ap.add_argument("--epochs", type=int, default=3)
    ap.add_argument("--nodes", type=int, default=20_000)
    ap.add_argument("--edges", type=int, default=200_000)
    ap.add_argument("--years", type=int, default=12)
    args = ap.parse_args()
    dev = torch.device("cuda:0")

    if args.dataset == "synthetic":
        graph = preprocess_citation_graph(
            synthetic_citation_data(args.nodes, args.edges, args.years, n_feat=16), "node_year", device=dev)
    else:
        graph = ogb_dataset(args.dataset)
    n_snapshot = len(graph)
    last, _ = graph.dispatcher(n_snapshot - 1)
    n_nodes = last.num_nodes()
    in_dim = last.num_node_features()
    model = build_model(args, in_dim, n_nodes).to(dev)
    opt = torch.optim.Adam(model.parameters(), lr=1e-3)

    first = min(5, n_snapshot - 4)
    for epoch in range(args.epochs):
        model.decoder.reset_memory_state()
        model.train()
        t0 = time.perf_counter()
        for t in range(first, n_snapshot - 2):
            snapshot, ids = graph.dispatcher(t)
            nxt, _ = graph.dispatcher(t + 1)
            opt.zero_grad()
            predict = model((snapshot, ids))
            label = nxt.node_feature()[: snapshot.num_nodes(), -1].float()
            loss = F.mse_loss(predict.squeeze(), label)
            loss.backward()
            opt.step()
        model.eval()
        with torch.no_grad():
            snapshot, ids = graph.dispatcher(n_snapshot - 2)
            nxt, _ = graph.dispatcher(n_snapshot - 1)
            predict = model((snapshot, ids))
            label = nxt.node_feature()[: snapshot.num_nodes(), -1].float()
            test = F.mse_loss(predict.squeeze(), label).item()
        torch.cuda.synchronize()
        print(f"epoch {epoch}: train loss {loss.item():.4f}  test loss {test:.4f}  "
              f"{time.perf_counter() - t0:.2f} s", flush=True)


if __name__ == "__main__":
    main()
