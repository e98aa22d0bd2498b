"""EvolveGCN-O: GCN layers whose weight matrices are the hidden state of a GRU and are
advanced once per snapshot (Pareja et al., "EvolveGCN", AAAI 2020, the -O variant:
W_t = GRU(input = W_{t-1}, hidden = W_{t-1})).

The reference contains no weight-evolution code; what it offers is the hook this module
uses: GraphConv(weight=False).forward(graph, feat, weight=W_t), and GCN passing its
keyword arguments through to GraphConv (staticGraphEncoder.py:93,96).  Parity is defined
against this repository's own CPU oracle (torch.nn.GRU applied to the weight matrix, rows
as batch, the formulation of torch_geometric_temporal's EvolveGCNO).

The GRU step on the [in_feats, out_feats] weight runs through the same fused cell kernel
as the node-memory decoders; features may be stored as bfloat16 (`storage_dtype`), in
which case the projection is a bf16 GEMM and the aggregation reads and writes bf16 rows
with float32 sums.
"""
from __future__ import annotations

from typing import List, Optional, Tuple

import torch
import torch.nn.functional as F
from torch import nn

from spiral_amd import ops
from spiral_amd.dtdg.types import Snapshot
from spiral_amd.nn.graphconv import GraphConv


class WeightGRU(nn.Module):
    """GRU cell over a weight matrix [rows, cols]: every row is one batch element, input and
    previous state are both the matrix itself.  Parameters are those of torch.nn.GRU."""

    def __init__(self, cols: int):
        super().__init__()
        self.gru = nn.GRU(input_size=cols, hidden_size=cols, batch_first=True)

    def forward(self, w: torch.Tensor) -> torch.Tensor:
        g = self.gru
        return ops.gru_cell_dense(w, w, g.weight_ih_l0, g.weight_hh_l0, g.bias_ih_l0, g.bias_hh_l0)


class EvolveGCNO(nn.Module):
    """Stack of GraphConv layers with evolving weights; same call contract as the static
    encoders: forward((snapshot, ids), training) -> (embeddings, ids).

    Every forward call is one snapshot: weights advance first, then the layers run.
    `detach_state=True` truncates back-propagation at the snapshot boundary (the same
    policy the reference applies to its node memories, implicitTimeModels.py:155-156);
    with False gradients flow through the whole weight history until `reset_weights()`.
    """

    def __init__(self, n_layers: int, in_feats: int, n_hidden: int, dropout: float = 0.0,
                 storage_dtype: torch.dtype = torch.float32, detach_state: bool = True, **kwargs):
        super().__init__()
        if "weight" in kwargs:
            raise ValueError("EvolveGCNO owns the layer weights")
        dims = [in_feats] + [n_hidden] * n_layers
        self.layers = nn.ModuleList(GraphConv(dims[i], dims[i + 1], weight=False, **kwargs)
                                    for i in range(n_layers))
        self.initial_weights = nn.ParameterList(nn.Parameter(torch.empty(dims[i], dims[i + 1]))
                                                for i in range(n_layers))
        self.evolvers = nn.ModuleList(WeightGRU(dims[i + 1]) for i in range(n_layers))
        for w in self.initial_weights:
            nn.init.xavier_uniform_(w)
        self.dropout_p = float(dropout)
        self.storage_dtype = storage_dtype
        self.detach_state = detach_state
        self.mini_batch = False
        self._state: Optional[List[torch.Tensor]] = None

    def reset_weights(self) -> None:
        """Start the next snapshot from the learned initial weights again."""
        self._state = None

    def current_weights(self) -> Optional[List[torch.Tensor]]:
        return self._state

    def forward(self, in_sample: Tuple[Snapshot, List], training: bool = False):
        snapshot, dst_node_ids = in_sample
        g = snapshot.observation
        previous = list(self.initial_weights) if self._state is None else self._state
        weights = [evolve(w) for evolve, w in zip(self.evolvers, previous)]
        h = snapshot.node_feature().to(self.storage_dtype)
        for i, (layer, w) in enumerate(zip(self.layers, weights)):
            if i > 0 and training and self.dropout_p > 0:
                h = F.dropout(h, self.dropout_p, training=True)
            h = layer(g, h, weight=w)
        self._state = [w.detach() for w in weights] if self.detach_state else weights
        return (h, dst_node_ids)
