"""SAGEConv, 'gcn' and 'mean' aggregators, with the constructor, parameters and forward of
dgl.nn.pytorch.conv.SAGEConv (DGL 0.8), which the reference wraps in GraphSage
(staticGraphEncoder.py:195-225; examples/torch/citation_prediction.py:71 uses 'gcn'):

    gcn:   h_neigh = (sum_{u->v} h_u + h_v) / (in_deg(v) + 1) ;  out = act(h_neigh W^T + b)
    mean:  h_neigh = mean_{u->v} h_u (0 for a node without in-edges) ;
           out = act(h_v W_self^T + h_neigh W^T + b)

The linear map of the neighbours runs before the aggregation when in_feats > out_feats.  The
neighbour sum (and the division by the in-degree of 'mean') is the libspmi aggregation
kernel; DGL's 'pool' and 'lstm' aggregators are not provided."""
from __future__ import annotations

import torch
from torch import nn

from spiral_amd.core.errors import GraphError
from spiral_amd.graph import SnapshotGraph
from spiral_amd.nn.linear import TallLinear
from spiral_amd.ops import gcn_aggregate


class SAGEConv(nn.Module):
    def __init__(self, in_feats: int, out_feats: int, aggregator_type: str, feat_drop: float = 0.0,
                 bias: bool = True, norm=None, activation=None):
        super().__init__()
        if aggregator_type not in ("mean", "pool", "lstm", "gcn"):
            raise GraphError(f"Invalid aggregator_type. Must be one of mean, pool, lstm, gcn. "
                             f"But got {aggregator_type!r} instead.")
        if aggregator_type not in ("gcn", "mean"):
            raise NotImplementedError("only the 'gcn' and 'mean' aggregators are implemented")
        self._in_src_feats = self._in_dst_feats = in_feats
        self._out_feats = out_feats
        self._aggre_type = aggregator_type
        self.norm = norm
        self.feat_drop = nn.Dropout(feat_drop)
        self.activation = activation
        self.fc_neigh = TallLinear(in_feats, out_feats, bias=False)
        if aggregator_type != "gcn":  # 'gcn' has no separate map of the node's own features
            self.fc_self = TallLinear(in_feats, out_feats, bias=False)
        if bias:
            self.bias = nn.Parameter(torch.zeros(out_feats))
        else:
            self.register_parameter("bias", None)
        self.reset_parameters()

    def reset_parameters(self) -> None:
        gain = nn.init.calculate_gain("relu")
        if self._aggre_type != "gcn":
            nn.init.xavier_uniform_(self.fc_self.weight, gain=gain)
        nn.init.xavier_uniform_(self.fc_neigh.weight, gain=gain)

    def forward(self, graph: SnapshotGraph, feat: torch.Tensor, edge_weight=None) -> torch.Tensor:
        feat = self.feat_drop(feat)
        lin_before_mp = self._in_src_feats > self._out_feats
        h = self.fc_neigh(feat) if lin_before_mp else feat
        if self._aggre_type == "mean":
            # sum / max(in_degree, 1): GraphConv's 'right' normalisation of the same kernel
            h_neigh = gcn_aggregate(graph, h, "right", None, edge_weight)
            if not lin_before_mp:
                h_neigh = self.fc_neigh(h_neigh)
            rst = self.fc_self(feat) + h_neigh
        else:
            neigh = gcn_aggregate(graph, h, "none", None, edge_weight)
            degs = graph.in_degrees().to(h.dtype)
            h_neigh = (neigh + h) / (degs.unsqueeze(-1) + 1)
            if not lin_before_mp:
                h_neigh = self.fc_neigh(h_neigh)
            rst = h_neigh
        if self.bias is not None:
            rst = rst + self.bias
        if self.activation is not None:
            rst = self.activation(rst)
        if self.norm is not None:
            rst = self.norm(rst)
        return rst
