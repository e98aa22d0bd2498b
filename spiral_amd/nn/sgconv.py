"""SGConv with the constructor, parameters and forward of dgl.nn.pytorch.conv.SGConv
(DGL 0.8), which the reference imports at spiro/core/types.py:34 and wraps in SGCN
(staticGraphEncoder.py:120-152):

    H = (D_in^-1/2 A D_in^-1/2)^k X ;  out = H W^T + b

Each hop is one fused libspmi aggregation (both degree scalings inside the kernel)."""
from __future__ import annotations

from typing import Optional

import torch
from torch import nn

from spiral_amd.core.errors import GraphError
from spiral_amd.graph import SnapshotGraph
from spiral_amd.nn.linear import TallLinear
from spiral_amd.ops import gcn_aggregate


class SGConv(nn.Module):
    def __init__(self, in_feats: int, out_feats: int, k: int = 1, cached: bool = False, bias: bool = True,
                 norm=None, allow_zero_in_degree: bool = False):
        super().__init__()
        self.fc = TallLinear(in_feats, out_feats, bias=bias)
        self._cached = cached
        self._cached_h: Optional[torch.Tensor] = None
        self._k = k
        self.norm = norm
        self._allow_zero_in_degree = allow_zero_in_degree
        self.reset_parameters()

    def reset_parameters(self) -> None:
        nn.init.xavier_uniform_(self.fc.weight)
        if self.fc.bias is not None:
            nn.init.zeros_(self.fc.bias)

    def set_allow_zero_in_degree(self, set_value: bool) -> None:
        self._allow_zero_in_degree = set_value

    def forward(self, graph: SnapshotGraph, feat: torch.Tensor, edge_weight: Optional[torch.Tensor] = None):
        if not self._allow_zero_in_degree and bool((graph.in_degrees() == 0).any()):
            raise GraphError("There are 0-in-degree nodes in the graph, output for those nodes will be "
                             "invalid. Setting ``allow_zero_in_degree`` to be `True` when constructing "
                             "this module will suppress the check and let the code run.")
        if self._cached_h is not None:
            feat = self._cached_h
        else:
            for _ in range(self._k):
                feat = gcn_aggregate(graph, feat, "sgc", None, edge_weight)
            if self.norm is not None:
                feat = self.norm(feat)
            if self._cached:
                self._cached_h = feat
        return self.fc(feat)
