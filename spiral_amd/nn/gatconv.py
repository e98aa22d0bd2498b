"""GATConv with the constructor, parameters, initialisation and forward of
dgl.nn.pytorch.conv.GATConv (DGL 0.8), which the reference wraps in GAT
(staticGraphEncoder.py:155-192; examples/torch/citation_prediction.py:69):

    ft = fc(h).view(N, H, D);  el = (ft * attn_l).sum(-1);  er = (ft * attn_r).sum(-1)
    e_uv = LeakyReLU(el_u + er_v);  a_uv = softmax over the incoming edges of v
    out_v = act(sum_u a_uv ft_u (+ residual) + bias)          -> [N, H, D]

The dense projections are library GEMMs; the edge softmax, the weighted sum and their
gradients run in libspmi.  Attention dropout is not provided (the reference never sets
it); `feat_drop` is."""
from __future__ import annotations

import torch
from torch import nn

from spiral_amd.core.errors import GraphError
from spiral_amd.graph import SnapshotGraph
from spiral_amd.nn.linear import TallLinear
from spiral_amd.nn.graphconv import _classify_activation
from spiral_amd.ops import gat_aggregate


class GATConv(nn.Module):
    def __init__(self, in_feats: int, out_feats: int, num_heads: int, feat_drop: float = 0.0,
                 attn_drop: float = 0.0, negative_slope: float = 0.2, residual: bool = False,
                 activation=None, allow_zero_in_degree: bool = False, bias: bool = True):
        super().__init__()
        if not 0.0 <= attn_drop < 1.0:
            raise ValueError(f"attn_drop must be in [0, 1), got {attn_drop}")
        self._attn_drop = float(attn_drop)
        self._num_heads = num_heads
        self._in_src_feats = self._in_dst_feats = in_feats
        self._out_feats = out_feats
        self._allow_zero_in_degree = allow_zero_in_degree
        self._negative_slope = negative_slope
        self.fc = TallLinear(in_feats, out_feats * num_heads, bias=False)
        self.attn_l = nn.Parameter(torch.empty(1, num_heads, out_feats))
        self.attn_r = nn.Parameter(torch.empty(1, num_heads, out_feats))
        self.feat_drop = nn.Dropout(feat_drop)
        if bias:
            self.bias = nn.Parameter(torch.empty(num_heads * out_feats))
        else:
            self.register_parameter("bias", None)
        if residual:
            if in_feats != out_feats * num_heads:
                self.res_fc = TallLinear(in_feats, num_heads * out_feats, bias=False)
            else:
                self.res_fc = nn.Identity()
        else:
            self.res_fc = None
        self.activation = activation
        self.reset_parameters()

    def reset_parameters(self) -> None:
        gain = nn.init.calculate_gain("relu")
        nn.init.xavier_normal_(self.fc.weight, gain=gain)
        nn.init.xavier_normal_(self.attn_l, gain=gain)
        nn.init.xavier_normal_(self.attn_r, gain=gain)
        if self.bias is not None:
            nn.init.constant_(self.bias, 0)
        if isinstance(self.res_fc, nn.Linear):
            nn.init.xavier_normal_(self.res_fc.weight, gain=gain)

    def set_allow_zero_in_degree(self, set_value: bool) -> None:
        self._allow_zero_in_degree = set_value

    def forward(self, graph: SnapshotGraph, feat: torch.Tensor, get_attention: bool = False):
        if not self._allow_zero_in_degree and bool((graph.in_degrees() == 0).any()):
            raise GraphError(
                "There are 0-in-degree nodes in the graph, output for those nodes will be invalid. "
                "Adding self-loop on the input graph by calling `g = dgl.add_self_loop(g)` will "
                "resolve the issue. Setting ``allow_zero_in_degree`` to be `True` when constructing "
                "this module will suppress the check and let the code run.")
        heads, d = self._num_heads, self._out_feats
        h = self.feat_drop(feat)
        ft = self.fc(h)
        ft3 = ft.view(-1, heads, d)
        el = (ft3 * self.attn_l).sum(dim=-1)
        er = (ft3 * self.attn_r).sum(dim=-1)
        code, slope, unfused = _classify_activation(self.activation)
        fuse = self.res_fc is None  # bias and activation ride in the aggregation epilogue
        rst, attn = gat_aggregate(graph, ft, el, er, heads, self._negative_slope,
                                  self.bias if fuse else None, code if fuse else None, slope,
                                  attn_drop=self._attn_drop if self.training else 0.0)
        if not fuse:
            rst = rst + self.res_fc(h)
            if self.bias is not None:
                rst = rst + self.bias
            if self.activation is not None:
                rst = self.activation(rst)
        elif unfused is not None:
            rst = unfused(rst)
        rst = rst.view(-1, heads, d)
        if get_attention:
            csr = graph.csr(want_eid=True)
            by_edge = torch.empty(csr.nnz, heads, 1, dtype=attn.dtype, device=attn.device)
            by_edge[csr.eid.long(), :, 0] = attn.t()
            return rst, by_edge
        return rst
