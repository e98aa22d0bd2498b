"""GraphConv with the constructor, parameters and forward signature of
dgl.nn.pytorch.conv.GraphConv (DGL 0.8), which the reference imports at
spiro/core/types.py:34 and calls at staticGraphEncoder.py:93,96,113,116.

    h = act( D_in^-1/2 A (D_out^-1/2 X) W + b )          (norm='both')

X W runs first when in_feats > out_feats, as in DGL, and also when in_feats == out_feats:
DGL aggregates first in that tie, but the sparse sum is equally wide either way and
multiplying first lets bias and activation ride in the aggregation kernel's epilogue
instead of two more passes over the output (same result up to fp32 rounding order,
inside the 1e-5 parity bar).  The degree scalings, the sparse sum, the bias and
(relu / leaky_relu) activation are one fused libspmi launch; the dense product is a plain
library GEMM.
"""
from __future__ import annotations

from typing import Callable, Optional, Tuple

import torch
import torch.nn.functional as F
from torch import nn

from spiral_amd.core.errors import GraphError
from spiral_amd.graph import SnapshotGraph
from spiral_amd.ops import (WeightCacheHooks, gcn_aggregate, graphconv_inference, graphconv_project_first,
                            project_rows_inference,
                            rows_times_weight)


def _classify_activation(act) -> Tuple[Optional[str], float, Optional[Callable]]:
    """(fused code, slope, unfused callable)."""
    if act is None:
        return None, 0.01, None
    if act in (F.relu, torch.relu) or isinstance(act, nn.ReLU):
        return "relu", 0.0, None
    if act is F.leaky_relu:
        return "leaky_relu", 0.01, None
    if isinstance(act, nn.LeakyReLU):
        return "leaky_relu", float(act.negative_slope), None
    return None, 0.01, act


class GraphConv(WeightCacheHooks, nn.Module):
    def __init__(self, in_feats: int, out_feats: int, norm: str = "both", weight: bool = True,
                 bias: bool = True, activation=None, allow_zero_in_degree: bool = False):
        super().__init__()
        if norm not in ("none", "both", "right", "left"):
            raise GraphError(
                f'Invalid norm value. Must be either "none", "both", "right" or "left". But got "{norm}".')
        self._in_feats, self._out_feats, self._norm = in_feats, out_feats, norm
        self._allow_zero_in_degree = allow_zero_in_degree
        # in_feats == out_feats: both orders give the same result.  DGL aggregates first on a tie
        # (dgl/nn/pytorch/conv/graphconv.py: `if self._in_feats > self._out_feats` multiplies first);
        # here the tie multiplies first by default, because the projection's output then feeds the
        # aggregation's fused epilogue.  "aggregate_first" takes DGL's order.
        self.tie_order = "multiply_first"
        if weight:
            self.weight = nn.Parameter(torch.empty(in_feats, out_feats))
        else:
            self.register_parameter("weight", None)
        if bias:
            self.bias = nn.Parameter(torch.empty(out_feats))
        else:
            self.register_parameter("bias", None)
        self.reset_parameters()
        self._activation = activation

    def reset_parameters(self) -> None:
        """Glorot-uniform weight, zero bias (DGL GraphConv.reset_parameters)."""
        if self.weight is not None:
            nn.init.xavier_uniform_(self.weight)
        if self.bias is not None:
            nn.init.zeros_(self.bias)

    def set_allow_zero_in_degree(self, set_value: bool) -> None:
        self._allow_zero_in_degree = set_value

    def forward(self, graph: SnapshotGraph, feat: torch.Tensor, weight: Optional[torch.Tensor] = None,
                edge_weight: Optional[torch.Tensor] = None) -> torch.Tensor:
        if not isinstance(graph, SnapshotGraph):
            raise GraphError(f"expected a spiral_amd SnapshotGraph, got {type(graph).__name__}")
        if not self._allow_zero_in_degree and bool((graph.in_degrees() == 0).any()):
            raise GraphError(
                "There are 0-in-degree nodes in the graph, output for those nodes will be invalid. "
                "Adding self-loop on the input graph by calling `g = g.add_self_loop()` will resolve "
                "the issue. Setting ``allow_zero_in_degree`` to be `True` when constructing this "
                "module will suppress the check and let the code run.")
        if weight is not None:
            if self.weight is not None:
                raise GraphError("External weight is provided while at the same time the module has "
                                 "defined its own weight parameter. Please create the module with "
                                 "flag weight=False.")
        else:
            weight = self.weight

        code, slope, unfused = _classify_activation(self._activation)
        if self.tie_order not in ("multiply_first", "aggregate_first"):
            raise GraphError(f'tie_order must be "multiply_first" or "aggregate_first", got "{self.tie_order}"')
        mult_first = self._in_feats > self._out_feats or (
            self._in_feats == self._out_feats and self.tie_order == "multiply_first")

        tracked = torch.is_grad_enabled() and (feat.requires_grad or (weight is not None and weight.requires_grad)
                                               or (self.bias is not None and self.bias.requires_grad))
        rst = None
        if (mult_first and weight is not None and edge_weight is None and not tracked
                and feat.dtype in (torch.float32, torch.bfloat16) and weight.dtype == torch.float32):
            # bfloat16 rows against the float32 weight: spmi_gemm_rows_bf16 keeps the weight's value
            rst = graphconv_inference(graph, feat, weight, self._norm, self.bias, code, slope)
        elif (mult_first and weight is not None and edge_weight is None and tracked
              and feat.dtype == torch.float32 and weight.dtype == torch.float32):
            # the same kernels under autograd (no per-edge weight in either direction)
            rst = graphconv_project_first(graph, feat, weight, self._norm, self.bias, code, slope)
        if weight is not None and weight.dtype != feat.dtype:
            weight = weight.to(feat.dtype)  # bf16 feature storage on the library path: the projection runs in bf16
        if rst is not None:
            pass
        elif mult_first:
            if weight is not None:
                feat = rows_times_weight(feat, weight)
            rst = gcn_aggregate(graph, feat, self._norm, self.bias, edge_weight, code, slope)
        else:
            rst = gcn_aggregate(graph, feat, self._norm, None, edge_weight, None)
            bias = None if self.bias is None else self.bias.to(rst.dtype)
            fused = None
            if weight is not None and not tracked and rst.dtype == torch.float32:
                fused = project_rows_inference(rst, weight, bias, code, slope)
            if fused is not None:
                rst, code = fused, None  # bias and activation rode in the projection's epilogue
            elif weight is not None:
                rst = rows_times_weight(rst, weight, bias)
            elif bias is not None:
                rst = rst + bias
            if code == "relu":
                rst = F.relu(rst)
            elif code == "leaky_relu":
                rst = F.leaky_relu(rst, slope)
        if unfused is not None:
            rst = unfused(rst)
        return rst

    def extra_repr(self) -> str:
        s = f"in={self._in_feats}, out={self._out_feats}, normalization={self._norm}"
        if self._activation is not None:
            s += f", activation={self._activation}"
        return s
