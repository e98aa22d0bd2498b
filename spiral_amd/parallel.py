"""One snapshot across the GPUs of a node: destination-row blocks, one process per GPU.

The reference has no distributed code; this is the part of the path that shards.  Rank q
owns a contiguous block of destination nodes (cut at equal numbers of nonzeros, because
degrees are power-law), the rows of X, of the node memory and of the output for those
nodes, and the CSR rows of that block.  Per GCN layer there is exactly one exchange:

  forward   all-gather of the source feature rows over RCCL/xGMI, then the local fused
            aggregation over this rank's rows (indices address the gathered matrix);
  backward  the local transposed aggregation produces a partial gradient for every
            source row; reduce-scatter sums the partials and hands each rank its rows.

Weight and bias gradients are summed with one all-reduce after the backward
(`allreduce_gradients`).  The recurrent update and the MLP are row-local: no exchange.

Blocks are padded to the largest block so that one `all_gather_into_tensor` /
`reduce_scatter_tensor` moves everything; source ids are remapped once per snapshot to
rows of the padded gathered matrix, so no compaction pass runs after the collective.

At 8 GPUs the gather, not the kernels, bounds a layer (every rank ingests 7/8 of the
projected matrix).  Two ways around it are provided for inference loops: a layer whose
input never changes and is held by every rank (`replicate_rows`) is projected redundantly
and needs no exchange at all, and the layer is available in two halves
(`graphconv_project` -> `gather_rows_async` ... `PendingGather.wait` ->
`graphconv_aggregate`) so that independent work, e.g. the recurrent update of the previous
snapshot, runs while the rows travel.
"""
from __future__ import annotations

from dataclasses import dataclass
from types import SimpleNamespace
from typing import List, Optional

import torch
import torch.distributed as dist

from . import graph as G
from . import ops

# Kernel entry points used below.  They are the libspmi-backed functions; the CPU tests of
# the exchange logic (gloo, no GPU) substitute checker implementations here.
kernels = SimpleNamespace(build_csr=G.build_csr, aggregate=G.aggregate, degree=G.degree,
                          act_bias_bwd=ops._act_bias_bwd)


def edge_balanced_bounds(in_degree: torch.Tensor, world: int) -> List[int]:
    """Row cuts [b_0=0, ..., b_world=N] such that every block holds about E/world nonzeros."""
    n = in_degree.numel()
    if world == 1 or n == 0:
        return [0] + [n] * world
    csum = torch.cumsum(in_degree.to(torch.int64), 0)
    total = int(csum[-1])
    targets = torch.arange(1, world, dtype=torch.int64, device=csum.device) * total // world
    cuts = (torch.searchsorted(csum, targets, right=False) + 1).clamp_(max=n).tolist()
    bounds = [0] + cuts + [n]
    for i in range(1, len(bounds)):
        bounds[i] = max(bounds[i], bounds[i - 1])
    return bounds


@dataclass
class RowPartition:
    """Block bounds shared by all ranks plus this rank's position."""

    n: int
    bounds: List[int]
    rank: int
    world: int

    @property
    def max_block(self) -> int:
        return max(self.bounds[q + 1] - self.bounds[q] for q in range(self.world))

    @property
    def n_gather(self) -> int:
        return self.world * self.max_block

    def block(self, q: Optional[int] = None):
        q = self.rank if q is None else q
        return self.bounds[q], self.bounds[q + 1]

    def padded_rows(self, ids: torch.Tensor) -> torch.Tensor:
        """Global node id -> row of the padded gathered matrix [world * max_block, F]."""
        b = torch.tensor(self.bounds, dtype=torch.int64, device=ids.device)
        q = torch.bucketize(ids, b[1:-1], right=True)
        return q * self.max_block + ids - b[q]


def _gather_rows(x_local: torch.Tensor, part: RowPartition, group) -> torch.Tensor:
    f = x_local.shape[1]
    send = x_local
    if x_local.shape[0] != part.max_block or not x_local.is_contiguous():
        send = x_local.new_zeros(part.max_block, f)
        send[: x_local.shape[0]] = x_local
    out = x_local.new_empty(part.n_gather, f)
    dist.all_gather_into_tensor(out, send, group=group)
    return out


def _scatter_sum_rows(g_full: torch.Tensor, part: RowPartition, group) -> torch.Tensor:
    out = g_full.new_empty(part.max_block, g_full.shape[1])
    dist.reduce_scatter_tensor(out, g_full.contiguous(), group=group)
    lo, hi = part.block()
    return out[: hi - lo]


class ShardedSnapshot(G.NormalisedAdjacency):
    """This rank's destination block of one snapshot."""

    def __init__(self, src: torch.Tensor, dst: torch.Tensor, part: RowPartition, group=None):
        """src/dst: the edges whose destination lies in this rank's block (global ids)."""
        self.part, self.group = part, group
        lo, hi = part.block()
        self.n_local = hi - lo
        self.src_rows = part.padded_rows(src.to(torch.int64))
        self.dst_rows = dst.to(torch.int64) - lo
        self._csr = self._csr_t = None
        self._norms = {}

    @classmethod
    def from_global(cls, src, dst, part: RowPartition, group=None) -> "ShardedSnapshot":
        """Keep the edges of a full edge list that end in this rank's block."""
        lo, hi = part.block()
        keep = (dst >= lo) & (dst < hi)
        return cls(src[keep], dst[keep], part, group)

    def csr(self, want_eid: bool = False, wait: bool = True):
        if self._csr is None:
            self._csr = kernels.build_csr(self.dst_rows, self.src_rows, self.n_local,
                                          self.part.n_gather)
        return self._csr

    def csr_t(self, want_eid: bool = False, wait: bool = True):
        if self._csr_t is None:
            self._csr_t = kernels.build_csr(self.src_rows, self.dst_rows, self.part.n_gather,
                                            self.n_local)
        return self._csr_t

    def norm(self, side: str, mode: int) -> torch.Tensor:
        """'in': this block's destination normaliser (local rows are complete).
        'out': source normaliser for every row of the gathered matrix; out-degrees are
        partial per rank and summed with one all-reduce."""
        key = (side, mode)
        if key not in self._norms:
            if side == "in":
                deg = self.csr().degrees()
            else:
                deg = kernels.degree(self.src_rows, self.part.n_gather)
                if self.part.world > 1:
                    deg = deg.to(torch.int32)  # degrees fit int32: half the bytes on the wire
                    dist.all_reduce(deg, group=self.group)
            d = deg.to(torch.float32).clamp_(min=1)
            self._norms[key] = d.sqrt_().reciprocal_() if mode == 0 else d.reciprocal_()
        return self._norms[key]


class _ShardedAggregate(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x_local, bias, shard: ShardedSnapshot, norm, act, slope):
        mode = 0 if norm == "both" else 1
        xg = _gather_rows(x_local.detach(), shard.part, shard.group)
        dst_scale = shard.norm("in", mode) if norm in ("right", "both") else None
        y = kernels.aggregate(shard.csr(), xg, None, shard.edge_scale("fwd", norm), dst_scale,
                              None if bias is None else bias.detach(), act, slope)
        ctx.shard, ctx.norm, ctx.act, ctx.slope, ctx.has_bias = shard, norm, act, slope, bias is not None
        ctx.save_for_backward(y if act not in (None, "none") else None)
        return y

    @staticmethod
    def backward(ctx, g_y):
        (y,) = ctx.saved_tensors
        shard, norm = ctx.shard, ctx.norm
        need_b = ctx.needs_input_grad[1] and ctx.has_bias
        g_p, g_b = g_y, None
        if y is not None or need_b:
            g_p, g_b = kernels.act_bias_bwd(g_y, y if y is not None else g_y, ctx.act, ctx.slope,
                                            need_b)
        g_x = None
        if ctx.needs_input_grad[0]:
            mode = 0 if norm == "both" else 1
            out_scale = shard.norm("out", mode) if norm in ("left", "both") else None
            g_full = kernels.aggregate(shard.csr_t(), g_p.contiguous(), None,
                                       shard.edge_scale("bwd", norm), out_scale, None, None)
            g_x = _scatter_sum_rows(g_full, shard.part, shard.group)
        return g_x, g_b, None, None, None, None


def sharded_aggregate(shard: ShardedSnapshot, x_local: torch.Tensor, norm: str = "both",
                      bias=None, act=None, slope: float = 0.01) -> torch.Tensor:
    """Y_local = act(norm_in * A_local (norm_out * allgather(X_local)) + bias)."""
    return _ShardedAggregate.apply(x_local, bias, shard, norm, act, slope)


def _split_activation(conv):
    from .nn.graphconv import _classify_activation

    return _classify_activation(conv._activation)


def _finish(conv, rst: torch.Tensor, code, slope, unfused) -> torch.Tensor:
    """What follows the aggregation when the layer aggregates first (in_feats < out_feats)."""
    import torch.nn.functional as F

    weight = conv.weight
    tracked = torch.is_grad_enabled() and (rst.requires_grad or any(
        p is not None and p.requires_grad for p in (weight, conv.bias)))
    if (weight is not None and not tracked and code in (None, "relu", "leaky_relu")
            and ops._gemm_rows_ok(rst, weight.shape[0], weight.shape[1])):
        # projection, bias and activation in one pass over the rows
        rst = ops.gemm_rows(rst, weight.detach().t(), conv.bias, code, slope,
                            planes=ops.weight_planes(weight, "f16x3", True))
        return rst if unfused is None else unfused(rst)
    if weight is not None:
        rst = ops.rows_times_weight(rst, weight, conv.bias)
    elif conv.bias is not None:
        rst = rst + conv.bias
    if code == "relu":
        rst = F.relu(rst)
    elif code == "leaky_relu":
        rst = F.leaky_relu(rst, slope)
    return rst if unfused is None else unfused(rst)


def sharded_graphconv(conv, shard: ShardedSnapshot, feat_local: torch.Tensor) -> torch.Tensor:
    """A spiral_amd GraphConv layer applied to this rank's rows of a sharded snapshot:
    same parameters and order of operations as GraphConv.forward, with the all-gather in
    front of the aggregation (after X W when in_feats >= out_feats, so the smaller matrix
    travels)."""
    norm = conv._norm
    code, slope, unfused = _split_activation(conv)
    if conv._in_feats >= conv._out_feats:
        if conv.weight is not None:
            feat_local = ops.rows_times_weight(feat_local, conv.weight)
        rst = sharded_aggregate(shard, feat_local, norm, conv.bias, code, slope)
        return rst if unfused is None else unfused(rst)
    return _finish(conv, sharded_aggregate(shard, feat_local, norm), code, slope, unfused)


# ---- the same layer in two halves, for callers that overlap the exchange -----------------
def replicate_rows(x_global: torch.Tensor, part: RowPartition) -> torch.Tensor:
    """Padded-gather layout [world * max_block, F] of a matrix every rank holds in full
    (input features that never change, e.g. the node table of a VEInvariantDTDG)."""
    out = x_global.new_zeros(part.n_gather, x_global.shape[1])
    rows = part.padded_rows(torch.arange(part.n, dtype=torch.int64, device=x_global.device))
    out[rows] = x_global
    return out


class PendingGather:
    """An all-gather of source rows in flight (or finished).  `wait()` returns the gathered
    padded matrix; the exchange runs on the communicator's own stream, so whatever the
    caller launches between `gather_rows_async` and `wait` overlaps it."""

    def __init__(self, out: torch.Tensor, send: torch.Tensor, work):
        self._out, self._send, self._work = out, send, work

    def wait(self) -> torch.Tensor:
        if self._work is not None:
            self._work.wait()
            self._work = None
        self._send = None
        return self._out


def gather_rows_async(x_local: torch.Tensor, part: RowPartition, group=None,
                      payload_dtype: Optional[torch.dtype] = None) -> PendingGather:
    """Start the all-gather of this rank's (projected) rows.  payload_dtype=torch.bfloat16
    halves the bytes on the wire; the aggregation then reads bfloat16 rows and still sums in
    float32 (a documented precision trade, off by default)."""
    send = x_local.detach()
    if payload_dtype is not None and send.dtype != payload_dtype:
        send = send.to(payload_dtype)
    if send.shape[0] != part.max_block or not send.is_contiguous():
        padded = send.new_zeros(part.max_block, send.shape[1])
        padded[: send.shape[0]] = send
        send = padded
    out = send.new_empty(part.n_gather, send.shape[1])
    work = dist.all_gather_into_tensor(out, send, group=group, async_op=True)
    return PendingGather(out, send, work)


def graphconv_project(conv, feat: torch.Tensor) -> torch.Tensor:
    """First half of a layer: the rows that have to reach every rank (X W when the layer
    multiplies first, X itself otherwise).  Row-local, so it applies to a rank's own rows
    or to a replicated matrix alike."""
    if conv._in_feats >= conv._out_feats and conv.weight is not None:
        return ops.rows_times_weight(feat, conv.weight)
    return feat


def graphconv_replicated(conv, shard: ShardedSnapshot, x_full: torch.Tensor) -> torch.Tensor:
    """A layer whose input every rank holds in full (`replicate_rows`): no exchange, and no
    redundant work either, because the rows are aggregated into this rank's block first and
    only the block is projected: A (X W) = (A X) W, the order DGL itself takes when
    in_feats <= out_feats.  Projecting the replicated matrix first would cost every rank the
    GEMM of the whole graph."""
    code, slope, unfused = _split_activation(conv)
    rst = ops.gcn_aggregate(shard, x_full, conv._norm, None, None, None)
    return _finish(conv, rst, code, slope, unfused)


def graphconv_aggregate(conv, shard: ShardedSnapshot, z_full: torch.Tensor) -> torch.Tensor:
    """Second half: aggregation of the gathered (or replicated) projected rows into this
    rank's destination block, then bias/activation (and W when the layer aggregates
    first).  Differentiable with respect to `z_full` and the layer parameters; the gradient
    of `z_full` is this rank's partial sum over its own edges."""
    norm = conv._norm
    code, slope, unfused = _split_activation(conv)
    first = conv._in_feats >= conv._out_feats
    if z_full.dtype != torch.float32:
        # reduced-precision payload (inference only): float32 sums and a float32 result
        mode = 0 if norm == "both" else 1
        dst_scale = shard.norm("in", mode) if norm in ("right", "both") else None
        bias = conv.bias.detach() if (first and conv.bias is not None) else None
        rst = kernels.aggregate(shard.csr(), z_full.detach(), None, shard.edge_scale("fwd", norm),
                                dst_scale, bias, code if first else None, slope,
                                out_dtype=torch.float32)
    else:
        rst = ops.gcn_aggregate(shard, z_full, norm, conv.bias if first else None, None,
                                code if first else None, slope)
    if first:
        return rst if unfused is None else unfused(rst)
    return _finish(conv, rst, code, slope, unfused)


def allreduce_gradients(module: torch.nn.Module, group=None) -> None:
    """Sum parameter gradients over ranks in one flat all-reduce (they are a few KB-MB).
    Every rank computed the loss over its own rows, so the sum is the full gradient."""
    grads = [p.grad for p in module.parameters() if p.grad is not None]
    if not grads or not dist.is_initialized() or dist.get_world_size(group) == 1:
        return
    flat = torch.cat([g.reshape(-1) for g in grads])
    dist.all_reduce(flat, group=group)
    off = 0
    for g in grads:
        g.copy_(flat[off: off + g.numel()].view_as(g))
        off += g.numel()
