"""One snapshot across the GPUs of a node: destination-row blocks, one process per GPU.

The reference has no distributed code; this is the part of the path that shards.  Rank q
owns a contiguous block of destination nodes (cut at equal numbers of nonzeros, because
degrees are power-law), the rows of X, of the node memory and of the output for those
nodes, and the CSR rows of that block.  Per GCN layer there is exactly one exchange:

  forward   every rank receives the source rows its own edges reference, and only those,
            from the ranks that own them (one all-to-all over RCCL: on xGMI that is a direct
            transfer on every one of the 7 links of a GPU at once, where a ring all-gather
            moves everything through one link), then runs the fused aggregation over its
            rows; the indices of its CSR address a compact source table
            [own block | rows received];
  backward  the local transposed aggregation produces a partial gradient for every row
            of that table; the remote part travels back the same way and the owners add it
            to their rows.

How much that saves depends on how thin a rank's edges are spread over the node range: at
E/P edges per rank over N nodes with E/P close to N (8 ranks, 10 edges per node) a power-law
snapshot names about half of the remote rows from any one rank, so half of what an
all-gather would move stays at home; at 2 ranks nearly every row is named and the gain is
the direct transfer alone.  The transfer can be cut into `stages`
batches, each holding a slice of every peer's rows (so every link is busy in every batch):
the aggregation over the sources of batch k runs while batch k+1 is on the wire
(`spmi_gcn_spmm_part` adds batch after batch into the same output).  The layer is also
available in two halves (`graphconv_project` -> `exchange_rows_async` ...
`graphconv_aggregate`) so that independent work, e.g. the recurrent update of the previous
snapshot, runs while the rows travel, and a layer whose input never changes and is held by
every rank needs no exchange at all (`graphconv_replicated`).

Weight and bias gradients are summed with one all-reduce after the backward
(`allreduce_gradients`).  The recurrent update and the MLP are row-local: no exchange.
"""
from __future__ import annotations

import copy
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

    def block(self, q: Optional[int] = None):
        q = self.rank if q is None else q
        return self.bounds[q], self.bounds[q + 1]

    def owner(self, ids: torch.Tensor) -> torch.Tensor:
        """Rank that owns each global node id."""
        b = torch.tensor(self.bounds[1:-1], dtype=torch.int64, device=ids.device)
        return torch.bucketize(ids, b, right=True)


def _slice_bounds(count: int, stages: int) -> List[int]:
    """Cut `count` rows into `stages` consecutive slices; both ends of a transfer compute the
    same cuts from the same count."""
    return [count * k // stages for k in range(stages + 1)]


class ExchangePlan:
    """Which source rows travel where for one snapshot.

    Built from the global source ids of this rank's edges.  The source table this rank
    aggregates from is [own block (n_local rows, in place) | batch 0 | batch 1 | ...]; batch k
    holds slice k of the rows needed from every peer, peers in rank order, rows in ascending
    id order.  `columns` gives every edge its row of that table.  Building the plan costs two
    small all-to-alls (counts, then the requested row numbers) and two host round trips."""

    def __init__(self, src: torch.Tensor, part: RowPartition, group=None, stages: int = 1,
                 data_group=None):
        """`group`: communicator of the plan's own small collectives; `data_group`: communicator
        of the row exchanges.  A caller that keeps the exchange of one snapshot in flight while
        it builds the plan of the next passes two different ones, so that the small messages do
        not queue behind the rows (collectives of one communicator run in order)."""
        self.part, self.group, self.stages = part, group, max(int(stages), 1)
        self.data_group = data_group
        dev = src.device
        world, rank = part.world, part.rank
        lo, hi = part.block()
        self.n_local = hi - lo
        src = src.to(torch.int64)

        # distinct sources by mark and scan over the node range (no sort)
        flag = torch.zeros(part.n, dtype=torch.bool, device=dev)
        flag[src] = True
        flag[lo:hi] = False                                   # own rows stay where they are
        uniq = torch.nonzero(flag).reshape(-1)                # ascending remote ids needed here
        cut = torch.searchsorted(uniq, torch.tensor(part.bounds, dtype=torch.int64, device=dev)).tolist()
        need = [cut[q + 1] - cut[q] for q in range(world)]    # rows needed from rank q (0 for self)
        self.n_remote = len(uniq)
        self.n_table = self.n_local + self.n_remote

        # table row of every needed id: batch-major, then owner, then ascending id
        peers = [q for q in range(world) if q != rank]
        need_cuts = {q: _slice_bounds(need[q], self.stages) for q in peers}
        self.recv_counts = [[need_cuts[q][k + 1] - need_cuts[q][k] if q != rank else 0
                             for q in range(world)] for k in range(self.stages)]
        self.batch_rows = [sum(c) for c in self.recv_counts]
        self.batch_start = [self.n_local + sum(self.batch_rows[:k]) for k in range(self.stages + 1)]
        seg_pos, seg_row = [], []                             # per (owner, slice): start in uniq, start in table
        for q in peers:
            for k in range(self.stages):
                seg_pos.append(cut[q] + need_cuts[q][k])
                seg_row.append(self.batch_start[k] + sum(self.recv_counts[k][:q]))
        if self.n_remote:
            pos = torch.arange(self.n_remote, dtype=torch.int64, device=dev)
            seg_pos_t = torch.tensor(seg_pos, dtype=torch.int64, device=dev)
            seg_row_t = torch.tensor(seg_row, dtype=torch.int64, device=dev)
            seg = torch.bucketize(pos, seg_pos_t[1:], right=True)
            table_row = pos - seg_pos_t[seg] + seg_row_t[seg]
        else:
            table_row = torch.empty(0, dtype=torch.int64, device=dev)
        colmap = torch.full((part.n,), -1, dtype=torch.int64, device=dev)
        colmap[lo:hi] = torch.arange(self.n_local, dtype=torch.int64, device=dev)
        colmap[uniq] = table_row
        self._colmap = colmap
        # global id of every table row (own block first), e.g. to look up per-node vectors
        order = torch.empty(self.n_remote, dtype=torch.int64, device=dev)
        order[table_row - self.n_local] = uniq
        self.table_ids = torch.cat([torch.arange(lo, hi, dtype=torch.int64, device=dev), order])

        # tell every owner which of its rows are wanted here
        self.send_counts = [[0] * world for _ in range(self.stages)]
        self.send_index = torch.empty(0, dtype=torch.int64, device=dev)
        self.send_per_peer: List[int] = [0] * world
        if world > 1:
            counts_in = torch.tensor(need, dtype=torch.int64, device=dev)
            counts_out = torch.empty_like(counts_in)
            dist.all_to_all_single(counts_out, counts_in, group=group)
            wanted = counts_out.tolist()                      # rows rank p wants from this rank
            self.send_per_peer = wanted
            bounds_t = torch.tensor(part.bounds, dtype=torch.int64, device=dev)
            request = uniq - bounds_t[part.owner(uniq)]       # row inside the owner's block
            asked = torch.empty(sum(wanted), dtype=torch.int64, device=dev)
            dist.all_to_all_single(asked, request, wanted, need, group=group)
            # reorder the requests batch-major, the order the rows are sent in
            starts = [sum(wanted[:p]) for p in range(world)]
            pieces = []
            for k in range(self.stages):
                for p in range(world):
                    c = _slice_bounds(wanted[p], self.stages)
                    self.send_counts[k][p] = c[k + 1] - c[k]
                    pieces.append(asked[starts[p] + c[k]: starts[p] + c[k + 1]])
            self.send_index = torch.cat(pieces) if pieces else asked
        self.send_batch_start = [sum(sum(c) for c in self.send_counts[:k]) for k in range(self.stages + 1)]

    def columns(self, src: torch.Tensor) -> torch.Tensor:
        """Row of this rank's source table for every global source id of its edges."""
        return self._colmap[src.to(torch.int64)]

    def batch_of_column(self, cols: torch.Tensor) -> torch.Tensor:
        """Batch whose arrival completes a column: own rows and batch 0 are 0."""
        b = torch.tensor(self.batch_start[1:-1], dtype=torch.int64, device=cols.device)
        return torch.bucketize(cols, b, right=True)


class PendingTable:
    """A source table whose remote batches may still be on the wire.  `wait(k)` returns once
    batches 0..k have arrived (k defaults to all); the transfers run on the communicator's
    own stream, so whatever the caller launches in between overlaps them."""

    def __init__(self, table: torch.Tensor, works: list, keep, prescaled: bool):
        self.table, self._works, self._keep, self.prescaled = table, list(works), keep, prescaled
        self._done = 0

    def wait(self, k: Optional[int] = None) -> torch.Tensor:
        upto = len(self._works) if k is None else min(k + 1, len(self._works))
        while self._done < upto:
            if self._works[self._done] is not None:
                self._works[self._done].wait()
            self._done += 1
        if self._done == len(self._works):
            self._keep = None
        return self.table


def exchange_rows_async(z_local: torch.Tensor, plan: ExchangePlan,
                        payload_dtype: Optional[torch.dtype] = None,
                        table: Optional[torch.Tensor] = None, prescaled: bool = False) -> PendingTable:
    """Start moving the rows of `z_local` (this rank's block, [n_local, F]) that other ranks
    need, and the rows this rank needs from them, into its source table.  One all-to-all per
    batch of the plan, queued back to back.  `table`: a [n_table, F] tensor whose head already
    holds `z_local` (the projection wrote it there), otherwise one is allocated and the block
    copied in.  payload_dtype=torch.bfloat16 halves the bytes on the wire; the aggregation then
    reads bfloat16 rows and still sums in float32 (a documented precision trade, off by
    default)."""
    z = z_local.detach()
    dtype = payload_dtype or z.dtype
    f = z.shape[1]
    if table is None or table.dtype != dtype:
        table = torch.empty(plan.n_table, f, dtype=dtype, device=z.device)
        table[: plan.n_local] = z
    works = []
    send = None
    if plan.part.world > 1:
        send = z.index_select(0, plan.send_index)
        if send.dtype != dtype:
            send = send.to(dtype)
        for k in range(plan.stages):
            out = table[plan.batch_start[k]: plan.batch_start[k + 1]]
            inp = send[plan.send_batch_start[k]: plan.send_batch_start[k + 1]]
            works.append(dist.all_to_all_single(out, inp, plan.recv_counts[k], plan.send_counts[k],
                                                group=plan.data_group, async_op=True))
    else:
        works = [None] * plan.stages
    return PendingTable(table, works, send, prescaled)


def _return_rows(g_table: torch.Tensor, plan: ExchangePlan) -> torch.Tensor:
    """The reverse transfer for the backward pass: the remote part of a gradient over the
    source table goes back to the owners of those rows, who add what they receive to their
    own block's part.  Returns the gradient of this rank's block [n_local, F].  The rows one
    peer sends back are distinct, so the additions are done peer by peer without atomics
    (bit-reproducible)."""
    g_local = g_table[: plan.n_local].clone()
    if plan.part.world == 1:
        return g_local
    f = g_table.shape[1]
    back = g_table.new_empty(plan.send_index.numel(), f)
    for k in range(plan.stages):
        inp = g_table[plan.batch_start[k]: plan.batch_start[k + 1]].contiguous()
        out = back[plan.send_batch_start[k]: plan.send_batch_start[k + 1]]
        dist.all_to_all_single(out, inp, plan.send_counts[k], plan.recv_counts[k], group=plan.data_group)
    off = 0
    for k in range(plan.stages):
        for p in range(plan.part.world):
            c = plan.send_counts[k][p]
            if c:
                g_local.index_add_(0, plan.send_index[off: off + c], back[off: off + c])
            off += c
    return g_local


def _with_indices(csr, indices: torch.Tensor, n_cols: int):
    """The same grouping (row pointers, hub-row plan) over another numbering of the columns."""
    out = copy.copy(csr)
    out.indices, out.n_cols = indices, n_cols
    return out


class ShardedSnapshot(G.NormalisedAdjacency):
    """This rank's destination block of one snapshot: the edges that end in the block, their
    sources renumbered to rows of the compact source table of the exchange plan."""

    def __init__(self, src: torch.Tensor, dst: torch.Tensor, part: RowPartition, group=None,
                 stages: int = 1, data_group=None):
        """src/dst: the edges whose destination lies in this rank's block (global ids).
        `group` carries the per-snapshot bookkeeping (plan, out-degrees), `data_group` the row
        exchanges (see ExchangePlan)."""
        self.part, self.group = part, group
        lo, hi = part.block()
        self.n_local = hi - lo
        self.src_global = src.to(torch.int64)
        self.plan = ExchangePlan(self.src_global, part, group, stages, data_group)
        self.cols = self.plan.columns(self.src_global)
        self.dst_rows = dst.to(torch.int64) - lo
        self._csr = self._csr_t = self._batch_csrs = self._global = None
        self._out_degree = None
        self._norms = {}

    @classmethod
    def from_global(cls, src, dst, part: RowPartition, group=None, stages: int = 1,
                    data_group=None) -> "ShardedSnapshot":
        """Keep the edges of a full edge list that end in this rank's block."""
        lo, hi = part.block()
        keep = (dst >= lo) & (dst < hi)
        return cls(src[keep], dst[keep], part, group, stages, data_group)

    def csr(self, want_eid: bool = False, wait: bool = True):
        if self._csr is None or (want_eid and self._csr.eid is None):
            self._csr = kernels.build_csr(self.dst_rows, self.cols, self.n_local, self.plan.n_table,
                                          want_eid=want_eid)
        return self._csr

    def csr_t(self, want_eid: bool = False, wait: bool = True):
        if self._csr_t is None:
            self._csr_t = kernels.build_csr(self.cols, self.dst_rows, self.plan.n_table, self.n_local)
        return self._csr_t

    def batch_csrs(self) -> list:
        """One CSR per batch of the plan: the edges whose source row arrives with that batch
        (own rows count to batch 0)."""
        if self._batch_csrs is None:
            if self.plan.stages == 1:
                self._batch_csrs = [self.csr()]
            else:
                batch = self.plan.batch_of_column(self.cols)
                self._batch_csrs = []
                for k in range(self.plan.stages):
                    keep = batch == k
                    self._batch_csrs.append(kernels.build_csr(self.dst_rows[keep], self.cols[keep],
                                                              self.n_local, self.plan.n_table))
        return self._batch_csrs

    def out_degree(self) -> torch.Tensor:
        """Out-degree of every node of the graph (int32 [N]): per-rank partial counts summed
        with one all-reduce per snapshot."""
        if self._out_degree is None:
            deg = kernels.degree(self.src_global, self.part.n).to(torch.int32)
            if self.part.world > 1:
                dist.all_reduce(deg, group=self.group)
            self._out_degree = deg
        return self._out_degree

    def _norm_of(self, deg: torch.Tensor, mode: int) -> torch.Tensor:
        d = deg.to(torch.float32).clamp_(min=1)
        return d.sqrt_().reciprocal_() if mode == 0 else d.reciprocal_()

    def norm(self, side: str, mode: int) -> torch.Tensor:
        """'in': this block's destination normaliser (local rows are complete).
        'out': source normaliser of every row of this rank's source table."""
        key = (side, mode)
        if key not in self._norms:
            if side == "in":
                self._norms[key] = self._norm_of(self.csr().degrees(), mode)
            else:
                self._norms[key] = self.norm_global(mode).index_select(0, self.plan.table_ids)
        return self._norms[key]

    def norm_global(self, mode: int) -> torch.Tensor:
        """Source normaliser of every node of the graph [N]."""
        key = ("global", mode)
        if key not in self._norms:
            self._norms[key] = self._norm_of(self.out_degree(), mode)
        return self._norms[key]

    def norm_out_local(self, mode: int) -> torch.Tensor:
        """Source normaliser of this rank's own rows [n_local]: what the owner multiplies its
        projected rows by before they travel."""
        lo, hi = self.part.block()
        return self.norm_global(mode)[lo:hi]

    def global_view(self) -> "_GlobalSources":
        """The same block with sources numbered by global node id, for inputs every rank holds
        in full."""
        if self._global is None:
            self._global = _GlobalSources(self)
        return self._global


class _GlobalSources(G.NormalisedAdjacency):
    """A ShardedSnapshot whose CSR addresses a matrix with one row per node of the graph."""

    def __init__(self, shard: ShardedSnapshot):
        self.shard = shard
        self._csr = self._csr_t = None

    def csr(self, want_eid: bool = False, wait: bool = True):
        if self._csr is None:
            base = self.shard.csr(want_eid=True)
            cols = self.shard.src_global.index_select(0, base.eid.to(torch.int64)).to(base.indices.dtype)
            self._csr = _with_indices(base, cols, self.shard.part.n)
        return self._csr

    def csr_t(self, want_eid: bool = False, wait: bool = True):
        if self._csr_t is None:
            self._csr_t = kernels.build_csr(self.shard.src_global, self.shard.dst_rows, self.shard.part.n,
                                            self.shard.n_local)
        return self._csr_t

    def norm(self, side: str, mode: int) -> torch.Tensor:
        return self.shard.norm("in", mode) if side == "in" else self.shard.norm_global(mode)


class _ShardedAggregate(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x_local, bias, shard: ShardedSnapshot, norm, act, slope):
        mode = 0 if norm == "both" else 1
        table = exchange_rows_async(x_local, shard.plan).wait()
        dst_scale = shard.norm("in", mode) if norm in ("right", "both") else None
        y = kernels.aggregate(shard.csr(), table, None, shard.edge_scale("fwd", norm), dst_scale,
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
            g_table = kernels.aggregate(shard.csr_t(), g_p.contiguous(), None,
                                        shard.edge_scale("bwd", norm), out_scale, None, None)
            g_x = _return_rows(g_table, shard.plan)
        return g_x, g_b, None, None, None, None


class _TableAggregate(torch.autograd.Function):
    """ops.GcnAggregate through the `kernels` indirection of this module: the aggregation of
    a matrix this rank already holds in full over any NormalisedAdjacency `g` (no exchange)."""

    @staticmethod
    def forward(ctx, x, bias, g, norm, act, slope):
        mode = 0 if norm == "both" else 1
        dst_scale = g.norm("in", mode) if norm in ("right", "both") else None
        y = kernels.aggregate(g.csr(), x.detach(), None, g.edge_scale("fwd", norm), dst_scale,
                              None if bias is None else bias.detach(), act, slope)
        ctx.g, ctx.norm, ctx.act, ctx.slope, ctx.has_bias = g, norm, act, slope, bias is not None
        ctx.save_for_backward(y if act not in (None, "none") else None)
        return y

    @staticmethod
    def backward(ctx, g_y):
        (y,) = ctx.saved_tensors
        g, norm = ctx.g, ctx.norm
        need_b = ctx.needs_input_grad[1] and ctx.has_bias
        g_p, g_b = g_y, None
        if y is not None or need_b:
            g_p, g_b = kernels.act_bias_bwd(g_y, y if y is not None else g_y, ctx.act, ctx.slope, need_b)
        g_x = None
        if ctx.needs_input_grad[0]:
            mode = 0 if norm == "both" else 1
            out_scale = g.norm("out", mode) if norm in ("left", "both") else None
            g_x = kernels.aggregate(g.csr_t(), g_p.contiguous(), None, g.edge_scale("bwd", norm),
                                    out_scale, None, None)
        return g_x, g_b, None, None, None, None


def sharded_aggregate(shard: ShardedSnapshot, x_local: torch.Tensor, norm: str = "both",
                      bias=None, act=None, slope: float = 0.01) -> torch.Tensor:
    """Y_local = act(norm_in * A_local (norm_out * exchanged(X_local)) + bias), differentiable."""
    return _ShardedAggregate.apply(x_local, bias, shard, norm, act, slope)


def _split_activation(conv):
    from .nn.graphconv import _classify_activation

    return _classify_activation(conv._activation)


def _tracked(conv, *tensors) -> bool:
    return torch.is_grad_enabled() and (any(t is not None and t.requires_grad for t in tensors) or any(
        p is not None and p.requires_grad for p in (conv.weight, conv.bias)))


def _finish(conv, rst: torch.Tensor, code, slope, unfused) -> torch.Tensor:
    """What follows the aggregation when the layer aggregates first (in_feats < out_feats)."""
    import torch.nn.functional as F

    weight = conv.weight
    if weight is not None and not _tracked(conv, rst):
        fused = ops.project_rows_inference(rst, weight, conv.bias, code, slope)
        if fused is not None:  # projection, bias and activation in one pass over the rows
            return fused if unfused is None else unfused(fused)
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
    same parameters and order of operations as GraphConv.forward, with the exchange in
    front of the aggregation (after X W when in_feats >= out_feats, so the smaller matrix
    travels).  Differentiable; without autograd it takes the two-half path below."""
    if not _tracked(conv, feat_local):
        return graphconv_aggregate(conv, shard, graphconv_project(conv, feat_local, shard))
    norm = conv._norm
    code, slope, unfused = _split_activation(conv)
    if conv._in_feats >= conv._out_feats:
        if conv.weight is not None:
            feat_local = ops.rows_times_weight(feat_local, conv.weight)
        rst = sharded_aggregate(shard, feat_local, norm, conv.bias, code, slope)
        return rst if unfused is None else unfused(rst)
    return _finish(conv, sharded_aggregate(shard, feat_local, norm), code, slope, unfused)


# ---- the same layer in two halves, for callers that overlap the exchange -----------------
def graphconv_project(conv, feat_local: torch.Tensor, shard: ShardedSnapshot,
                      payload_dtype: Optional[torch.dtype] = None) -> PendingTable:
    """First half of a layer without autograd: produce the rows that have to travel (X W when
    the layer multiplies first, X itself otherwise) and start their exchange.  When the
    projection runs on spmi_gemm_rows it writes straight into the head of the source table and
    applies the source normaliser in its epilogue, so the aggregation needs no per-edge
    weight."""
    plan, norm = shard.plan, conv._norm
    first = conv._in_feats >= conv._out_feats and conv.weight is not None
    feat_local = feat_local.detach()
    if not first:
        return exchange_rows_async(feat_local, plan, payload_dtype)
    w = conv.weight.detach()
    mode = 0 if norm == "both" else 1
    if ops._gemm_rows_ok(feat_local, w.shape[0], w.shape[1]):
        scale = shard.norm_out_local(mode).contiguous() if norm in ("both", "left") else None
        table = None
        if payload_dtype in (None, torch.float32):
            table = torch.empty(plan.n_table, w.shape[1], dtype=torch.float32, device=feat_local.device)
        z = ops.gemm_rows(feat_local, w.t(), None, None, 0.01, scale,
                          planes=ops.weight_planes(conv.weight, "f16x3", True, ops.F16X3_GEMM_ROWS_LIMIT),
                          out=None if table is None else table[: plan.n_local])
        return exchange_rows_async(z, plan, payload_dtype, table=table, prescaled=True)
    return exchange_rows_async(feat_local @ w, plan, payload_dtype)


def _edge_scale_of(shard, csr, norm: str) -> Optional[torch.Tensor]:
    if norm not in ("left", "both"):
        return None
    return shard.norm("out", 0 if norm == "both" else 1).index_select(0, csr.indices.to(torch.int64))


def graphconv_aggregate(conv, shard: ShardedSnapshot, pending: PendingTable) -> torch.Tensor:
    """Second half without autograd: the aggregation of the source table into this rank's
    destination block, batch by batch as the batches arrive, then bias/activation (and W when
    the layer aggregates first).  Sums are float32 whatever the payload type."""
    norm = conv._norm
    code, slope, unfused = _split_activation(conv)
    first = conv._in_feats >= conv._out_feats
    mode = 0 if norm == "both" else 1
    dst_scale = shard.norm("in", mode) if norm in ("right", "both") else None
    bias = conv.bias.detach() if (first and conv.bias is not None) else None
    act = code if first else None
    csrs = shard.batch_csrs()
    rst = None
    for k, csr in enumerate(csrs):
        table = pending.wait(k)
        edge_w = None if pending.prescaled else _edge_scale_of(shard, csr, norm)
        last = k == len(csrs) - 1
        if len(csrs) == 1:
            rst = kernels.aggregate(csr, table, None, edge_w, dst_scale, bias, act, slope,
                                    out_dtype=torch.float32)
        else:
            rst = kernels.aggregate(csr, table, None, edge_w, dst_scale, bias, act, slope, out=rst,
                                    out_dtype=torch.float32, accumulate=k > 0, finalize=last)
    if first:
        return rst if unfused is None else unfused(rst)
    return _finish(conv, rst, code, slope, unfused)


def graphconv_replicated(conv, shard: ShardedSnapshot, x_full: torch.Tensor) -> torch.Tensor:
    """A layer whose input every rank holds in full ([N, F] in node order, e.g. the node table
    of a VEInvariantDTDG, which never changes): no exchange, and no redundant work either,
    because the rows are aggregated into this rank's block first and only the block is
    projected: A (X W) = (A X) W, the order DGL itself takes when in_feats <= out_feats.
    Projecting the replicated matrix first would cost every rank the GEMM of the whole graph.
    Differentiable with respect to the layer's parameters."""
    code, slope, unfused = _split_activation(conv)
    rst = _TableAggregate.apply(x_full, None, shard.global_view(), conv._norm, None, 0.01)
    return _finish(conv, rst, code, slope, unfused)


def allreduce_gradients(module: torch.nn.Module, group=None) -> None:
    """Sum parameter gradients over ranks in one flat all-reduce (they are a few KB-MB).
    The sum is the gradient of the global loss only if every rank's loss is its share of one
    sum normalised by the global row count (e.g. `F.mse_loss(..., reduction="sum") / N_total`):
    a per-rank mean would weight rows by the inverse size of their block."""
    grads = [p.grad for p in module.parameters() if p.grad is not None]
    if not grads or not dist.is_initialized() or dist.get_world_size(group) == 1:
        return
    flat = torch.cat([g.reshape(-1) for g in grads])
    dist.all_reduce(flat, group=group)
    off = 0
    for g in grads:
        g.copy_(flat[off: off + g.numel()].view_as(g))
        off += g.numel()
